"""nms_rotated N = 10 000 / 2 000 and box_iou_rotated 2000 x 512 in a loop, for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sm3det_amd import mmcv_ops as ops
from tests import synth
for n in (10000, 2000):
    b = torch.from_numpy(synth.rotated_boxes(n, 0)).cuda() if hasattr(synth, 'rotated_boxes') else None
    s = torch.from_numpy(synth.unique_scores(n, 1)).cuda()
    for _ in range(10):
        ops.nms_rotated(b, s, 0.1)
b1 = torch.from_numpy(synth.rotated_boxes(2000, 2)).cuda(); b2 = torch.from_numpy(synth.rotated_boxes(512, 3)).cuda()
for _ in range(10):
    ops.box_iou_rotated(b1, b2)
torch.cuda.synchronize()
