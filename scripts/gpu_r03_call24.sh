#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_assign_gpu.py tests/test_losses_gpu.py tests/test_detector_slice_gpu.py -m gpu -q 2>&1 | tail -1
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from tests import synth
from sm3det_amd.assign import MaxIoUAssigner
from sm3det_amd.rpn_head import grid_anchors as _ga
dev = lambda a: torch.from_numpy(a).cuda()
anc = torch.cat(_ga([(256 >> i, 256 >> i) for i in range(5)], [4, 8, 16, 32, 64], [8], [0.5, 1.0, 2.0], device='cuda'))
ghb = dev(synth.hboxes(8, 31, extent=1024.0))
asg = MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True)
for _ in range(3): asg.assign(anc, ghb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): asg.assign(anc, ghb)
e1.record(); torch.cuda.synchronize()
print('max_iou_assign_rpn_261888x8 us', e0.elapsed_time(e1) / 20 * 1e3)
PY
