"""rocprofv3 target: RoIAlignRotated backward (tiled form) on the 256^2 x 256 level, 20 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sm3det_amd import mmcv_ops as ops
from tests import synth
os.environ['SM3_ROI_BWD'] = sys.argv[1] if len(sys.argv) > 1 else 'tiled'
xl = torch.randn(1, 256, 256, 256, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
rois = torch.from_numpy(np.ascontiguousarray(synth.rois_for_level(512, 6, batch=1, extent=1024.0))).cuda()
layer = ops.RoIAlignRotated(output_size=7, spatial_scale=0.25, sampling_ratio=2, clockwise=True)
y = layer(xl, rois)
go = torch.randn_like(y)
for _ in range(20):
    torch.autograd.grad(y, xl, go, retain_graph=True)
torch.cuda.synchronize()
