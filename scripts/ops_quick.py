"""Quick A/B timings of the round-4 operator kernels (microseconds, median of 3 x 10 calls):
RoIAlignRotated fwd / bwd (tiled vs atomic) on the 256^2 x 256 level, fused extractor, DeformConv2d fwd fused vs im2col+GEMM."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sm3det_amd import mmcv_ops as ops  # noqa: E402
from tests import synth  # noqa: E402


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    b = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        b.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(b)[1]


dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
x = torch.randn(1, 256, 256, 256, device='cuda', requires_grad=True)
xl = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
rois = dev(synth.rois_for_level(512, 6, batch=1, extent=1024.0))
layer = ops.RoIAlignRotated(output_size=7, spatial_scale=0.25, sampling_ratio=2, clockwise=True)
print('roi fwd nchw', round(timeit(lambda: layer(x, rois)), 1), 'nhwc', round(timeit(lambda: layer(xl, rois)), 1), flush=True)
y, yl = layer(x, rois), layer(xl, rois)
go = torch.randn_like(y)
for mode in ('tiled', 'atomic'):
    os.environ['SM3_ROI_BWD'] = mode
    print('roi bwd', mode, 'nchw', round(timeit(lambda: torch.autograd.grad(y, x, go, retain_graph=True)), 1),
          'nhwc', round(timeit(lambda: torch.autograd.grad(yl, xl, go, retain_graph=True)), 1), flush=True)
from sm3det_amd.roi_head import RotatedSingleRoIExtractor  # noqa: E402
ext = RotatedSingleRoIExtractor(dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True), 256, [4, 8, 16, 32])
lv = [torch.randn(2, 256, 256 >> i, 256 >> i, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
      for i in range(4)]
gr = torch.Generator().manual_seed(12)
rr = torch.zeros(1024, 6)
rr[:, 0] = torch.randint(0, 2, (1024,), generator=gr).float()
rr[:, 1:3] = torch.rand(1024, 2, generator=gr) * 1024
rr[:, 3] = torch.exp(torch.rand(1024, generator=gr) * 4.2 + 2.0)
rr[:, 4] = rr[:, 3] * (0.3 + torch.rand(1024, generator=gr))
rr[:, 5] = (torch.rand(1024, generator=gr) - 0.5) * 3.1
rr = rr.cuda()
ro = ext(lv, rr)
gro = torch.randn_like(ro)
print('extract fwd', round(timeit(lambda: ext(lv, rr)), 1), flush=True)
for mode in ('tiled', 'atomic'):
    os.environ['SM3_ROI_BWD'] = mode
    print('extract bwd', mode, round(timeit(lambda: torch.autograd.grad(ro, lv, gro, retain_graph=True), 5), 1), flush=True)
os.environ['SM3_ROI_BWD'] = 'tiled'
from sm3det_amd.mmcv_deform_conv import deform_conv2d  # noqa: E402
xd = torch.randn(2, 256, 128, 128, device='cuda')
od = torch.randn(2, 18, 128, 128, device='cuda') * 2
wd = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
for mode in ('1', '0'):
    os.environ['SM3_DEFORM_FUSED'] = mode
    print('deform fwd fused=' + mode, round(timeit(lambda: deform_conv2d(xd, od, wd, 1, 1, 1, 1, 1, False, 2), 5), 1), flush=True)
