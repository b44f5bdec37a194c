"""Run the hot path (b) operators at their SURVEY.md 8(d) shapes a few times each (for rocprofv3: kernel-trace stats and
PMC passes, scripts/gpu_r03_ops_profile.sh) -- one operator per process so that kernel names map to operators.

    python scripts/ops_profile.py <op> [reps]      op in OPS below
"""
import os
import sys

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sm3det_amd import mmcv_ops as ops  # noqa: E402
from tests import synth  # noqa: E402

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731


def make(op):
    if op in ('box_iou_rotated_2000x512', 'box_iou_rotated_2000x64'):
        m = int(op.split('x')[-1])
        b1, b2 = dev(synth.rotated_boxes(2000, 0)), dev(synth.rotated_boxes(m, 1))
        return lambda: ops.box_iou_rotated(b1, b2)
    if op in ('nms_rotated_2000', 'nms_rotated_10000'):
        n = int(op.split('_')[-1])
        d = dev(synth.rotated_boxes(n, 2 if n == 2000 else 7, cluster=(n == 2000)))
        s = dev(synth.unique_scores(n, 3 if n == 2000 else 8))
        return lambda: ops.nms_rotated(d, s, 0.1)
    if op == 'nms_8768':
        hb, hs = dev(synth.hboxes(8768, 4, cluster=True)), dev(synth.unique_scores(8768, 5))
        return lambda: ops.nms(hb, hs, iou_threshold=0.8)
    if op.startswith('roi_align_rotated'):
        x = torch.randn(1, 256, 256, 256, device='cuda')
        if op.endswith('nhwc'):
            x = x.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        rois = dev(synth.rois_for_level(512, 6, batch=1, extent=1024.0))
        layer = ops.RoIAlignRotated(output_size=7, spatial_scale=0.25, sampling_ratio=2, clockwise=True)
        if '_fwd_' in op:
            return lambda: layer(x, rois)
        y = layer(x, rois)
        go = torch.randn_like(y)
        return lambda: torch.autograd.grad(y, x, go, retain_graph=True)
    if op.startswith('deform_conv2d'):
        from sm3det_amd.mmcv_deform_conv import deform_conv2d
        xd = torch.randn(2, 256, 128, 128, device='cuda', requires_grad=True)
        od = (torch.randn(2, 18, 128, 128, device='cuda') * 2).requires_grad_(True)
        wd = (torch.randn(256, 256, 3, 3, device='cuda') * 0.02).requires_grad_(True)
        if op.endswith('fwd'):
            return lambda: deform_conv2d(xd, od, wd, 1, 1, 1, 1, 1, False, 2)
        yd = deform_conv2d(xd, od, wd, 1, 1, 1, 1, 1, False, 2)
        gd = torch.randn_like(yd)
        return lambda: torch.autograd.grad(yd, (xd, od, wd), gd, retain_graph=True)
    raise SystemExit(f'unknown op {op}')


OPS = ['box_iou_rotated_2000x512', 'box_iou_rotated_2000x64', 'nms_rotated_2000', 'nms_rotated_10000', 'nms_8768',
       'roi_align_rotated_fwd_nchw', 'roi_align_rotated_fwd_nhwc', 'roi_align_rotated_bwd_nchw',
       'roi_align_rotated_bwd_nhwc', 'deform_conv2d_fwd', 'deform_conv2d_bwd']

if __name__ == '__main__':
    op = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    fn = make(op)
    fn()
    torch.cuda.synchronize()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print('done', op, reps)
