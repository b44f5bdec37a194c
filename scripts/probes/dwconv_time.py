"""Depthwise 7x7 forward / input-gradient / weight-gradient launch times on the four stage shapes of one training step
(ConvNeXt-T, bs 2 @ 1024^2), cold operands (four rotating argument sets).  SM3DET_HIP_LIB selects the library build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sm3det_amd import _lib_backbone as LB

ROT = 4
tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
for (B, H, W, C, nblk) in ((2, 256, 256, 96, 3), (2, 128, 128, 192, 3), (2, 64, 64, 384, 9), (2, 32, 32, 768, 3)):
    xs = [torch.randn(B, H, W, C, device='cuda') for _ in range(ROT)]
    ys = [torch.empty(B, H, W, C, device='cuda') for _ in range(ROT)]
    rs = [torch.randn(B, H, W, C, device='cuda') for _ in range(ROT)]
    w49, b = torch.randn(49, C, device='cuda'), torch.randn(C, device='cuda')
    dw, db = torch.empty(49, C, device='cuda'), torch.empty(C, device='cuda')
    runs = {'fwd': lambda i: LB.call('dwconv7_fwd', xs[i], w49, b, None, ys[i], B, H, W, C, 0),
            'dgrad': lambda i: LB.call('dwconv7_fwd', xs[i], w49, None, rs[i], ys[i], B, H, W, C, 1),
            'wgrad': lambda i: LB.call('dwconv7_bwd_weight', xs[i], rs[i], dw, db, B, H, W, C)}
    out = []
    for name, fn in runs.items():
        for i in range(ROT):
            fn(i)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(2 * ROT):
                fn(i % ROT)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (2 * ROT) * 1e3)
        tot[name] += nblk * best
        gb = B * H * W * C * 4 * (2 if name != 'dgrad' else 3) / best / 1e3
        out.append(f'{name} {best:6.1f} us {gb:6.0f} GB/s')
    print(f'{B}x{H}x{W}x{C} x{nblk}: ' + ' | '.join(out))
print('per step: ' + ' '.join(f'{k} {v / 1e3:.3f} ms' for k, v in tot.items()))
