import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sm3det_amd import _lib_backbone as LB
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, C) in [(2, 256, 96), (2, 128, 192), (2, 64, 384), (2, 32, 768)]:
    W = H; T = B * H * W
    x = torch.randn(T, C, device='cuda'); w49 = torch.randn(49, C, device='cuda'); b = torch.randn(C, device='cuda')
    y = torch.empty_like(x); du = torch.randn_like(x); dw = torch.empty(49, C, device='cuda'); db = torch.empty(C, device='cuda')
    lw = torch.ones(C, device='cuda'); mean = torch.empty(T, device='cuda'); rstd = torch.empty(T, device='cuda')
    f = t(lambda: LB.call('dwconv7_fwd', x, w49, b, None, y, B, H, W, C, 0))
    fa = t(lambda: LB.call('dwconv7_fwd', x, w49, None, du, y, B, H, W, C, 1))
    bw = t(lambda: LB.call('dwconv7_bwd_weight', x, du, dw, db, B, H, W, C))
    ln = t(lambda: LB.call('layernorm_fwd', x, lw, b, 1e-6, y, mean, rstd, T, C, 0, H, W))
    mb = T * C * 4 / 1e6
    print(f'C={C} T={T} ({mb:.0f} MB/pass): dw fwd {f:.1f} us ({2*mb/f*1e-3*1e3/1e3:.2f} TB/s) | dw+addend {fa:.1f} us ({3*mb/fa:.2f} TB/s) | bwd_w {bw:.1f} us ({2*mb/bw:.2f} TB/s) | ln fwd {ln:.1f} us ({2*mb/ln:.2f} TB/s)')
