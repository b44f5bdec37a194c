"""FPN timings on the GPU box: the three implicit-GEMM 3x3 kernels at the stride-4 level of a 1024^2 batch, and the
whole MultitaskFPN forward+backward (main_SM3Det.py neck, start_level 0) on backbone-shaped inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sm3det_amd import _lib, _lib_backbone as LB
from sm3det_amd.fpn import MultitaskFPN

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (B, H, W, Cin, Cout, s) in [(2, 256, 256, 256, 256, 1), (2, 128, 128, 256, 256, 1), (2, 64, 64, 256, 256, 1), (2, 32, 32, 256, 256, 2)]:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, H, W, Cin, device='cuda'); w = torch.randn(Cout, 3, 3, Cin, device='cuda') * 0.02
    b = torch.randn(Cout, device='cuda'); y = torch.empty(B, Ho, Wo, Cout, device='cuda')
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    nb = _lib.lib().sm3_conv3x3_nhwc_bwd_weight_workspace_bytes(B, H, W, Cin, Cout, s); ws = _lib.workspace(nb, x.device)
    fl = 2.0 * B * Ho * Wo * Cout * 9 * Cin
    n1 = _lib.lib().sm3_conv3x3_nhwc_workspace_bytes(B, H, W, Cin, Cout, s, 0); w1 = _lib.workspace(n1, x.device)
    n2 = _lib.lib().sm3_conv3x3_nhwc_workspace_bytes(B, H, W, Cin, Cout, s, 1); w2 = _lib.workspace(n2, x.device)
    t1 = timeit(lambda: LB.call('conv3x3_nhwc_fwd', x, w, b, y, B, H, W, Cin, Cout, s, 0, w1, n1))
    t2 = timeit(lambda: LB.call('conv3x3_nhwc_bwd_input', y, w, dx, B, H, W, Cin, Cout, s, w2, n2))
    t3 = timeit(lambda: LB.call('conv3x3_nhwc_bwd_weight', x, y, dw, B, H, W, Cin, Cout, s, ws, nb))
    print(f'conv3x3 {B}x{H}x{W} {Cin}->{Cout} s{s}: fwd {t1*1e3:7.1f} us {fl/t1/1e9:6.1f} TF/s | dgrad {t2*1e3:7.1f} us {fl/t2/1e9:6.1f} | wgrad {t3*1e3:7.1f} us {fl/t3/1e9:6.1f}', flush=True)

net = MultitaskFPN(in_channels=[96, 192, 384, 768], out_channels=256, extra_level=1, add_extra_convs='on_output', num_outs=5).cuda()
xs = [torch.randn(2, c, 256 >> i, 256 >> i, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
      for i, c in enumerate([96, 192, 384, 768])]
def step():
    for p in net.parameters(): p.grad = None
    outs = net(xs)
    sum((o * o).mean() for o in outs).backward()
print(f'MultitaskFPN fwd+bwd, bs2 @1024^2 pyramid, start_level 0: {timeit(step):.2f} ms', flush=True)
def fwd():
    with torch.no_grad(): net(xs)
print(f'MultitaskFPN fwd only: {timeit(fwd):.2f} ms', flush=True)
