"""per-call timing of one MultitaskFPN forward+backward (bs 2 @ 1024^2 pyramid) through the C-ABI profiling hook"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sm3det_amd import _lib_backbone as LB
from sm3det_amd.fpn import MultitaskFPN
fpn = MultitaskFPN(in_channels=[96, 192, 384, 768], out_channels=256, extra_level=1, add_extra_convs='on_output', num_outs=5).cuda()
feats = [torch.randn(2, c, 256 >> i, 256 >> i, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True) for i, c in enumerate([96, 192, 384, 768])]
def step():
    for q in fpn.parameters(): q.grad = None
    sum((o * o).mean() for o in fpn(feats)).backward()
for _ in range(3): step()
torch.cuda.synchronize()
LB.PROFILE, LB.PROFILE_SHAPES = [], True
step(); torch.cuda.synchronize()
prof, LB.PROFILE = LB.PROFILE, None
tot = 0
for name, fl, nb, e0, e1 in prof:
    ms = e0.elapsed_time(e1); tot += ms
    print(f'{name:60s} {ms*1e3:9.1f} us  {fl/ms/1e9 if ms else 0:7.1f} TF')
print('total', tot)
import time
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('wall us per step', timeit(step))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
