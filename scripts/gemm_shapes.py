"""List every GEMM launch of one training step with its time and achieved TFLOP/s (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sm3det_amd import _lib_backbone as LB

net = bench.build_model().cuda().train()
x = torch.randn(2, 3, 1024, 1024, device='cuda')
def step():
    for p in net.parameters(): p.grad = None
    outs, gl = net(x, ['single'])
    (sum((o * o).mean() for o in outs) + gl).backward()
for _ in range(3): step()
torch.cuda.synchronize()
LB.PROFILE, LB.PROFILE_SHAPES = [], True
step(); torch.cuda.synchronize()
prof, LB.PROFILE = LB.PROFILE, None
agg = {}
for name, fl, nb, e0, e1 in prof:
    a = agg.setdefault(name, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(v[1] for k, v in agg.items() if k.startswith('gemm'))
print(f'total gemm ms {tot:.2f}')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    if k.startswith('gemm'):
        print(f'{k:48s} n={v[0]:3d} ms={v[1]:7.3f} TF={v[2]/v[1]/1e9 if v[1] else 0:6.1f}')
