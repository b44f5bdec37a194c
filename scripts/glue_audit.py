"""Which Python lines still launch torch-native (non-sm3det) kernels in one eager training step?  torch.profiler with
stacks; prints aten op -> count -> innermost repo frame."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

net = bench.build_model().cuda().train()
params = [p for p in net.parameters() if p.requires_grad]
from sm3det_amd.data_parallel import BucketedGradReducer
from sm3det_amd.optim import MultiTensorAdamW
red = BucketedGradReducer(params)
opt = MultiTensorAdamW([dict(params=[p]) for p in params], lr=1e-4, weight_decay=0.05, max_grad_norm=35.0)
x = torch.randn(bench.BATCH, 3, bench.RES, bench.RES).cuda()
proj = None
def step():
    global proj
    red.zero_grad()
    outs, gl = net(x, ['single'])
    if proj is None:
        proj = [torch.randn_like(o) for o in outs]
    loss = bench.loss_fn(outs, gl, proj)
    loss.backward(); red.finalize(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != 'CPU' or not ev.name.startswith('aten::'):
        continue
    if not getattr(ev, 'kernels', None):
        continue
    frame = next((s for s in ev.stack if '/sm3det_amd/' in s or 'bench.py' in s or 'scripts/' in s), ev.stack[0] if ev.stack else '?')
    cnt[(ev.name, frame.strip()[-90:])] += len(ev.kernels)
for (name, frame), n in cnt.most_common(45):
    print(f'{n:5d} {name:34s} {frame}')
