"""eager vs hipGraph-replayed training steps: loss trajectories must agree (same seeds -> same RNG stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sm3det_amd.data_parallel import BucketedGradReducer

def run(use_graph, n=6):
    torch.manual_seed(0); torch.cuda.manual_seed(0)
    net = bench.build_model().cuda().train()
    params = list(net.parameters())
    red = BucketedGradReducer(params)
    opt = torch.optim.AdamW(params, lr=1e-3, fused=True, capturable=True)
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(1)).cuda()
    proj = [None]
    def step():
        red.zero_grad()
        outs, gl = net(x, ['single'])
        if proj[0] is None:
            gp = torch.Generator().manual_seed(2)
            proj[0] = [torch.randn(o.shape, generator=gp).cuda() for o in outs]
        loss = bench.loss_fn(outs, gl, proj[0])
        loss.backward(); red.finalize(); opt.step()
        return loss
    losses = []
    if not use_graph:
        for _ in range(n + 2): losses.append(float(step()))
        return losses
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): losses.append(float(step()))
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        l = step()
    for _ in range(n):
        g.replay(); losses.append(float(l))
    return losses
a = run(False); b = run(True)
print('eager', [f'{v:.4f}' for v in a]); print('graph', [f'{v:.4f}' for v in b])
