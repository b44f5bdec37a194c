import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
from sm3det_amd import _lib_backbone as LB
net = bench.build_model(bench.DEFAULT_CONFIG).cuda().train()
seen = []
orig = LB.gemm
def gemm(mode, A, B, C, M, N, K, **kw):
    o = kw.get('offsets')
    if o is not None and mode == LB.TN and len(seen) < 40:
        seen.append((M, N, K, o.clone()))
    return orig(mode, A, B, C, M, N, K, **kw)
LB.gemm = gemm
import sm3det_amd.backbone_ops as ops
if hasattr(ops, 'gemm'): ops.gemm = gemm
x = torch.randn(2, 3, 1024, 1024).cuda()
outs, gl = net(x, ['single'])
(sum(o.sum() for o in outs) + gl).backward()
torch.cuda.synchronize()
for M, N, K, o in seen[::2]:
    c = (o[1:] - o[:-1]).tolist()
    print(M, N, K, c, 'max/mean', round(max(c) / (sum(c) / len(c)), 3))
