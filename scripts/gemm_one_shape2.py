"""One isolated GEMM launch series for PMC passes: python scripts/gemm_one_shape2.py nt 8192 1536 384 [epi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sm3det_amd import _lib_backbone as LB
mode, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
md = dict(nt=LB.NT, nn=LB.NN, tn=LB.TN)[mode]
sets = []
for _ in range(4):
    if mode == 'nt': A, B = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda') * 0.05
    elif mode == 'nn': A, B = torch.randn(M, K, device='cuda'), torch.randn(K, N, device='cuda') * 0.05
    else: A, B = torch.randn(K, M, device='cuda'), torch.randn(K, N, device='cuda')
    sets.append((A, B, torch.empty(M, N, device='cuda')))
for i in range(8):
    A, B, C = sets[i % 4]; LB.gemm(md, A, B, C, M, N, K)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    A, B, C = sets[i % 4]; LB.gemm(md, A, B, C, M, N, K)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f'{mode} {M}x{N}x{K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF/s')
