import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sm3det_amd import _lib_backbone as LB
M, N, K = 16384, 1536, 384
A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
for _ in range(5): LB.gemm(LB.NT, A, B, C, M, N, K)
torch.cuda.synchronize()
