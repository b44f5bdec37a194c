"""One isolated plain NT GEMM (16384x1536x384, the stage-2 expert shape) timed with HIP events -- calibration point
for the MFMA-utilisation PMC pass (scripts/collect_mfma_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sm3det_amd import _lib_backbone as LB
M, N, K = 16384, 1536, 384
A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
for _ in range(5): LB.gemm(LB.NT, A, B, C, M, N, K)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): LB.gemm(LB.NT, A, B, C, M, N, K)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f'NT {M}x{N}x{K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF/s')
