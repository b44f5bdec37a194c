"""GEMM calibration sweep: TFLOP/s of sm3_gemm_f32 on square and hot-path shapes, BK=16 vs 32 (SM3_GEMM_BK env)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sm3det_amd import _lib_backbone as LB

def bench(mode, M, N, K, n=10, **kw):
    if mode == LB.NT:
        A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda')
    elif mode == LB.NN:
        A = torch.randn(M, K, device='cuda'); B = torch.randn(K, N, device='cuda')
    else:
        A = torch.randn(K, M, device='cuda'); B = torch.randn(K, N, device='cuda')
    C = torch.empty(M, N, device='cuda')
    for _ in range(3): LB.gemm(mode, A, B, C, M, N, K, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): LB.gemm(mode, A, B, C, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return 2.0 * M * N * K / ms / 1e9, ms

shapes = [(LB.NT, 4096, 4096, 4096), (LB.NT, 8192, 8192, 1024), (LB.NT, 16384, 1536, 384), (LB.NT, 16384, 384, 1536),
          (LB.NN, 16384, 1536, 384), (LB.NT, 131072, 384, 96), (LB.NT, 131072, 96, 384), (LB.NT, 65536, 768, 192),
          (LB.NT, 2048, 768, 3072), (LB.NT, 8192, 224, 384)]
# SM3_GEMM_BK / SM3_GEMM_KC are read once per process by the library: run this script once per setting
tag = f"BK={os.environ.get('SM3_GEMM_BK', 'auto')} KC={os.environ.get('SM3_GEMM_KC', 'default')}"
for mode, M, N, K in shapes:
    tf, ms = bench(mode, M, N, K)
    print(f'{tag} mode={mode} {M}x{N}x{K}: {ms*1e3:8.1f} us  {tf:6.1f} TF/s', flush=True)
