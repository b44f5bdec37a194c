"""Decompose the cost of the fused epilogues / expert grouping on the stage-2 expert GEMM (16384x1536x384)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sm3det_amd import _lib_backbone as LB

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (M, N, K) in [(16384, 1536, 384), (131072, 384, 96), (65536, 768, 192)]:
    A = torch.randn(M, K, device='cuda'); C = torch.empty(M, N, device='cuda'); aux = torch.empty(M, N, device='cuda')
    W = torch.randn(8, N, K, device='cuda') * 0.05; b = torch.randn(8, N, device='cuda')
    off = torch.arange(0, M + 1, M // 8, dtype=torch.int32, device='cuda')
    skew = torch.tensor([0, M // 16, M // 4, M // 4 + M // 32, M // 2, M // 2 + M // 8, M - M // 8, M - M // 16, M],
                        dtype=torch.int32, device='cuda')
    fl = 2.0 * M * N * K
    cases = {
        'plain': lambda: LB.gemm(LB.NT, A, W[0], C, M, N, K),
        'bias': lambda: LB.gemm(LB.NT, A, W[0], C, M, N, K, epilogue=LB.EPI_BIAS, bias=b[0]),
        'bias_gelu+aux': lambda: LB.gemm(LB.NT, A, W[0], C, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=b[0], aux_out=aux),
        'g8 plain': lambda: LB.gemm(LB.NT, A, W, C, M, N, K, offsets=off, num_groups=8),
        'g8 bias_gelu+aux': lambda: LB.gemm(LB.NT, A, W, C, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=b, aux_out=aux,
                                            offsets=off, num_groups=8),
        'g8 skewed bias_gelu+aux': lambda: LB.gemm(LB.NT, A, W, C, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=b,
                                                   aux_out=aux, offsets=skew, num_groups=8),
    }
    for name, fn in cases.items():
        us = timeit(fn)
        print(f'{M}x{N}x{K} {name:26s} {us:8.1f} us {fl / us / 1e6:6.1f} TF/s', flush=True)
