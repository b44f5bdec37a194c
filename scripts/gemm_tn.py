"""TN (wgrad) calibration: main-loop rate vs split-K factor, grouped vs not."""
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

for (M, N, K, G) in [(384, 1536, 16384, 1), (1536, 384, 16384, 1), (384, 1536, 16384, 8), (384, 96, 131072, 1), (96, 384, 131072, 1),
                     (192, 768, 65536, 8), (1536, 1536, 16384, 1)]:
    A = torch.randn(K, M, device='cuda'); B = torch.randn(K, N, device='cuda')
    C = torch.empty(G, M, N, device='cuda')
    off = torch.arange(0, K + 1, K // G, dtype=torch.int32, device='cuda') if G > 1 else None
    fl = 2.0 * M * N * K
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * G
    auto = LB.tn_splits(tiles, K // G)
    for s in sorted(set([1, 2, 3, 4, 8, 16, 32, auto])):
        if tiles * s > 8192: continue
        us = timeit(lambda: LB.gemm(LB.TN, A, B, C, M, N, K, offsets=off, num_groups=G, splits=s))
        print(f'TN {M}x{N}x{K} g{G} tiles={tiles} splits={s:3d}{"*" if s == auto else " "} {us:8.1f} us {fl / us / 1e6:6.1f} TF/s', flush=True)
