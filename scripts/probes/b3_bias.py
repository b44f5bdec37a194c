"""Is the bf16x3 form biased?  Mean SIGNED error of C = A . B^T against the fp64 product, in units of the fp32 ulp of each
output, for the native fp32 MFMA form and the bf16x3 form, on zero-mean and on positive operands (K = 384 and 3072).
A coherent bias matters for what the training step reduces over ~1e5 tokens (bias gradients)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sm3det_amd import _lib_backbone as LB

def ulp(x):
    return torch.ldexp(torch.ones_like(x), torch.frexp(x.abs().clamp_min(1e-30))[1] - 24)

for K in (384, 3072):
    for name, mk in (('zero-mean', lambda *s: torch.randn(*s, device='cuda')),
                     ('positive', lambda *s: torch.rand(*s, device='cuda') + 0.5),
                     ('negative-A', lambda *s: -(torch.rand(*s, device='cuda') + 0.5))):
        torch.manual_seed(K)
        M, N = 4096, 384
        A = mk(M, K)
        B = (torch.rand(N, K, device='cuda') + 0.5) if name != 'zero-mean' else torch.randn(N, K, device='cuda')
        ref = A.double() @ B.double().t()
        out = []
        for ar in (0, 2):
            LB.ARITH32 = ar
            C = torch.empty(M, N, device='cuda')
            LB.gemm(LB.NT, A, B, C, M, N, K)
            e = (C.double() - ref) / ulp(ref.float()).double()
            # what a bias / LayerNorm gradient does with such an output: the sum over the token rows, in units of the
            # quadrature sum of the rows' ulps (a coherent -0.1 ulp bias over 4096 rows shows as -0.1 * sqrt(4096) = -6.4)
            cs = (C.double().sum(0) - ref.sum(0)) / ulp(ref.float()).double().pow(2).sum(0).sqrt()
            out.append((float(e.mean()), float(e.abs().mean()), float(e.abs().max()), float(cs.mean()), float(cs.abs().max())))
        print(f'K={K:5d} {name:11s} native: mean {out[0][0]:+.3f} |mean| {out[0][1]:.3f} max {out[0][2]:.2f} ulp colsum mean {out[0][3]:+.2f} max {out[0][4]:.2f}   '
              f'bf16x3: mean {out[1][0]:+.3f} |mean| {out[1][1]:.3f} max {out[1][2]:.2f} ulp colsum mean {out[1][3]:+.2f} max {out[1][4]:.2f}')
