"""Sweep (tile, k-step) over the GEMM shapes of one AMP training step (bs 2 @ 1024^2, ConvNeXt-T e8t2) with the tensors the
AMP data path stores as fp16 in fp16 (gemm_h16.hip); prints per shape the default configuration's time and every
alternative, best first, plus the step total of the defaults and of the per-shape best.

    python scripts/gemm_sweep_amp.py [--default-only] [--cold]  > gpurun_out/gemm_sweep_amp.txt
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from scripts.gemm_sweep2 import SHAPES, E, tune as _tune  # noqa: E402
from sm3det_amd import _lib_backbone as LB  # noqa: E402
from sm3det_amd import amp  # noqa: E402
import numpy as np  # noqa: E402

BK = {None: 0, 16: 1, 32: 2, 64: 3}
# --cold: six rotating argument sets per shape, so that no launch finds its operands in the 256 MB MALL / the L2s
ROT = 6 if '--cold' in sys.argv else 1
FP32 = '--fp32' in sys.argv  # the same sweep over the fp32 GEMM family (all tensors fp32, no autocast)


PERSIST = (1 << 17) if '--persist' in sys.argv else 0  # the opt-in persistent form of the NT / NN launches


def tune(tile=None, bk=None, splits=0):
    return ((tile + 1) if tile is not None else 0) | (BK[bk] << 4) | (splits << 8) | PERSIST


def candidates(mode):
    out = [dict()]
    if '--default-only' in sys.argv:
        return out
    tiles = ((0, 1, 2, 3, 4) if FP32 else (0, 1, 2)) if mode == 'tn' else ((0, 1, 3, 5) if FP32 else (0, 1, 5))
    for bk in ((32, 16) if FP32 else (32, 64, 16)):
        out.append(dict(bk=bk))
    for t in tiles:
        for bk in ((16, 32) if FP32 else (32, 64)):
            if FP32 and t in (3, 4) and bk == 32:
                continue
            out.append(dict(tile=t, bk=bk))
            if '--splits' in sys.argv:
                for sp in ((2, 3, 4, 6, 8) if mode != 'tn' else (1, 2, 4, 8, 16, 32, 64)):
                    out.append(dict(tile=t, bk=bk, splits=sp))
    return out


def main():
    dev = torch.device('cuda')
    h = torch.float32 if FP32 else torch.float16
    tot_def = tot_best = 0.0
    import contextlib
    with (contextlib.nullcontext() if FP32 else amp.autocast()):
        for mode, M, N, K, G, epi, cnt in SHAPES:
            rows = K if mode == 'tn' else M
            offs = None
            if G > 1:
                frac = np.array([1.3, 0.7, 1.1, 0.9, 1.0, 1.0, 1.2, 0.8]) / 8.0
                c = (frac * rows).astype(np.int64)
                c[-1] += rows - c.sum()
                offs = torch.tensor(np.concatenate([[0], np.cumsum(c)]), dtype=torch.int32, device=dev)
            sets = []
            for _rep in range(ROT):
                bias = torch.randn(G, N, device=dev)
                gamma = torch.randn(N, device=dev)
                cs = torch.empty(G, N, device=dev)
                kw = dict(offsets=offs, num_groups=G)
                if mode == 'nt':
                    A, B = torch.randn(M, K, device=dev).to(h), torch.randn(G, N, K, device=dev) * 0.05
                    if epi == LB.EPI_BIAS_GELU:  # FC1: act and GELU' stored as halves
                        C = torch.empty(M, N, device=dev, dtype=h)
                        kw.update(epilogue=epi, bias=bias, aux_out=torch.empty(M, N, device=dev, dtype=h))
                    elif epi == LB.EPI_BIAS_SCALE_RES:
                        C = torch.empty(M, N, device=dev)
                        kw.update(epilogue=epi, bias=bias, aux_in=torch.randn(M, N, device=dev), aux_out=torch.empty(M, N, device=dev),
                                  gamma=gamma)
                    else:
                        C = torch.empty(M, N, device=dev)
                        kw.update(epilogue=epi, bias=bias)
                elif mode == 'nn':
                    B = torch.randn(G, K, N, device=dev) * 0.05
                    if epi == LB.EPI_GELU_BWD:  # dy (fp32) . W2 x GELU' (half) -> dh (half)
                        A = torch.randn(M, K, device=dev)
                        C = torch.empty(M, N, device=dev, dtype=h)
                        kw.update(epilogue=epi, aux_in=torch.randn(M, N, device=dev).to(h), colsum_out=cs)
                    else:                       # dh (half) . W1 -> fp32
                        A = torch.randn(M, K, device=dev).to(h)
                        C = torch.empty(M, N, device=dev)
                else:
                    # weight gradients: dh^T x (both halves) when the output is 4C x C, dy^T act (fp32, half) when C x 4C
                    A = torch.randn(K, M, device=dev)
                    if M > N:
                        A = A.to(h)
                    B = torch.randn(K, N, device=dev).to(h)
                    C = torch.empty(G, M, N, device=dev)
                sets.append((A, B, C, dict(kw)))
            md = dict(nt=LB.NT, nn=LB.NN, tn=LB.TN)[mode]
            cands = candidates(mode)
            best_t = [float('inf')] * len(cands)
            errs = {}
            for rnd in range(2):
                for ci, cand in enumerate(cands):
                    if ci in errs:
                        continue
                    LB.TUNING = tune(**cand)
                    try:
                        for r_ in range(2):
                            A, B, C, kw = sets[r_ % ROT]
                            LB.gemm(md, A, B, C, M, N, K, **kw)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for r_ in range(6):
                            A, B, C, kw = sets[(r_ + 2) % ROT]
                            LB.gemm(md, A, B, C, M, N, K, **kw)
                        e1.record()
                        torch.cuda.synchronize()
                        best_t[ci] = min(best_t[ci], e0.elapsed_time(e1) / 6 * 1e3)
                    except Exception as ex:  # noqa: BLE001
                        errs[ci] = str(ex)[:60]
                    LB.TUNING = 0
            res = [(best_t[ci], cands[ci]) for ci in range(len(cands)) if ci not in errs]
            fl = 2.0 * M * N * K
            d_us = res[0][0]
            res.sort(key=lambda r: r[0])
            tot_def += d_us * cnt
            tot_best += res[0][0] * cnt
            print(f'{mode} {M}x{N}x{K} g{G} e{epi} x{cnt}: default {d_us:.1f} us ({fl / d_us / 1e6:.0f} TF)  best {res[0][0]:.1f} us '
                  f'{res[0][1]}   ' + '  '.join(f'{t:.1f}:{c.get("tile", "-")}/{c.get("bk", "-")}/{c.get("splits", "-")}' for t, c in res[1:12]), flush=True)
            if errs:
                print('      errors:', errs)
            print('JSON ' + json.dumps(dict(mode=mode, M=M, N=N, K=K, G=G, epi=epi, count=cnt, default_us=d_us,
                                            results=[[t, c] for t, c in res])), flush=True)
    print(f'TOTAL per step (isolated launches): default {tot_def / 1e3:.3f} ms, per-shape best {tot_best / 1e3:.3f} ms')


if __name__ == '__main__':
    main()
