"""Sweep (tile, k-step, split-K) over the GEMM shapes of one training step (bs 2 @ 1024^2, ConvNeXt-T e8t2) on the
GPU box; prints per shape the default configuration's time and every alternative, best first.

    python scripts/gemm_sweep2.py [--quick]  > gpurun_out/gemm_sweep2.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sm3det_amd import _lib_backbone as LB  # noqa: E402

E = 8


def tune(tile=None, bk=None, splits=0, separate=False, staged=False):  # staged: historical A/B (always on now)
    # bit 16 = TN in-kernel fix-up; `separate` (the default path) is the absence of it
    return ((tile + 1) if tile is not None else 0) | ({None: 0, 16: 1, 32: 2}[bk] << 4) | (splits << 8) | (int(not separate) << 16)


# (mode, M, N, K, groups, epilogue, count per step)
SHAPES = [
    ('nt', 131072, 384, 96, 1, LB.EPI_BIAS_GELU, 3), ('nt', 131072, 96, 384, 1, LB.EPI_BIAS_SCALE_RES, 3),
    ('nn', 131072, 384, 96, 1, LB.EPI_GELU_BWD, 3), ('nn', 131072, 96, 384, 1, LB.EPI_NONE, 3),
    ('tn', 384, 96, 131072, 1, 0, 3), ('tn', 96, 384, 131072, 1, 0, 3),
    ('nt', 32768, 768, 192, 1, LB.EPI_BIAS_GELU, 1), ('nt', 32768, 192, 768, 1, LB.EPI_BIAS_SCALE_RES, 1),
    ('nn', 32768, 768, 192, 1, LB.EPI_GELU_BWD, 1), ('nn', 32768, 192, 768, 1, LB.EPI_NONE, 1),
    ('tn', 768, 192, 32768, 1, 0, 1), ('tn', 192, 768, 32768, 1, 0, 1),
    ('nt', 65536, 768, 192, E, LB.EPI_BIAS_GELU, 2), ('nt', 65536, 192, 768, E, LB.EPI_BIAS, 2),
    ('nn', 65536, 768, 192, E, LB.EPI_GELU_BWD, 2), ('nn', 65536, 192, 768, E, LB.EPI_NONE, 2),
    ('tn', 768, 192, 65536, E, 0, 2), ('tn', 192, 768, 65536, E, 0, 2),
    ('nt', 8192, 1536, 384, 1, LB.EPI_BIAS_GELU, 4), ('nt', 8192, 384, 1536, 1, LB.EPI_BIAS_SCALE_RES, 4),
    ('nn', 8192, 1536, 384, 1, LB.EPI_GELU_BWD, 4), ('nn', 8192, 384, 1536, 1, LB.EPI_NONE, 4),
    ('tn', 1536, 384, 8192, 1, 0, 4), ('tn', 384, 1536, 8192, 1, 0, 4),
    ('nt', 16384, 1536, 384, E, LB.EPI_BIAS_GELU, 5), ('nt', 16384, 384, 1536, E, LB.EPI_BIAS, 5),
    ('nn', 16384, 1536, 384, E, LB.EPI_GELU_BWD, 5), ('nn', 16384, 384, 1536, E, LB.EPI_NONE, 5),
    ('tn', 1536, 384, 16384, E, 0, 5), ('tn', 384, 1536, 16384, E, 0, 5),
    ('nt', 2048, 3072, 768, 1, LB.EPI_BIAS_GELU, 1), ('nt', 2048, 768, 3072, 1, LB.EPI_BIAS_SCALE_RES, 1),
    ('nn', 2048, 3072, 768, 1, LB.EPI_GELU_BWD, 1), ('nn', 2048, 768, 3072, 1, LB.EPI_NONE, 1),
    ('tn', 3072, 768, 2048, 1, 0, 1), ('tn', 768, 3072, 2048, 1, 0, 1),
    ('nt', 4096, 3072, 768, E, LB.EPI_BIAS_GELU, 2), ('nt', 4096, 768, 3072, E, LB.EPI_BIAS, 2),
    ('nn', 4096, 3072, 768, E, LB.EPI_GELU_BWD, 2), ('nn', 4096, 768, 3072, E, LB.EPI_NONE, 2),
    ('tn', 3072, 768, 4096, E, 0, 2), ('tn', 768, 3072, 4096, E, 0, 2),
    ('nt', 8192, 224, 384, 1, LB.EPI_BIAS, 5), ('nn', 8192, 384, 224, 1, LB.EPI_NONE, 5), ('tn', 224, 384, 8192, 1, 0, 5),
    ('nt', 32768, 128, 192, 1, LB.EPI_BIAS, 2), ('nt', 2048, 288, 768, 1, LB.EPI_BIAS, 2),
]


def candidates(mode, M, N, K, quick):
    out = [dict()]
    if '--default-only' in sys.argv:
        return out
    if '--staged' in sys.argv:  # A/B of the LDS-staged epilogue on the default configuration and its neighbours
        if mode == 'tn':
            return [dict(separate=True), dict(separate=True, staged=True)]
        for t in (0, 1, 5):
            for bk in (16, 32):
                out.append(dict(tile=t, bk=bk, splits=1))
                out.append(dict(tile=t, bk=bk, splits=1, staged=True))
        return out + [dict(staged=True)]
    if mode == 'tn':
        for t in (0, 1, 2, 3, 4):
            for bk in ((16, 32) if t < 3 else (16,)):
                for s in (1, 2, 3, 4, 6, 8):
                    out.append(dict(tile=t, bk=bk, splits=s))
                for s in (2, 3, 4, 5, 6, 8, 10, 12, 16, 21, 24, 32, 48, 64, 96, 128, 192, 256, 341):
                    out.append(dict(tile=t, bk=bk, splits=s, separate=True))
    else:
        for t in (0, 1, 3, 5):
            for bk in ((16, 32) if t != 3 else (16,)):
                for s in (1, 2, 3, 4, 6, 8):
                    out.append(dict(tile=t, bk=bk, splits=s))
    if quick:
        out = out[:8]
    return out


def main():
    quick = '--quick' in sys.argv
    dev = torch.device('cuda')
    for mode, M, N, K, G, epi, cnt in SHAPES:
        if '--tn-only' in sys.argv and mode != 'tn':
            continue
        if '--aux-only' in sys.argv and epi not in (LB.EPI_BIAS_SCALE_RES, LB.EPI_GELU_BWD):
            continue
        rows = K if mode == 'tn' else M
        offs = None
        if G > 1:  # mildly ragged expert loads
            frac = np.array([1.3, 0.7, 1.1, 0.9, 1.0, 1.0, 1.2, 0.8]) / 8.0
            c = (frac * rows).astype(np.int64)
            c[-1] += rows - c.sum()
            offs = torch.tensor(np.concatenate([[0], np.cumsum(c)]), dtype=torch.int32, device=dev)
        if mode == 'nt':
            A, B = torch.randn(M, K, device=dev), torch.randn(G, N, K, device=dev) * 0.05
        elif mode == 'nn':
            A, B = torch.randn(M, K, device=dev), torch.randn(G, K, N, device=dev) * 0.05
        else:
            A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
        C = torch.empty((G, M, N) if mode == 'tn' else (M, N), device=dev)
        bias = torch.randn(G, N, device=dev)
        aux1, aux2 = torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
        gamma = torch.randn(N, device=dev)
        cs = torch.empty(G, N, device=dev)
        kw = dict(offsets=offs, num_groups=G)
        if epi == LB.EPI_BIAS_GELU:
            kw.update(epilogue=epi, bias=bias, aux_out=aux2)
        elif epi == LB.EPI_BIAS_SCALE_RES:
            kw.update(epilogue=epi, bias=bias, aux_in=aux1, aux_out=aux2, gamma=gamma)
        elif epi == LB.EPI_GELU_BWD:
            kw.update(epilogue=epi, aux_in=aux1, colsum_out=cs)
        elif epi == LB.EPI_BIAS:
            kw.update(epilogue=epi, bias=bias)
        md = dict(nt=LB.NT, nn=LB.NN, tn=LB.TN)[mode]
        cands = candidates(mode, M, N, K, quick)
        if not cands:
            continue
        best_t = [float('inf')] * len(cands)
        errs = {}
        for rnd in range(2):  # two interleaved rounds, keep the minimum: clock / cache drift hits all candidates alike
            for ci, cand in enumerate(cands):
                if ci in errs:
                    continue
                LB.TUNING = tune(**cand)
                try:
                    for _ in range(2):
                        LB.gemm(md, A, B, C, M, N, K, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(6):
                        LB.gemm(md, A, B, C, M, N, K, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    best_t[ci] = min(best_t[ci], e0.elapsed_time(e1) / 6 * 1e3)
                except Exception as ex:  # noqa: BLE001
                    errs[ci] = str(ex)[:40]
                LB.TUNING = 0
        res = [(best_t[ci], cands[ci]) for ci in range(len(cands)) if ci not in errs]
        fl = 2.0 * M * N * K
        d_us = res[0][0]
        res.sort(key=lambda r: r[0])
        print(f'{mode} {M}x{N}x{K} g{G} e{epi} x{cnt}: default {d_us:.1f} us ({fl / d_us / 1e6:.1f} TF)  best {res[0][0]:.1f} us '
              f'({fl / res[0][0] / 1e6:.1f} TF) {res[0][1]}', flush=True)
        for t, c in res[1:10]:
            print(f'      {t:.1f} us {c}')
        import json
        print('JSON ' + json.dumps(dict(mode=mode, M=M, N=N, K=K, G=G, epi=epi, count=cnt, default_us=d_us, results=[[t, c] for t, c in res])), flush=True)
    print('done')


if __name__ == '__main__':
    main()
