"""bf16x3 GEMMs with operands as bf16x3 PLANES vs fp32 operands on the NT / NN shapes of one training step: bit-equality of
the outputs and cold-operand timings (six rotating argument sets) for A planes, B planes, both.

    python scripts/gemm_pl_eval.py > gpurun_out/gemm_pl_eval.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from scripts.gemm_sweep2 import SHAPES  # noqa: E402
from sm3det_amd import _lib_backbone as LB  # noqa: E402

ROT = 6


def offsets_for(rows, G, dev):
    if G <= 1:
        return None
    frac = np.array([1.3, 0.7, 1.1, 0.9, 1.0, 1.0, 1.2, 0.8]) / 8.0
    c = (frac * rows).astype(np.int64)
    c[-1] += rows - c.sum()
    return torch.tensor(np.concatenate([[0], np.cumsum(c)]), dtype=torch.int32, device=dev)


def main():
    dev = torch.device('cuda')
    LB.ARITH32 = 2
    tot = dict(f32=0.0, a=0.0, b=0.0, ab=0.0, best=0.0)
    print('# mode M N K G epi cnt | us fp32 operands | A planes | B planes | both | bit-equal (A, B, both)')
    for mode, M, N, K, G, epi, cnt in SHAPES:
        if mode == 'tn':
            continue
        offs = offsets_for(M, G, dev)
        md = dict(nt=LB.NT, nn=LB.NN)[mode]
        sets = []
        for _ in range(ROT):
            A = torch.randn(M, K, device=dev)
            B = torch.randn(G, N, K, device=dev) * 0.05 if mode == 'nt' else torch.randn(G, K, N, device=dev) * 0.05
            C = torch.empty(M, N, device=dev)
            kw = dict(offsets=offs, num_groups=G)
            if epi == LB.EPI_BIAS_GELU:
                kw.update(epilogue=epi, bias=torch.randn(G, N, device=dev), aux_out=torch.empty(M, N, device=dev))
            elif epi == LB.EPI_BIAS_SCALE_RES:
                kw.update(epilogue=epi, bias=torch.randn(G, N, device=dev), aux_in=torch.randn(M, N, device=dev),
                          aux_out=torch.empty(M, N, device=dev), gamma=torch.randn(N, device=dev))
            elif epi == LB.EPI_GELU_BWD:
                kw.update(epilogue=epi, aux_in=torch.randn(M, N, device=dev), colsum_out=torch.empty(G, N, device=dev))
            elif epi == LB.EPI_BIAS:
                kw.update(epilogue=epi, bias=torch.randn(G, N, device=dev))
            Ap = LB.planes(A)
            Bp = torch.empty(3, K // 8, G * N, 8, dtype=torch.bfloat16, device=dev)
            for g in range(G):
                LB.planes(B[g], transpose=(mode == 'nn'), out=Bp, row_off=g * N)
            sets.append((A, B, C, kw, Ap, Bp))

        def run(s, variant):
            A, B, C, kw, Ap, Bp = s
            LB.gemm(md, Ap if variant in ('a', 'ab') else A, Bp if variant in ('b', 'ab') else B, C, M, N, K, **kw)

        # bit-equality on set 0
        ref = None
        eq = []
        for v in ('f32', 'a', 'b', 'ab'):
            sets[0][2].fill_(float('nan'))
            try:
                run(sets[0], v)
            except Exception as ex:  # noqa: BLE001 (an unsupported epilogue / io pair)
                eq.append('n/a')
                continue
            out = sets[0][2].clone()
            if v == 'f32':
                ref = out
            else:
                eq.append(str(bool(torch.equal(out, ref))))

        def timed(v):
            try:
                for r_ in range(2):
                    run(sets[r_ % ROT], v)
            except Exception:  # noqa: BLE001
                return float('nan')
            best = float('inf')
            for _rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for r_ in range(6):
                    run(sets[(r_ + 2) % ROT], v)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 6 * 1e3)
            return best
        t = {v: timed(v) for v in ('f32', 'a', 'b', 'ab')}
        t2 = {v: timed(v) for v in ('f32', 'a', 'b', 'ab')}  # second pass: order effects
        t = {v: min(t[v], t2[v]) if t[v] == t[v] else t[v] for v in t}
        for v in t:
            if t[v] == t[v]:
                tot[v] += cnt * t[v]
            else:
                tot[v] += cnt * t['f32']
        tot['best'] += cnt * min(x for x in t.values() if x == x)
        print(f'{mode} {M} {N} {K} g{G} e{epi} x{cnt} | {t["f32"]:.1f} | {t["a"]:.1f} | {t["b"]:.1f} | {t["ab"]:.1f} | {" ".join(eq)}',
              flush=True)
        del sets
        torch.cuda.empty_cache()
    print('# NT + NN launches of the step, count-weighted, cold operands (unsupported variants counted at the fp32-operand time): '
          + ', '.join(f'{k} {v / 1e3:.3f} ms' for k, v in tot.items()))


if __name__ == '__main__':
    main()
