"""bf16x3 vs native fp32 MFMA on the GEMM shapes of one training step (bs 2 @ 1024^2, ConvNeXt-T e8t2): per shape the error of
both arithmetics against the fp64 product of the same fp32 operands, and cold-operand timings (six rotating argument sets)
of the library's default configuration in both arithmetics plus every bf16x3 tile candidate.

    python scripts/gemm_b3_eval.py [--no-sweep] > gpurun_out/gemm_b3_eval.txt

Guardrail of the round-4 review (VERDICT Next #1a): per-shape error vs fp64 of the bf16x3 form <= 1.5 x the native kernel's.
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


def reference64(mode, A, B, offs, G):
    A64, B64 = A.double(), B.double()
    if mode == 'nt':
        if G == 1:
            return A64 @ B64[0].t()
        o = offs.tolist()
        return torch.cat([A64[o[g]:o[g + 1]] @ B64[g].t() for g in range(G)])
    if mode == 'nn':
        if G == 1:
            return A64 @ B64[0]
        o = offs.tolist()
        return torch.cat([A64[o[g]:o[g + 1]] @ B64[g] for g in range(G)])
    if G == 1:
        return (A64.t() @ B64)[None]
    o = offs.tolist()
    return torch.stack([A64[o[g]:o[g + 1]].t() @ B64[o[g]:o[g + 1]] for g in range(G)])


def run(md, A, B, C, M, N, K, kw, arith, tuning=0):
    LB.ARITH32, LB.TUNING = arith, tuning
    LB.gemm(md, A, B, C, M, N, K, **kw)


def main():
    dev = torch.device('cuda')
    sweep = '--no-sweep' not in sys.argv
    tot = {'f32': 0.0, 'b3': 0.0, 'b3best': 0.0}
    worst_ratio = 0.0
    print('# mode M N K G epi cnt | err_max/scale f32, b3, ratio | err_rms f32, b3, ratio | us f32, b3 (default cfg) | b3 tiles')
    for mode, M, N, K, G, epi, cnt in SHAPES:
        rows = K if mode == 'tn' else M
        offs = offsets_for(rows, G, dev)
        md = dict(nt=LB.NT, nn=LB.NN, tn=LB.TN)[mode]
        sets = []
        for _ in range(ROT):
            if mode == 'nt':
                A, B = torch.randn(M, K, device=dev), torch.randn(G, N, K, device=dev) * 0.05
            elif mode == 'nn':
                A, B = torch.randn(M, K, device=dev), torch.randn(G, K, N, device=dev) * 0.05
            else:
                A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
            C = torch.empty((G, M, N) if mode == 'tn' else (M, N), device=dev)
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
            sets.append((A, B, C, kw))
        # ---- error of the raw product (no epilogue) in both arithmetics
        A, B, C, _ = sets[0]
        ref = reference64(mode, A, B, offs, G)
        scale = ref.abs().max().item()
        rms = ref.pow(2).mean().sqrt().item()
        errs = {}
        for name, ar in (('f32', 0), ('b3', 2)):
            C.fill_(float('nan'))
            run(md, A, B, C, M, N, K, dict(offsets=offs, num_groups=G), ar)
            d = (C.double().reshape(ref.shape) - ref)
            errs[name] = (d.abs().max().item() / scale, d.pow(2).mean().sqrt().item() / rms)
        del ref
        # ---- cold timings
        def timed(arith, tuning=0):
            for r_ in range(2):
                A_, B_, C_, kw_ = sets[r_ % ROT]
                run(md, A_, B_, C_, M, N, K, kw_, arith, tuning)
            best = float('inf')
            for _rep in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for r_ in range(6):
                    A_, B_, C_, kw_ = sets[(r_ + 2) % ROT]
                    run(md, A_, B_, C_, M, N, K, kw_, arith, tuning)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 6 * 1e3)
            return best
        t32, tb3 = timed(0), timed(2)
        alts = []
        if sweep:
            big = '--big-tiles' in sys.argv  # + 128x192 (NT / NN / TN) and 192x128 (TN): two workgroups per CU
            for t in (((0, 1, 2, 3, 4) if big else (0, 1, 2)) if mode == 'tn' else ((0, 1, 5, 3) if big else (0, 1, 5))):
                sl = [0]
                if '--splits' in sys.argv:
                    sl = [1, 2, 3, 4, 6, 8] if mode != 'tn' else [8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768]
                for sp in sl:
                    if mode == 'tn' and sp and (K // G) // sp < 64:
                        continue
                    try:
                        alts.append((timed(2, (t + 1) | ((sp & 255) << 8) | ((sp >> 8) << 20)), f'{t}s{sp}' if sp else str(t)))
                    except Exception as ex:  # noqa: BLE001
                        alts.append((float('inf'), f'{t}s{sp}'))
            if '--splits' in sys.argv:
                alts = sorted(alts)[:6]
        tbest = min([tb3] + [a[0] for a in alts])
        tot['f32'] += cnt * t32
        tot['b3'] += cnt * tb3
        tot['b3best'] += cnt * tbest
        rmax = errs['b3'][0] / max(errs['f32'][0], 1e-30)
        rrms = errs['b3'][1] / max(errs['f32'][1], 1e-30)
        worst_ratio = max(worst_ratio, rmax, rrms)
        print(f'{mode} {M} {N} {K} g{G} e{epi} x{cnt} | {errs["f32"][0]:.3e} {errs["b3"][0]:.3e} {rmax:.2f} | '
              f'{errs["f32"][1]:.3e} {errs["b3"][1]:.3e} {rrms:.2f} | {t32:.1f} {tb3:.1f} | ' +
              ' '.join(f't{t}:{x:.1f}' for x, t in alts), flush=True)
        del sets
        torch.cuda.empty_cache()
    print(f'# summed over the step (count-weighted, cold operands): f32 {tot["f32"] / 1e3:.3f} ms, bf16x3 {tot["b3"] / 1e3:.3f} ms, '
          f'bf16x3 best tile per shape {tot["b3best"] / 1e3:.3f} ms')
    print(f'# worst error ratio bf16x3 / native over all shapes (max-norm and rms): {worst_ratio:.3f}  (guardrail: <= 1.5)')


if __name__ == '__main__':
    main()
