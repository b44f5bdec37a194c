"""Per-workgroup phase timestamps of the fp32 GEMM kernel on the shapes of one training step (measurement build:
`python -m sm3det_amd.build --variant trace`, run with SM3DET_HIP_LIB=sm3det_amd/csrc/libsm3det_hip_trace.so).

    SM3DET_HIP_LIB=... python scripts/gemm_trace.py gpurun_out/gemm_trace.npz        (GPU box)
    python scripts/gemm_trace.py --analyse gpurun_out/gemm_trace.npz                 (anywhere)

Every workgroup records s_memtime at entry / k-loop start / k-loop end / after the split-K fix-up / exit, its HW_ID and
XCC_ID and s_memrealtime (100 MHz) at entry.  The analysis prints, per shape: the kernel span, the phases' durations, how
many workgroups of a CU are inside their k-loop at a time (the matrix pipe is busy only then) and how far apart the
co-resident workgroups' epilogues are (lockstep = all at once = the store burst is exposed).
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

MAXB = 1 << 18


def operands(mode, M, N, K, G, epi, dev):
    import torch
    from sm3det_amd import _lib_backbone as LB
    rows = K if mode == 'tn' else M
    offs = None
    if G > 1:
        frac = np.array([1.3, 0.7, 1.1, 0.9, 1.0, 1.0, 1.2, 0.8]) / 8.0
        c = (frac * rows).astype(np.int64)
        c[-1] += rows - c.sum()
        offs = torch.tensor(np.concatenate([[0], np.cumsum(c)]), dtype=torch.int32, device=dev)
    bias = torch.randn(G, N, device=dev)
    kw = dict(offsets=offs, num_groups=G)
    if mode == 'nt':
        A, B, C = torch.randn(M, K, device=dev), torch.randn(G, N, K, device=dev) * 0.05, torch.empty(M, N, device=dev)
        kw.update(epilogue=epi, bias=bias)
        if epi == LB.EPI_BIAS_GELU:
            kw.update(aux_out=torch.empty(M, N, device=dev))
        elif epi == LB.EPI_BIAS_SCALE_RES:
            kw.update(aux_in=torch.randn(M, N, device=dev), aux_out=torch.empty(M, N, device=dev),
                      gamma=torch.randn(N, device=dev))
    elif mode == 'nn':
        A, B, C = torch.randn(M, K, device=dev), torch.randn(G, K, N, device=dev) * 0.05, torch.empty(M, N, device=dev)
        if epi == LB.EPI_GELU_BWD:
            kw.update(epilogue=epi, aux_in=torch.randn(M, N, device=dev), colsum_out=torch.empty(G, N, device=dev))
    else:
        A, B, C = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), torch.empty(G, M, N, device=dev)
    return A, B, C, kw


def collect(path):
    import torch
    from scripts.gemm_sweep2 import SHAPES
    from sm3det_amd import _lib, _lib_backbone as LB
    L = _lib.lib()
    L.sm3_gemm_set_trace.argtypes = [ctypes.c_void_p]
    L.sm3_gemm_set_trace.restype = None
    dev = torch.device('cuda')
    trace = torch.zeros(MAXB * 8, dtype=torch.int64, device=dev)
    out = {}
    for si, (mode, M, N, K, G, epi, cnt) in enumerate(SHAPES):
        sets = [operands(mode, M, N, K, G, epi, dev) for _ in range(3)]  # rotate: cold operands like inside the step
        md = dict(nt=LB.NT, nn=LB.NN, tn=LB.TN)[mode]
        L.sm3_gemm_set_trace(None)
        for r in range(3):
            A, B, C, kw = sets[r]
            LB.gemm(md, A, B, C, M, N, K, **kw)
        torch.cuda.synchronize()
        trace.zero_()
        L.sm3_gemm_set_trace(ctypes.c_void_p(trace.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        A, B, C, kw = sets[0]
        e0.record()
        LB.gemm(md, A, B, C, M, N, K, **kw)
        e1.record()
        torch.cuda.synchronize()
        L.sm3_gemm_set_trace(None)
        t = trace.cpu().numpy().reshape(MAXB, 8)
        t = t[t[:, 0] != 0].copy()
        out[f's{si:02d}'] = t
        out[f'm{si:02d}'] = np.array([dict(nt=0, nn=1, tn=2)[mode], M, N, K, G, epi, cnt, int(e0.elapsed_time(e1) * 1e3)])
        print(f'{mode} {M}x{N}x{K} g{G} e{epi}: {t.shape[0]} workgroups, {e0.elapsed_time(e1) * 1e3:.1f} us', flush=True)
        del sets
    np.savez_compressed(path, **out)


def analyse(path, verbose=False):
    z = np.load(path)
    keys = sorted(k for k in z.files if k.startswith('s'))
    names = ['nt', 'nn', 'tn']
    print('shape | WGs | span us (event us) | ideal MFMA us | prologue / loop / fixup / epilogue us (median) | '
          'per-CU: time share with 0/1/2/3+ WGs in loop | loop us per k-tile alone-equivalent | lockstep spread')
    for k in keys:
        t = z[k].astype(np.int64)
        m = z['m' + k[1:]]
        mode, M, N, K, G, epi, cnt, ev_us = [int(v) for v in m]
        done = t[:, 4] != 0
        t0 = t[:, 0].min()
        # shader clock from s_memrealtime (100 MHz): cycles per us over the launch
        rt = t[:, 7]
        span_c = t[done, 4].max() - t0 if done.any() else t[:, 3].max() - t0
        span_rt = (rt.max() - rt.min()) / 100.0  # us between the first and the last workgroup's entry
        ent_c = t[:, 0].max() - t0
        mhz = ent_c / span_rt if span_rt > 2 else 2400.0
        us = lambda c: c / mhz  # noqa: E731
        pro = np.median(t[:, 1] - t[:, 0])
        loop = np.median(t[:, 2] - t[:, 1])
        fix = np.median((t[:, 3] - t[:, 2])[t[:, 3] != 0]) if (t[:, 3] != 0).any() else 0
        epi_c = np.median((t[done, 4] - t[done, 3])) if done.any() else 0
        flops = 2.0 * M * N * K
        ideal = flops / 157.3e12 * 1e6
        # per-CU occupancy of the k-loop phase
        hw = t[:, 5]
        cu_key = ((hw >> 32) & 0xf) * 256 + ((hw >> 8) & 0xff)  # XCC_ID, HW_ID[15:8] = (se, sh, cu)
        shares = np.zeros(5)
        spreads = []
        for c in np.unique(cu_key):
            w = t[cu_key == c]
            ev = []
            for r in w:
                if r[2] > r[1]:
                    ev.append((r[1], 1))
                    ev.append((r[2], -1))
            ev.sort()
            cur, last = 0, t0
            for tt, d in ev:
                shares[min(cur, 4)] += tt - last
                last = tt
                cur += d
            shares[0] += (t0 + span_c) - last
            # spread of loop-end times among workgroups that started within 2000 cycles of each other (one "round")
            st = np.sort(w[:, 0])
            grp = w[np.abs(w[:, 0] - st[0]) < 2000]
            if len(grp) >= 2:
                spreads.append((grp[:, 2].max() - grp[:, 2].min()) / max(np.median(grp[:, 2] - grp[:, 1]), 1))
        shares /= shares.sum()
        nkm = np.median(t[:, 6] & 0xffffffff)
        print(f'{names[mode]} {M}x{N}x{K} g{G} e{epi} x{cnt} | {t.shape[0]} | {us(span_c):.1f} ({ev_us}) | {ideal:.1f} | '
              f'{us(pro):.1f} / {us(loop):.1f} / {us(fix):.1f} / {us(epi_c):.1f} | '
              f'{shares[0]:.2f} {shares[1]:.2f} {shares[2]:.2f} {shares[3] + shares[4]:.2f} | '
              f'{us(loop) / max(nkm, 1):.3f} us/k-tile x{nkm:.0f} | {np.median(spreads) if spreads else -1:.2f} | '
              f'{mhz:.0f} MHz, {len(np.unique(cu_key))} CUs')


if __name__ == '__main__':
    if sys.argv[1] == '--analyse':
        analyse(sys.argv[2])
    else:
        collect(sys.argv[1])
        analyse(sys.argv[1])
