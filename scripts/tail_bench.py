"""Time the HBM-bound kernels of the backbone step in isolation at the step's shapes (bs 2 @ 1024^2, ConvNeXt-T):
depthwise 7x7 forward / input gradient / weight gradient, LayerNorm forward / backward, scale_bwd_prep, combine.
Prints microseconds and algorithmic GB/s per (kernel, stage) and the per-step total (launch counts of config #2).

    python scripts/tail_bench.py [--reps 20]
"""
import os
import sys
import time

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sm3det_amd import _lib_backbone as LB  # noqa: E402

call = LB.call
REPS = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 20
BLOCKS = [3, 3, 9, 3]  # blocks per stage (each runs dwconv fwd + dgrad + wgrad, LN fwd + bwd once)


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / REPS * 1e3)
    return sorted(ts)[1]


def main():
    B = 2
    total = {}
    for s in range(4):
        H = W = 256 >> s
        C = 96 << s
        T = B * H * W
        r = lambda *sh: torch.randn(*sh, device='cuda')  # noqa: E731
        x, u, du, dout = r(T, C), r(T, C), r(T, C), r(T, C)
        w49, bdw, lnw, lnb = r(49, C), r(C), r(C), r(C)
        mean, rstd = torch.zeros(T, device='cuda'), torch.ones(T, device='cuda')
        xn, dx = torch.empty(T, C, device='cuda'), torch.empty(T, C, device='cuda')
        dwb = torch.zeros(50, C, device='cuda')
        ws, nb = LB.row_ws(C, x)
        rows = [
            ('dwconv7_fwd', lambda: call('dwconv7_fwd', x, w49, bdw, None, u, B, H, W, C, 0), 8.0 * T * C),
            ('dwconv7_dgrad', lambda: call('dwconv7_fwd', du, w49, None, dout, dx, B, H, W, C, 1), 12.0 * T * C),
            ('dwconv7_wgrad', lambda: call('dwconv7_bwd_weight', x, du, dwb[:49], dwb[49], B, H, W, C), 8.0 * T * C),
            ('layernorm_fwd', lambda: call('layernorm_fwd', u, lnw, lnb, 1e-6, xn, mean, rstd, T, C, 0, H, W), 8.0 * T * C),
            ('layernorm_bwd', lambda: call('layernorm_bwd', dout, u, lnw, mean, rstd, dx, None, T, C, 0, H, W, 0, ws, nb),
             12.0 * T * C),
        ]
        for name, fn, nbytes in rows:
            us = timeit(fn)
            total[name] = total.get(name, 0.0) + us * BLOCKS[s]
            print(f'stage {s} T={T:6d} C={C:4d} {name:16s} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s  '
                  f'(HBM floor {nbytes / 6.3e6:6.1f} us)')
    print('per-step totals (us):', {k: round(v, 1) for k, v in total.items()}, 'sum', round(sum(total.values()), 1))


if __name__ == '__main__':
    main()
