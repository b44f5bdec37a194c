"""Idle time between the kernels of one hipGraph-replayed training step.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ops
    python scripts/graph_gaps.py /tmp/kt            (prints; redirect into profiles/rNN/graph_gaps.txt)

A step is delimited by consecutive `adamw_multi_kernel` launches; the last complete step of the trace (a graph replay) is
analysed: busy time (union of the kernel intervals), idle time, the idle time by the kernel that PRECEDES the gap, and the
kernels that overlapped another one (side streams)."""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    f = [p for p in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)]
    assert f, 'no *kernel_trace.csv under ' + d
    rows = []
    for r in csv.DictReader(open(f[0])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if 'adamw_multi_kernel' in r[2]]
    assert len(marks) >= 3, 'need >= 3 optimizer launches'
    # the steps between consecutive optimizer launches; the graph replays are the shortest ones (warm-up / event-timed eager
    # passes are longer): analyse the one with the median span of the shortest five
    spans = sorted((rows[marks[i + 1]][1] - rows[marks[i]][1], i) for i in range(len(marks) - 1))
    pick = spans[:5][len(spans[:5]) // 2][1]
    a, b = marks[pick] + 1, marks[pick + 1] + 1
    step = rows[a:b]
    t0, t1 = rows[marks[pick]][1], step[-1][1]
    print('step spans (ms), shortest first:', [round(x / 1e6, 3) for x, _ in spans[:8]])
    busy = 0
    idle_by = collections.Counter()
    n_by = collections.Counter()
    cur_end, prev = t0, 'adamw_multi_kernel (previous step)'
    overlapped = 0
    for s, e, n in step:
        if s > cur_end:
            idle_by[prev] += s - cur_end
            n_by[prev] += 1
            busy += e - s
            cur_end, prev = e, n
        else:
            overlapped += 1
            if e > cur_end:
                busy += e - cur_end
                cur_end, prev = e, n
    span = t1 - t0
    short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:70]  # noqa: E731
    print(f'last replayed step: {len(step)} kernels, span {span / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, idle '
          f'{(span - busy) / 1e6:.3f} ms = {100.0 * (span - busy) / span:.1f} %; sum of kernel durations '
          f'{sum(e - s for s, e, _ in step) / 1e6:.3f} ms; {overlapped} kernels started while another one ran')
    tot = collections.Counter()
    for k, v in idle_by.items():
        tot[short(k)] += v
    cnt = collections.Counter()
    for k, v in n_by.items():
        cnt[short(k)] += v
    print('idle time after a kernel of this name (us total, gaps, us per gap):')
    for k, v in tot.most_common(25):
        print(f'  {v / 1e3:8.1f} {cnt[k]:5d} {v / 1e3 / cnt[k]:6.2f}  {k}')
    dur = collections.Counter()
    num = collections.Counter()
    for s, e, n in step:
        dur[short(n)] += e - s
        num[short(n)] += 1
    print('kernel time of the step by name (us total, launches):')
    for k, v in dur.most_common(40):
        print(f'  {v / 1e3:8.1f} {num[k]:5d}  {k}')


if __name__ == '__main__':
    main()
