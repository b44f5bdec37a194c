"""Idle time between consecutive kernels of the replayed training step, from a rocprofv3 --kernel-trace csv.

usage: python scripts/kernel_gaps.py <..._kernel_trace.csv> [steps_to_skip]
Prints, for the steady-state part of the trace: kernels per step, busy time, idle time between kernels (gaps), and the gap
histogram by the kernel that FOLLOWS the gap (who waits longest to start)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda k: k[0])
# steady state: the last 40 % of the trace (graph replays of the timed region)
n = len(ks)
ks = ks[int(n * 0.6):]
adam = [i for i, k in enumerate(ks) if 'adamw_multi' in k[2]]
if len(adam) < 3:
    raise SystemExit('fewer than 3 optimizer launches in the window')
lo, hi = adam[0] + 1, adam[-1] + 1
steps = len(adam) - 1
seg = ks[lo:hi]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
gaps = defaultdict(lambda: [0, 0])
tot_gap = 0
prev_end = seg[0][1]
for s, e, nm in seg[1:]:
    g = s - prev_end
    if g > 0:
        tot_gap += g
        key = nm.split('(')[0][-60:]
        gaps[key][0] += g
        gaps[key][1] += 1
    prev_end = max(prev_end, e)
print(f'steps {steps}  kernels/step {len(seg) / steps:.1f}  span/step {span / steps / 1e6:.3f} ms  busy/step {busy / steps / 1e6:.3f} ms  '
      f'gaps/step {tot_gap / steps / 1e6:.3f} ms  mean gap {tot_gap / max(1, len(seg)) / 1e3:.2f} us')
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f'{g / steps / 1e3:9.1f} us/step  {c / steps:6.1f} gaps/step  mean {g / c / 1e3:6.2f} us  before {k}')
