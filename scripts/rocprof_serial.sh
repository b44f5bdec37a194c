#!/bin/bash
# quick look: rocprofv3 kernel statistics of the serialized training step (no PMC): gpurun_out/<tag>_kernel_stats_serial.csv
TAG=${1:-q}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
SM3_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ops ${@:2} > $O/${TAG}_rocprof_serial.log 2>&1
find /tmp/prof_q -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_serial.csv \;
python - <<EOF
import csv
rows=list(csv.DictReader(open("$O/${TAG}_kernel_stats_serial.csv")))
g=o=0
for r in rows:
    t=float(r['TotalDurationNs'])/1e6
    if 'gemm_f32_kernel' in r['Name'] or 'splitk_reduce' in r['Name']: g+=t
    else: o+=t
print('per step: gemm family %.3f ms  other %.3f ms' % (g/12, o/12))
for r in rows[:70]:
    if 'gemm_f32_kernel' in r['Name']: continue
    print('%-70s calls=%5s avg=%8.1f us  /step=%.3f ms' % (r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/12e6))
EOF
