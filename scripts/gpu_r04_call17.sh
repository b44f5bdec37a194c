#!/bin/bash
# round 4, call 17: rocprofv3 kernel stats of the full-model workload alone
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_full
SM3_BENCH_OPS=full timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/r04c17_full.log 2>&1
find /tmp/prof_full -name "*kernel_stats.csv" -exec cp {} $O/r04c17_full_kernel_stats.csv \;
grep -h '^{' $O/r04c17_full.log | tail -1 | head -c 200
