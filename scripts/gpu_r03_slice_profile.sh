#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_slice
SM3_BENCH_OPS=slice timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_slice -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/c25_slice.log 2>&1
find /tmp/prof_slice -name "*kernel_stats.csv" -exec cp {} $O/c25_slice_kernel_stats.csv \;
grep -h '^{' $O/c25_slice.log | tail -1 | head -c 300
