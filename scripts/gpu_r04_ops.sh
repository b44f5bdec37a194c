#!/bin/bash
# round 4: per-operator rocprofv3 profiles (path b) + the detector-slice kernel stats
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r04f}
bash $R/scripts/gpu_r03_ops_profile.sh $TAG > $O/${TAG}_ops_summary.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_slice
SM3_BENCH_OPS=slice timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_slice -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/${TAG}_slice.log 2>&1
find /tmp/prof_slice -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_slice_kernel_stats.csv \;
grep -h '^{' $O/${TAG}_slice.log | tail -1 | head -c 300
