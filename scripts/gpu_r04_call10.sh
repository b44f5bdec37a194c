#!/bin/bash
# round 4, call 10: wave-per-16-pixels gather form of the tiled RoIAlignRotated backward -- parity, timings, kernel stats
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c10; S=$O/${T}_summary.txt; : > $S
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q > $O/${T}_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
timeout 300 python scripts/ops_quick.py > $O/${T}_ops.log 2>&1; echo "ops_quick rc=$?" | tee -a $S
grep -E "roi|extract" $O/${T}_ops.log | tee -a $S
rm -rf /tmp/prof_roi; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_roi -o roi -- python scripts/roi_bwd_probe.py tiled > $O/${T}_prof.log 2>&1
f=$(find /tmp/prof_roi -name "*kernel_stats.csv" | head -1)
cp $f $O/${T}_roi_bwd_tiled_kernel_stats.csv
head -12 $f | cut -c1-200 | tee -a $S
