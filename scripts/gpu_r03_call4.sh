#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/c4_summary.txt
: > $S
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $S; }
timeout 900 python -m pytest tests/test_losses_gpu.py tests/test_amp_gpu.py -m gpu -q -rf > $O/c4_pytest_a.log 2>&1
stamp "pytest A (losses, amp) rc=$? $(tail -1 $O/c4_pytest_a.log)"
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -rf -k amp > $O/c4_pytest_b.log 2>&1
stamp "fullsize amp rc=$? $(tail -1 $O/c4_pytest_b.log)"
cd /tmp
rm -rf /tmp/op_st
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/op_st -o p -- python $R/scripts/ops_profile.py deform_conv2d_bwd 5 > $O/c4_deform_stats.log 2>&1
find /tmp/op_st -name "*kernel_stats.csv" -exec cp {} $O/c4_deform_bwd_stats.csv \;
stamp "deform stats done"
cat $S
