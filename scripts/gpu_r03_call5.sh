#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/c5_summary.txt
: > $S
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $S; }
timeout 900 python -m pytest tests/test_deform_conv_gpu.py tests/test_ref_wrappers.py -m gpu -q -rf > $O/c5_pytest_a.log 2>&1
stamp "pytest A (deform) rc=$? $(tail -1 $O/c5_pytest_a.log)"
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -rf -k amp > $O/c5_pytest_b.log 2>&1
stamp "fullsize amp rc=$? $(tail -1 $O/c5_pytest_b.log)"
cd /tmp
rm -rf /tmp/op_st
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/op_st -o p -- python $R/scripts/ops_profile.py deform_conv2d_bwd 5 > $O/c5_deform_stats.log 2>&1
find /tmp/op_st -name "*kernel_stats.csv" -exec cp {} $O/c5_deform_bwd_stats.csv \;
stamp "deform stats done"
cat $S
