#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/c8_summary.txt
: > $S
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $S; }
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_rpn_gpu.py -m gpu -q -rf > $O/c8_pytest_a.log 2>&1
stamp "pytest A (ops, rpn) rc=$? $(tail -1 $O/c8_pytest_a.log)"
cd /tmp
for OP in nms_8768 nms_rotated_10000; do
  rm -rf /tmp/op_st
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/op_st -o p -- python $R/scripts/ops_profile.py $OP 5 > $O/c8_${OP}.log 2>&1
  find /tmp/op_st -name "*kernel_stats.csv" -exec cp {} $O/c8_${OP}_stats.csv \;
done
stamp "nms stats done"
cat $S
