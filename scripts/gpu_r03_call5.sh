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
true
stamp "pytest A (deform) rc=$? $(tail -1 $O/c5_pytest_a.log)"
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -rf -k amp > $O/c5_pytest_b.log 2>&1
stamp "fullsize amp rc=$? $(tail -1 $O/c5_pytest_b.log)"
cd /tmp
rm -rf /tmp/op_st
stamp "deform stats done"
cat $S
