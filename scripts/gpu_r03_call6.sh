#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/c6_summary.txt
: > $S
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $S; }
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_backbone_gpu.py tests/test_amp_gpu.py -m gpu -q -rf > $O/c6_pytest_b.log 2>&1
stamp "fullsize + backbone + amp rc=$? $(tail -1 $O/c6_pytest_b.log)"
cat $S
