#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/c7_summary.txt
: > $S
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $S; }
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_rpn_gpu.py tests/test_ref_wrappers.py tests/test_detector_slice_gpu.py -m gpu -q -rf > $O/c7_pytest_a.log 2>&1
stamp "pytest A (ops, rpn) rc=$? $(tail -1 $O/c7_pytest_a.log)"
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -rf -k amp > $O/c7_pytest_b.log 2>&1
stamp "fullsize amp rc=$? $(tail -1 $O/c7_pytest_b.log)"
timeout 900 python bench.py --no-cpu-baseline > $O/c7_bench.json 2> $O/c7_bench.err
stamp "bench rc=$? $(head -c 200 $O/c7_bench.json)"
cat $S
