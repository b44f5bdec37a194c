#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py > $O/c26_bench.json 2> $O/c26_bench.err; head -c 300 $O/c26_bench.json; echo
timeout 600 python -m pytest tests/test_dp_rccl_gpu.py tests/test_detector_slice_gpu.py -m gpu -q 2>&1 | tail -1
