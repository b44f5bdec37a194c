#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py --no-cpu-baseline --no-ops > $O/c29_bench.json 2> $O/c29_bench.err; head -c 300 $O/c29_bench.json; echo; tail -2 $O/c29_bench.err
python bench.py --config SM3Det_convnext_t --no-cpu-baseline --no-ops > $O/c29_bench_amp.json 2> $O/c29_bench_amp.err; head -c 300 $O/c29_bench_amp.json; echo
timeout 300 python -m pytest tests/test_optim_gpu.py -m gpu -q 2>&1 | tail -1
