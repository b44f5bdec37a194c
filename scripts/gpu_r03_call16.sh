#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_amp_gpu.py -m gpu -q -x 2>&1 | tail -1
timeout 900 python scripts/gemm_sweep_amp.py --cold > $O/c16_sweep_cold.txt 2>&1; tail -1 $O/c16_sweep_cold.txt
