#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python scripts/gemm_sweep_amp.py --cold --splits > $O/c14_sweep_cold.txt 2>&1; tail -1 $O/c14_sweep_cold.txt
