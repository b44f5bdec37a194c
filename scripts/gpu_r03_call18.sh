#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python scripts/gemm_sweep_amp.py --fp32 --default-only 2>&1 | tail -1
timeout 900 python scripts/gemm_sweep_amp.py --fp32 --cold --splits > $O/c18_sweep_fp32_cold.txt 2>&1; tail -1 $O/c18_sweep_fp32_cold.txt
