#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/tail_bench.py > gpurun_out/tail_bench.log 2>&1
cat gpurun_out/tail_bench.log | grep -v amdgpu.ids
