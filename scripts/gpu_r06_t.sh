#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06t; mkdir -p $O
SM3_EQ_PRIO=2 SM3DET_HIP_LIB=$PWD/sm3det_amd/csrc/libsm3det_hip_trace.so python scripts/gemm_trace.py $O/gemm_trace_eq2.npz > $O/gemm_trace_eq2.txt 2>&1
for M in 0 2 0 2; do
  SM3_EQ_PRIO=$M SM3_BENCH_NATIVE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('EQ_PRIO=$M', d['ms_per_step'], d['value'], 'gemm', r['gemm_ms_per_step'], 'other', r['other_kernels_ms_per_step'])" | tee -a $O/ab.txt
done
