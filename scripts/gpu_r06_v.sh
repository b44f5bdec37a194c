#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06v; mkdir -p $O
for R in 3 4 6 3 4 6; do
  SM3_EQ_PRIO_MAXR=$R SM3_BENCH_NATIVE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('MAXR=$R', d['ms_per_step'], d['value'], 'gemm', r['gemm_ms_per_step'], 'other', r['other_kernels_ms_per_step'])" | tee -a $O/ab.txt
done
