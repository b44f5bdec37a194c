#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ah; mkdir -p $O
run() {
  env "${@:2}" SM3_BENCH_NATIVE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops 2>$O/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['ms_per_step'], d['value'], 'gemm', r['gemm_ms_per_step'], 'other', r['other_kernels_ms_per_step'])" | tee -a $O/ab.txt
}
for i in 1 2; do
  run pen1 SM3_TN_PEN_SCALE=1
  run pen2 SM3_TN_PEN_SCALE=2
  run pen4 SM3_TN_PEN_SCALE=4
  run pen05 SM3_TN_PEN_SCALE=0.5
done
