#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ag; mkdir -p $O
run() {
  env "${@:2}" SM3_BENCH_NATIVE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops 2>$O/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['ms_per_step'], d['value'], 'gemm', r['gemm_ms_per_step'], 'other', r['other_kernels_ms_per_step'])" | tee -a $O/ab.txt
}
for i in 1 2; do
  run s1 SM3_SIDE_STREAMS=1
  run s2 SM3_SIDE_STREAMS=2
  run s3 SM3_SIDE_STREAMS=3
done
