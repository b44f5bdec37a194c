#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ak; mkdir -p $O
run() {
  env "${@:2}" SM3_BENCH_NATIVE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops 2>$O/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['ms_per_step'], d['value'], 'gemm', r['gemm_ms_per_step'], 'other', r['other_kernels_ms_per_step'], d['config'].get('backward_segments'))" | tee -a $O/ab.txt
}
for i in 1 2; do
  run one
  run seg2 SM3_BENCH_SPLIT=1 SM3_BENCH_SEGMENTS=2
  run seg4 SM3_BENCH_SPLIT=1 SM3_BENCH_SEGMENTS=4
done
