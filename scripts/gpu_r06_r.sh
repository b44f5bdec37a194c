#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06r; mkdir -p $O
run() { # name, env...
  env "${@:2}" SM3_BENCH_NATIVE=0 python bench.py $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-ops 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $CFG', d['ms_per_step'], d['value'], 'gemm', r['gemm_ms_per_step'], 'other', r['other_kernels_ms_per_step'])" | tee -a $O/ab2.txt
}
CFG="--config SM3Det_convnext_t"
for i in 1 2; do run amp_off SM3_EQ_PRIO=0; run amp_auto SM3_EQ_PRIO=2; done
CFG="--config SM3Det_convnext_b"
run ampB_off SM3_EQ_PRIO=0; run ampB_auto SM3_EQ_PRIO=2
CFG=""
for i in 1 2; do run f32_off SM3_EQ_PRIO=0; run f32_auto SM3_EQ_PRIO=2; done
python -m pytest tests/test_gemm_gpu.py tests/test_amp_gpu.py -q -m gpu -x 2>&1 | tail -n 3
