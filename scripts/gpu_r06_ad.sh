#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ad; mkdir -p $O
run() {
  env "${@:2}" SM3_BENCH_NATIVE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops 2>$O/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['ms_per_step'], d['value'], 'gemm', r['gemm_ms_per_step'], 'other', r['other_kernels_ms_per_step'])" | tee -a $O/ab.txt
}
for i in 1 2; do
  run p4 SM3_PAIR_DGRAD=4
  run p5 SM3_PAIR_DGRAD=5
  run p5_8M SM3_PAIR_DGRAD=5 SM3_FWD_SPLIT_MAX_OUTPUTS=8000000
done
tail -n 3 $O/err_p5.txt
SM3_PAIR_DGRAD=5 python -m pytest tests/test_backbone_gpu.py tests/test_graph_replay_gpu.py -q -m gpu -x 2>&1 | tail -n 3
