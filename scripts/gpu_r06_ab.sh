#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ab; mkdir -p $O
python -m pytest tests/test_backbone_gpu.py tests/test_graph_replay_gpu.py tests/test_dp_rccl_gpu.py tests/test_amp_gpu.py -q -m gpu -x 2>&1 | tail -n 5 | tee $O/tests.txt
run() {
  env "${@:2}" SM3_BENCH_NATIVE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops 2>$O/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['ms_per_step'], d['value'], 'gemm', r['gemm_ms_per_step'], 'other', r['other_kernels_ms_per_step'], d.get('peak_mem_gb'))" | tee -a $O/ab.txt
}
for i in 1 2; do
  run defer1 SM3_DEFER_JOIN=1
  run defer0 SM3_DEFER_JOIN=0
done
python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x 2>&1 | tail -n 3 | tee -a $O/tests.txt
