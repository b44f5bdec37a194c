#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06x; mkdir -p $O
python -m pytest tests/test_backbone_gpu.py tests/test_graph_replay_gpu.py tests/test_fullsize_gpu.py tests/test_dp_rccl_gpu.py tests/test_amp_gpu.py tests/test_detector_gpu.py tests/test_detector_slice_gpu.py -q -m gpu 2>&1 | tail -n 6 | tee $O/tests.txt
run() {
  env "${@:3}" SM3_BENCH_NATIVE=0 python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline --no-ops 2>$O/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['ms_per_step'], d['value'], 'gemm', r['gemm_ms_per_step'], 'other', r['other_kernels_ms_per_step'])" | tee -a $O/ab.txt
}
run amp_pair0 "--config SM3Det_convnext_t" SM3_PAIR_DGRAD=0
run amp_pair4 "--config SM3Det_convnext_t" SM3_PAIR_DGRAD=4
run ampB_pair0 "--config SM3Det_convnext_b" SM3_PAIR_DGRAD=0
run ampB_pair4 "--config SM3Det_convnext_b" SM3_PAIR_DGRAD=4
run bf32_pair0 "--config SM3Det_convnext_b --fp32" SM3_PAIR_DGRAD=0
run bf32_pair4 "--config SM3Det_convnext_b --fp32" SM3_PAIR_DGRAD=4
run e16_pair0 "--config e16t2" SM3_PAIR_DGRAD=0
run e16_pair4 "--config e16t2" SM3_PAIR_DGRAD=4
for LS in 0 4; do
SM3_PAIR_DGRAD=$LS SM3_BENCH_NATIVE=0 SM3_BENCH_OPS=full python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$O/full_$LS.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); fm=d.get('full_model') or {}
print('full PAIR=$LS', {k:v for k,v in fm.items() if 'ms' in k})" | tee -a $O/ab.txt
done
