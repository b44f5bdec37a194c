#!/bin/bash
# round 4, call 14: fp16 weight shadows kept by the optimizer -- AMP / optimizer tests, AMP bench
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c14; S=$O/${T}_summary.txt; : > $S
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_amp_gpu.py tests/test_optim_gpu.py tests/test_graph_replay_gpu.py -m gpu -x -q > $O/${T}_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
for w in 0 1 0 1; do
SM3_AMP_W16=$w timeout 600 python bench.py --config SM3Det_convnext_t --no-cpu-baseline --no-ops > $O/${T}_bench_w$w.json 2> $O/${T}_bench_w$w.err; echo "bench w16=$w rc=$? $(python -c "import json;d=json.loads(open('$O/${T}_bench_w$w.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['kernels_ms_per_step'].get('adamw_multi'), d['kernels_ms_per_step'].get('cast_f32_f16'))")" | tee -a $S
done
