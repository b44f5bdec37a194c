#!/bin/bash
# round 4, call 22: GFL loss over all levels in one pass -- GPU tests, full-model workload
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c22; S=$O/${T}_summary.txt; : > $S
timeout 900 python -m pytest tests/test_gfl_gpu.py tests/test_detector_gpu.py -m gpu -x -q > $O/${T}_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
SM3_BENCH_OPS=full timeout 600 python bench.py --no-cpu-baseline --steps 5 > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?" | tee -a $S
python - <<'PY' | tee -a $S
import json
d=json.loads(open('/root/repo/gpurun_out/r04c22_bench.json').read().strip().splitlines()[-1])
fm=d['full_model']; print(fm.get('ms_per_step_graph'), fm.get('ms_per_step_eager'), fm.get('imgs_per_sec')); print(fm.get('losses_first_step')); print(fm.get('losses_last_step'))
PY
