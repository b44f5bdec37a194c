#!/bin/bash
# round 4, call 16: device-side sampler (sampler.hip) -- parity with the host rule, the heads' tests, the slice / full model
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c16; S=$O/${T}_summary.txt; : > $S
timeout 900 python -m pytest tests/test_assign_gpu.py tests/test_rpn_gpu.py tests/test_roi_head_gpu.py tests/test_detector_slice_gpu.py tests/test_detector_gpu.py tests/test_losses_gpu.py -m gpu -x -q > $O/${T}_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
timeout 600 python bench.py --no-cpu-baseline --steps 10 > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?" | tee -a $S
python - <<'PY' | tee -a $S
import json
d=json.loads(open('/root/repo/gpurun_out/r04c16_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","full_model_imgs_per_sec","full_slice_imgs_per_sec")})
print(d['ops_us'].get('detector_slice_eager_train_step_bs2_1024'), d['ops_us'].get('detector_slice_train_step_bs2_1024'), d['full_model'].get('ms_per_step_graph'), d['full_model'].get('ms_per_step_eager'))
PY
