#!/bin/bash
# round 4, call 4: RoIAlignRotated (vector forward, tiled backward) + fused DeformConv2d forward: tests and timings
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
T=r04c4
S=$O/${T}_summary.txt
: > $S
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s]"; }
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_deform_conv_gpu.py tests/test_rpn_gpu.py tests/test_roi_head_gpu.py tests/test_ref_wrappers.py tests/test_detector_gpu.py -m gpu -q > $O/${T}_pytest.log 2>&1; echo "$(el) pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
grep -E "FAILED|Error" $O/${T}_pytest.log | head -20 >> $S
timeout 900 python bench.py --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "$(el) bench rc=$? $(cut -c1-200 $O/${T}_bench.json)" | tee -a $S
timeout 300 env SM3_ROI_BWD=atomic SM3_DEFORM_FUSED=0 SM3_BENCH_OPS=skip_models python - <<'PY' >> $S 2>&1
import json
d=json.load(open('gpurun_out/r04c4_bench.json'))
u=d['ops_us']
for k in sorted(u):
    if any(s in k for s in ('roi_','deform','slice')): print(k, u[k])
print('full_model', d.get('full_model_imgs_per_sec'), (d.get('full_model') or {}).get('ms_per_step_graph'))
print('slice', d.get('full_slice_imgs_per_sec'))
r=d.get('ops_roofline',{})
for k in r:
    if 'roi' in k or 'deform' in k: print(k, {kk:r[k][kk] for kk in ('us','achieved','frac') if kk in r[k]})
PY
echo "$(el) done" | tee -a $S
