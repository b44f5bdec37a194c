#!/bin/bash
# round 4, call 3: TriSourceDetector training step (tests + bench full_model), sustained MFMA peak probe, torch-launch audit
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
T=r04c3
S=$O/${T}_summary.txt
: > $S
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s]"; }
timeout 600 python -m pytest tests/test_detector_gpu.py tests/test_gemm_persistent_gpu.py tests/test_backbone_gpu.py -m gpu -x -q > $O/${T}_pytest.log 2>&1; echo "$(el) pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
timeout 120 python scripts/probes/mfma_peak.py > $O/${T}_mfma_peak.txt 2>&1; echo "$(el) mfma probe rc=$?" | tee -a $S; cat $O/${T}_mfma_peak.txt >> $S
timeout 300 python scripts/glue_audit.py > $O/${T}_glue.txt 2>&1; echo "$(el) glue audit rc=$?" | tee -a $S
timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "$(el) bench rc=$? $(cut -c1-200 $O/${T}_bench.json)" | tee -a $S
python - <<'PY' >> $S 2>&1
import json
d=json.load(open('gpurun_out/r04c3_bench.json'))
print('full_model', json.dumps(d.get('full_model'))[:1500])
print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:600])
print('slice', d.get('full_slice_imgs_per_sec'), 'roofline', d['roofline']['frac'], d['ms_per_step'])
PY
echo "$(el) done" | tee -a $S
