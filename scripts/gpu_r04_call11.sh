#!/bin/bash
# round 4, call 11: gather-form RoI backward with overwrite mode -- parity, timings, kernel stats, detector slice + bench
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c11; S=$O/${T}_summary.txt; : > $S
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_roi_head_gpu.py tests/test_detector_gpu.py -m gpu -x -q > $O/${T}_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
timeout 300 python scripts/ops_quick.py > $O/${T}_ops.log 2>&1; echo "ops_quick rc=$?" | tee -a $S
grep -E "roi|extract" $O/${T}_ops.log | tee -a $S
rm -rf /tmp/prof_roi; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_roi -o roi -- python scripts/roi_bwd_probe.py tiled > $O/${T}_prof.log 2>&1
f=$(find /tmp/prof_roi -name "*kernel_stats.csv" | head -1)
cp $f $O/${T}_roi_bwd_tiled_kernel_stats.csv
head -8 $f | cut -c1-160 | tee -a $S
timeout 600 python bench.py --no-cpu-baseline > $O/${T}_bench_full.json 2> $O/${T}_bench_full.err; echo "bench rc=$?" | tee -a $S
python - <<'PY' | tee -a $S
import json,sys
try:
    d=json.loads(open('/root/repo/gpurun_out/r04c11_bench_full.json').read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","full_model_imgs_per_sec","full_slice_imgs_per_sec","full_model")})
except Exception as e: print('parse fail',e)
PY
