#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c6; S=$O/${T}_summary.txt; : > $S
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o roi -- python scripts/roi_bwd_probe.py tiled > $O/${T}_prof.log 2>&1; echo "rocprof rc=$?" | tee -a $S
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1); echo $f >> $S; head -12 "$f" | cut -c1-200 >> $S
timeout 600 python -m pytest tests/test_amp_gpu.py tests/test_gemm_gpu.py -m gpu -q > $O/${T}_pytest.log 2>&1; echo "pytest amp rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
grep -E "^FAILED|^ERROR" $O/${T}_pytest.log | head >> $S
for w in 0 1; do
timeout 200 env SM3_AMP_W16=$w python bench.py --no-cpu-baseline --no-ops --config SM3Det_convnext_t > $O/${T}_bench_amp_w$w.json 2> $O/${T}_bench_amp_w$w.err; echo "bench amp w16=$w rc=$? $(cut -c100-260 $O/${T}_bench_amp_w$w.json)" | tee -a $S
done
