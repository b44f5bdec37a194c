#!/bin/bash
# round 3, GPU call 2: loss kernels vs oracle, fp16-storage GEMMs / AMP data path, bench with the real-loss slice, ops PMC
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/c2_summary.txt
: > $S
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $S; }
timeout 900 python -m pytest tests/test_losses_gpu.py tests/test_amp_gpu.py tests/test_assign_gpu.py tests/test_rpn_gpu.py tests/test_roi_head_gpu.py -m gpu -q -rf -x > $O/c2_pytest_a.log 2>&1
stamp "pytest A (losses, amp gemm) rc=$? $(tail -1 $O/c2_pytest_a.log)"
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_backbone_gpu.py tests/test_detector_slice_gpu.py -m gpu -q -rf > $O/c2_pytest_b.log 2>&1
stamp "pytest B (fullsize, backbone) rc=$? $(tail -1 $O/c2_pytest_b.log)"
for C in SM3Det_convnext_t SM3Det_convnext_b; do
  timeout 600 python bench.py --config $C --no-ops --no-cpu-baseline > $O/c2_bench_$C.json 2> $O/c2_bench_$C.err
  stamp "bench $C rc=$? $(head -c 250 $O/c2_bench_$C.json)"
  SM3_AMP_STORAGE=fp32 timeout 600 python bench.py --config $C --no-ops --no-cpu-baseline > $O/c2_bench_${C}_fp32storage.json 2> $O/c2_bench_${C}_fp32storage.err
  stamp "bench $C fp32-storage rc=$? $(head -c 250 $O/c2_bench_${C}_fp32storage.json)"
done
timeout 900 python bench.py --no-cpu-baseline > $O/c2_bench.json 2> $O/c2_bench.err
stamp "bench default rc=$? $(head -c 250 $O/c2_bench.json)"
bash scripts/gpu_r03_ops_profile.sh c2 > $O/c2_ops_profile.log 2>&1
stamp "ops profile done"
cat $S
