#!/bin/bash
# round 3, GPU call 1: validate the new fixtures / tests / bench entry, first lines for configs #3 / #4 / #5, GELU A/B
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/c1_summary.txt
: > $S
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $S; }
# 1. the changed / new GPU tests (the whole suite runs at the end of the round)
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py tests/test_assign_gpu.py tests/test_rpn_gpu.py \
  tests/test_roi_head_gpu.py tests/test_graph_replay_gpu.py tests/test_gemm_gpu.py tests/test_optim_gpu.py tests/test_amp_gpu.py \
  -m gpu -q -rf > $O/c1_pytest_a.log 2>&1
stamp "pytest A rc=$? $(tail -1 $O/c1_pytest_a.log)"
timeout 1200 python -m pytest tests/test_dp_rccl_gpu.py -m gpu -q -rf > $O/c1_pytest_b.log 2>&1
stamp "pytest B (bench entry) rc=$? $(tail -1 $O/c1_pytest_b.log)"
# 2. the headline bench line, as the driver runs it
timeout 600 python bench.py > $O/c1_bench.json 2> $O/c1_bench.err
stamp "bench default rc=$? $(head -c 400 $O/c1_bench.json)"
# 3. the other BASELINE configurations at full size
for C in e16t2 SM3Det_convnext_b SM3Det_convnext_t; do
  timeout 600 python bench.py --config $C --no-ops --no-cpu-baseline > $O/c1_bench_$C.json 2> $O/c1_bench_$C.err
  stamp "bench $C rc=$? $(head -c 300 $O/c1_bench_$C.json)"
done
timeout 600 python bench.py --config SM3Det_convnext_b --fp32 --no-ops --no-cpu-baseline > $O/c1_bench_convnext_b_fp32.json 2> $O/c1_bench_convnext_b_fp32.err
stamp "bench convnext_b fp32 rc=$? $(head -c 300 $O/c1_bench_convnext_b_fp32.json)"
# 4. GELU A/B: the headline parity case on a build whose GELU epilogues use ocml erff / expf
mkdir -p $O/c1_gelu_poly $O/c1_gelu_exact
cp $O/fullsize_full_e8t2_b2.json $O/c1_gelu_poly/ 2>/dev/null
SM3DET_HIP_LIB=$R/sm3det_amd/csrc/libsm3det_hip_gelu_exact.so timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -k "full_e8t2_b2 and not amp" > $O/c1_gelu_exact.log 2>&1
stamp "gelu exact rc=$? $(tail -1 $O/c1_gelu_exact.log)"
cp $O/fullsize_full_e8t2_b2.json $O/c1_gelu_exact/ 2>/dev/null
# 5. per-kernel tables of the two bigger configs (kernels serialized)
cd /tmp
for C in SM3Det_convnext_b e16t2; do
  rm -rf /tmp/prof_$C
  SM3_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$C -o p -- python $R/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline --no-ops > $O/c1_rocprof_$C.log 2>&1
  find /tmp/prof_$C -name "*kernel_stats.csv" -exec cp {} $O/c1_kernel_stats_$C.csv \;
  stamp "rocprof $C done"
done
stamp done
cat $S
