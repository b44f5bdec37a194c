#!/bin/bash
# One entry point for this round's GPU-box calls: gpurun -- 'bash scripts/gpu_r05.sh <step> [<step> ...]'
# Every step writes under gpurun_out/r05/ (scratch); scripts/publish_profiles.sh copies what is judged into profiles/r05/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05
mkdir -p $O
for step in "$@"; do
  echo "=== $step"
  case $step in
    gemm_tests)   timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -15 | tee $O/gemm_tests.txt ;;
    b3_eval)      timeout 900 python scripts/gemm_b3_eval.py > $O/gemm_b3_eval.txt 2>$O/gemm_b3_eval.err; tail -5 $O/gemm_b3_eval.txt; tail -3 $O/gemm_b3_eval.err ;;
    b3_eval_occ3) SM3DET_HIP_LIB=sm3det_amd/csrc/libsm3det_hip_b3_occ3.so timeout 600 python scripts/gemm_b3_eval.py --no-sweep > $O/gemm_b3_eval_occ3.txt 2>&1; tail -3 $O/gemm_b3_eval_occ3.txt ;;
    b3_ablations) for v in abl_noload abl_nostore abl_nocvt abl_nomfma abl_noepi abl_loop_only_mfma; do
        SM3DET_HIP_LIB=sm3det_amd/csrc/libsm3det_hip_$v.so timeout 300 python scripts/gemm_b3_eval.py --no-sweep > $O/gemm_b3_$v.txt 2>&1
        echo "$v: $(tail -2 $O/gemm_b3_$v.txt | head -1)"; done ;;
    b3_variants) for v in ${B3_VARIANTS:-b3_occ3 b3_nomix b3_f1 b3_f1_occ3 b3_f1_nomix}; do
        SM3DET_HIP_LIB=sm3det_amd/csrc/libsm3det_hip_$v.so timeout 300 python scripts/gemm_b3_eval.py --no-sweep > $O/gemm_b3_$v.txt 2>&1
        echo "$v: $(tail -2 $O/gemm_b3_$v.txt | head -1)"; done ;;
    b3_eval_quick) timeout 600 python scripts/gemm_b3_eval.py --no-sweep > $O/gemm_b3_eval_quick.txt 2>&1; tail -2 $O/gemm_b3_eval_quick.txt ;;
    b3_trace)     SM3DET_HIP_LIB=sm3det_amd/csrc/libsm3det_hip_trace.so timeout 600 python scripts/gemm_trace.py $O/gemm_trace_b3.npz > $O/gemm_trace_b3.log 2>&1
                  timeout 300 python scripts/gemm_trace_analyse.py $O/gemm_trace_b3.npz > $O/gemm_trace_b3.txt 2>&1; tail -60 $O/gemm_trace_b3.txt ;;
    b3_sweep)     timeout 1500 python scripts/gemm_b3_eval.py --splits > $O/gemm_b3_sweep.txt 2>$O/gemm_b3_sweep.err; tail -3 $O/gemm_b3_sweep.txt; tail -3 $O/gemm_b3_sweep.err ;;
    bench_quick)  timeout 900 python bench.py --no-ops --no-cpu-baseline > $O/bench_quick.json 2>$O/bench_quick.err; python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r05/bench_quick.json') if l.startswith('{')][-1])
print({k: r.get(k) for k in ('value', 'ms_per_step', 'value_native_f32_mfma', 'ms_per_step_native_f32_mfma', 'loss')})
print(r['roofline'])
print(r['kernels_ms_per_step'])
PY
      tail -3 $O/bench_quick.err ;;
    b3_ab) for v in ${B3_VARIANTS:-main b3_f0 b3_f0_occ2 b3_f0_v1 b3_f1_occ2 b3_f2}; do
        L=sm3det_amd/csrc/libsm3det_hip_$v.so; [ $v = main ] && L=sm3det_amd/csrc/libsm3det_hip.so
        SM3DET_HIP_LIB=$L timeout 300 python scripts/gemm_b3_eval.py --no-sweep > $O/gemm_b3_$v.txt 2>&1
        SM3DET_HIP_LIB=$L SM3_BENCH_NATIVE=0 timeout 600 python bench.py --no-ops --no-cpu-baseline > $O/bench_ab_$v.json 2>/dev/null
        python -c "import json; r=json.loads([l for l in open('$O/bench_ab_$v.json') if l.startswith('{')][-1]); print('$v: step', r['ms_per_step'], 'gemm', r['roofline']['gemm_ms_per_step'], 'other', r['roofline']['other_kernels_ms_per_step'])"
        echo "   $(tail -2 $O/gemm_b3_$v.txt | head -1)"; done ;;
    bench_wgrad)  for w in 0 1; do SM3_WGRAD_STREAM=$w SM3_BENCH_NATIVE=0 timeout 600 python bench.py --no-ops --no-cpu-baseline > $O/bench_wgrad$w.json 2>/dev/null
        python -c "import json; r=json.loads([l for l in open('$O/bench_wgrad$w.json') if l.startswith('{')][-1]); print('wgrad_stream=$w', r['ms_per_step'], r['roofline']['gemm_ms_per_step'], r['roofline']['other_kernels_ms_per_step'])"; done ;;
    bench)        timeout 1500 python bench.py > $O/bench.json 2>$O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err ;;
    fullsize)     timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q 2>&1 | tail -15 | tee $O/fullsize_tests.txt ;;
    backbone)     timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_amp_gpu.py tests/test_graph_replay_gpu.py -x -q 2>&1 | tail -8 | tee $O/backbone_tests.txt ;;
    alltests)     timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/alltests.txt ;;
    heads)        timeout 900 python -m pytest tests/test_roi_head_gpu.py tests/test_detector_gpu.py tests/test_rpn_gpu.py -x -q 2>&1 | tail -12 | tee $O/heads_tests.txt ;;
    optim)        timeout 600 python -m pytest tests/test_optim_gpu.py -x -q 2>&1 | tail -8 | tee $O/optim_tests.txt ;;
    bench_full)   SM3_BENCH_OPS=full SM3_BENCH_NATIVE=0 timeout 1200 python bench.py --no-cpu-baseline > $O/bench_full.json 2>$O/bench_full.err
                  python -c "import json; r=json.loads([l for l in open('$O/bench_full.json') if l.startswith('{')][-1]); print(r['ms_per_step']); print(json.dumps(r.get('full_model'))[:1500])"; tail -3 $O/bench_full.err ;;
    heads2)       timeout 900 python -m pytest tests/test_assign_gpu.py tests/test_gfl_gpu.py tests/test_rpn_gpu.py tests/test_detector_gpu.py tests/test_detector_slice_gpu.py tests/test_losses_gpu.py -x -q 2>&1 | tail -6 | tee $O/heads2_tests.txt ;;
    fpn)          timeout 900 python -m pytest tests/test_fpn_gpu.py tests/test_gfl_gpu.py -x -q 2>&1 | tail -8 | tee $O/fpn_tests.txt ;;
    dp)           timeout 1500 python -m pytest tests/test_dp_rccl_gpu.py -x -q 2>&1 | tail -8 | tee $O/dp_tests.txt ;;
    amp_full)     timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -k amp 2>&1 | tail -5 | tee $O/amp_full_tests.txt ;;
    smoke)        timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -3 ;;
    *) echo "unknown step $step" ;;
  esac
done
