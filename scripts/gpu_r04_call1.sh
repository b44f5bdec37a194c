#!/bin/bash
# round 4, call 1: parity changes (double d(scale), natural-routing AMP output check, plain-seed fixture) + GEMM diagnostics
# (per-workgroup phase trace, static priority / staggered-start A/B sweeps, weight-gradient side stream A/B)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
T=r04c1
V=$R/sm3det_amd/csrc
S=$O/${T}_summary.txt
: > $S
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s]"; }
timeout 300 env SM3DET_HIP_LIB=$V/libsm3det_hip_trace.so python scripts/gemm_trace.py $O/${T}_trace.npz > $O/${T}_trace.txt 2>&1; echo "$(el) trace rc=$?" | tee -a $S
timeout 200 env SM3DET_HIP_LIB=$V/libsm3det_hip_trace_stagger.so python scripts/gemm_trace.py $O/${T}_trace_stagger.npz > $O/${T}_trace_stagger.txt 2>&1; echo "$(el) trace_stagger rc=$?" | tee -a $S
for v in base prio1 prio2 stagger; do
  L=$V/libsm3det_hip_$v.so; [ $v = base ] && L=$V/libsm3det_hip.so
  timeout 200 env SM3DET_HIP_LIB=$L python scripts/gemm_sweep_amp.py --fp32 --cold --default-only > $O/${T}_sweep_$v.txt 2>&1
  echo "$(el) sweep $v rc=$? $(tail -1 $O/${T}_sweep_$v.txt)" | tee -a $S
done
timeout 200 python bench.py --no-cpu-baseline --no-ops > $O/${T}_bench_base.json 2> $O/${T}_bench_base.err; echo "$(el) bench base rc=$? $(cut -c1-260 $O/${T}_bench_base.json)" | tee -a $S
timeout 200 env SM3_WGRAD_STREAM=1 python bench.py --no-cpu-baseline --no-ops > $O/${T}_bench_wgrad.json 2> $O/${T}_bench_wgrad.err; echo "$(el) bench wgrad-stream rc=$? $(cut -c1-260 $O/${T}_bench_wgrad.json)" | tee -a $S
timeout 200 env SM3DET_HIP_LIB=$V/libsm3det_hip_prio1.so python bench.py --no-cpu-baseline --no-ops > $O/${T}_bench_prio1.json 2> $O/${T}_bench_prio1.err; echo "$(el) bench prio1 rc=$? $(cut -c1-260 $O/${T}_bench_prio1.json)" | tee -a $S
timeout 200 env SM3DET_HIP_LIB=$V/libsm3det_hip_prio2.so python bench.py --no-cpu-baseline --no-ops > $O/${T}_bench_prio2.json 2> $O/${T}_bench_prio2.err; echo "$(el) bench prio2 rc=$? $(cut -c1-260 $O/${T}_bench_prio2.json)" | tee -a $S
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_backbone_gpu.py tests/test_amp_gpu.py tests/test_ops_gpu.py -m gpu -q > $O/${T}_pytest.log 2>&1; echo "$(el) pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
cp gpurun_out/fullsize_*.json $O/ 2>/dev/null
echo "$(el) done" | tee -a $S
