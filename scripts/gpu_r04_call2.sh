#!/bin/bash
# round 4, call 2: persistent GEMM form -- correctness, A/B against one-workgroup-per-tile (SM3_GEMM_PERSIST=0), trace
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
T=r04c2
V=$R/sm3det_amd/csrc
S=$O/${T}_summary.txt
: > $S
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s]"; }
timeout 600 python -m pytest tests/test_gemm_persistent_gpu.py tests/test_gemm_gpu.py -m gpu -x -q > $O/${T}_pytest_gemm.log 2>&1; echo "$(el) pytest gemm rc=$? $(tail -1 $O/${T}_pytest_gemm.log)" | tee -a $S
for v in on off; do
  P=1; [ $v = off ] && P=0
  timeout 200 env SM3_GEMM_PERSIST=$P python scripts/gemm_sweep_amp.py --fp32 --cold --default-only > $O/${T}_sweep_fp32_$v.txt 2>&1
  echo "$(el) sweep fp32 persist=$v rc=$? $(tail -1 $O/${T}_sweep_fp32_$v.txt)" | tee -a $S
  timeout 200 env SM3_GEMM_PERSIST=$P python scripts/gemm_sweep_amp.py --cold --default-only > $O/${T}_sweep_amp_$v.txt 2>&1
  echo "$(el) sweep amp persist=$v rc=$? $(tail -1 $O/${T}_sweep_amp_$v.txt)" | tee -a $S
done
timeout 300 env SM3DET_HIP_LIB=$V/libsm3det_hip_trace.so python scripts/gemm_trace.py $O/${T}_trace.npz > $O/${T}_trace.txt 2>&1; echo "$(el) trace rc=$?" | tee -a $S
for v in on off; do
  P=1; [ $v = off ] && P=0
  timeout 200 env SM3_GEMM_PERSIST=$P python bench.py --no-cpu-baseline --no-ops > $O/${T}_bench_$v.json 2> $O/${T}_bench_$v.err; echo "$(el) bench fp32 persist=$v rc=$? $(cut -c1-230 $O/${T}_bench_$v.json)" | tee -a $S
  timeout 200 env SM3_GEMM_PERSIST=$P python bench.py --no-cpu-baseline --no-ops --config SM3Det_convnext_t > $O/${T}_bench_amp_$v.json 2> $O/${T}_bench_amp_$v.err; echo "$(el) bench amp persist=$v rc=$? $(cut -c1-230 $O/${T}_bench_amp_$v.json)" | tee -a $S
done
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_amp_gpu.py tests/test_fpn_gpu.py tests/test_gfl_gpu.py -m gpu -q > $O/${T}_pytest.log 2>&1; echo "$(el) pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -k "e8t2_b2 or base_b1" > $O/${T}_pytest_full.log 2>&1; echo "$(el) pytest fullsize rc=$? $(tail -1 $O/${T}_pytest_full.log)" | tee -a $S
echo "$(el) done" | tee -a $S
