#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_amp_gpu.py -m gpu -q -x > $O/c12_pytest_amp.log 2>&1; echo "pytest amp rc=$? $(tail -1 $O/c12_pytest_amp.log)" | tee $O/c12_summary.txt
timeout 600 python scripts/gemm_sweep_amp.py > $O/c12_sweep_occ3.txt 2>&1; tail -1 $O/c12_sweep_occ3.txt | tee -a $O/c12_summary.txt
SM3DET_HIP_LIB=$R/sm3det_amd/csrc/libsm3det_hip_f16_occ2.so timeout 600 python scripts/gemm_sweep_amp.py > $O/c12_sweep_occ2.txt 2>&1; tail -1 $O/c12_sweep_occ2.txt | tee -a $O/c12_summary.txt
python bench.py --config SM3Det_convnext_t --no-cpu-baseline --no-ops > $O/c12_bench_amp.json 2>$O/c12_bench_amp.err; head -c 400 $O/c12_bench_amp.json | tee -a $O/c12_summary.txt
timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -k "amp and base" > $O/c12_pytest_full.log 2>&1; echo "pytest amp fullsize base rc=$? $(tail -1 $O/c12_pytest_full.log)" | tee -a $O/c12_summary.txt
