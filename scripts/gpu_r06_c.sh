#!/bin/bash
# round 6, call C: tile x split-K sweep and per-workgroup trace of the one-set (3 workgroups per CU) bf16x3 kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O
cd $R
python scripts/gemm_b3_eval.py --splits > $O/gemm_b3_sweep.txt 2> $O/sweep.err
SM3DET_HIP_LIB=$R/sm3det_amd/csrc/libsm3det_hip_trace.so python scripts/gemm_trace.py $O/gemm_trace.npz > $O/trace.log 2>&1
python scripts/gemm_trace_analyse.py $O/gemm_trace.npz > $O/gemm_trace.txt 2>&1
tail -n 4 $O/gemm_b3_sweep.txt; tail -n 3 $O/gemm_trace.txt
