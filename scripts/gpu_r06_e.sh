#!/bin/bash
# round 6, call E: phase ablations of the one-set bf16x3 loop (cold-operand sweep of the step's shapes, default configuration)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O
cd $R
for v in "" abl_nocvt abl_noload abl_nostore abl_noepi abl_loop_only_mfma abl_nomfma; do
  L=""; [ -n "$v" ] && L=$R/sm3det_amd/csrc/libsm3det_hip_$v.so
  SM3DET_HIP_LIB=$L python scripts/gemm_b3_eval.py --no-sweep > $O/eval_${v:-complete}.txt 2>&1
  echo "${v:-complete}: $(grep 'summed over' $O/eval_${v:-complete}.txt)"
done
