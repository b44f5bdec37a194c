#!/bin/bash
# phase ablations of the fp16-operand GEMM loop, warm (one argument set) and cold (six rotating sets)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for V in "" _abl_noload _abl_nostore _abl_nomfma _abl_noepi _abl_loop_only_mfma; do
  for MODE in "" "--cold"; do
    SM3DET_HIP_LIB=$R/sm3det_amd/csrc/libsm3det_hip$V.so timeout 300 python scripts/gemm_sweep_amp.py --default-only $MODE 2>&1 | grep -v "^JSON\|amdgpu.ids" > $O/c13_abl${V}${MODE#--}.txt
    echo "lib$V $MODE: $(tail -1 $O/c13_abl${V}${MODE#--}.txt)" | tee -a $O/c13_summary.txt
  done
done
