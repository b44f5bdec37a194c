#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for V in "" _f16_occ2 _f16_occ4; do
  SM3DET_HIP_LIB=$R/sm3det_amd/csrc/libsm3det_hip$V.so timeout 300 python scripts/gemm_sweep_amp.py --default-only --cold 2>&1 | grep -v "^JSON\|amdgpu.ids" > $O/c15_occ${V}.txt
  echo "lib$V cold: $(tail -1 $O/c15_occ${V}.txt)"
done
