#!/bin/bash
# A/B of two builds of the library on the serialized training step's GEMM launches (same box, alternating runs)
# usage: scripts/ab_gemm.sh <other .so relative to the repo root>
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for L in "" "$R/$1"; do
    SM3DET_HIP_LIB=$L SM3_WGRAD_STREAM=0 python $R/scripts/gemm_shapes.py 2>/dev/null | grep "total gemm" | sed "s|^|[${L:-new}] |"
  done
done
