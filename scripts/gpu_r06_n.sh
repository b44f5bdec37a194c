#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops"
for i in 1 2; do
for v in "" b3_slp aux_temporal; do
  L=""; [ -n "$v" ] && L=$R/sm3det_amd/csrc/libsm3det_hip_$v.so
  SM3DET_HIP_LIB=$L $B > $O/${v:-default}_$i.json 2> $O/${v:-default}_$i.err
done
done
for f in $O/*.json; do echo "$(basename $f) $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline'].get('gemm_ms_per_step'), d['roofline'].get('other_kernels_ms_per_step'))" 2>&1 | tail -1)"; done
