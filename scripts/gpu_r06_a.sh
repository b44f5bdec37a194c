#!/bin/bash
# round 6, call A: baseline on this box + the two cheap levers (weight-gradient side stream; one accumulator set = 3 WG/CU)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops"
U=$R/sm3det_amd/csrc/libsm3det_hip_b3_unsigned.so
for i in 1 2; do
$B > $O/base_$i.json 2> $O/base_$i.err
SM3_WGRAD_STREAM=1 $B > $O/wgs_$i.json 2> $O/wgs_$i.err
SM3DET_HIP_LIB=$U $B > $O/uns_$i.json 2> $O/uns_$i.err
SM3DET_HIP_LIB=$U SM3_WGRAD_STREAM=1 $B > $O/uns_wgs_$i.json 2> $O/uns_wgs_$i.err
done
for f in $O/*.json; do echo "$(basename $f) $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline'].get('gemm_ms_per_step'), d['roofline'].get('other_kernels_ms_per_step'))" 2>&1 | tail -1)"; done
