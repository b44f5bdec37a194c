#!/bin/bash
# round 6, call J: weight-gradient side stream on/off with the one-set GEMMs (A/B, alternating)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06j; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops"
for i in 1 2 3; do
SM3_WGRAD_STREAM=0 $B > $O/off_$i.json 2> $O/off_$i.err
SM3_WGRAD_STREAM=1 $B > $O/on_$i.json 2> $O/on_$i.err
done
SM3_WGRAD_STREAM=0 $B --config SM3Det_convnext_t > $O/amp_off.json 2> $O/amp_off.err
SM3_WGRAD_STREAM=1 $B --config SM3Det_convnext_t > $O/amp_on.json 2> $O/amp_on.err
SM3DET_HIP_LIB=$R/sm3det_amd/csrc/libsm3det_hip_f16_spills.so SM3_WGRAD_STREAM=0 $B --config SM3Det_convnext_t > $O/amp_spills_off.json 2> $O/amp_spills_off.err
SM3_WGRAD_STREAM=0 $B --config SM3Det_convnext_b > $O/ampb_off.json 2> $O/ampb_off.err
SM3_WGRAD_STREAM=1 $B --config SM3Det_convnext_b > $O/ampb_on.json 2> $O/ampb_on.err
for f in $O/*.json; do echo "$(basename $f) $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline'].get('gemm_ms_per_step'), d['roofline'].get('other_kernels_ms_per_step'))" 2>&1 | tail -1)"; done
