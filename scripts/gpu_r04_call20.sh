#!/bin/bash
# round 4, call 20: lanes per token of the router kernels, same-box A/B (parity on the variant first)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c20; S=$O/${T}_summary.txt; : > $S
SM3DET_HIP_LIB=$R/sm3det_amd/csrc/libsm3det_hip_lpt8.so timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -x -q > $O/${T}_pytest.log 2>&1
echo "pytest(lpt8) rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
for v in base lpt8 base lpt8; do
  L=$R/sm3det_amd/csrc/libsm3det_hip.so; [ $v = lpt8 ] && L=$R/sm3det_amd/csrc/libsm3det_hip_lpt8.so
  SM3DET_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-ops > $O/${T}_bench_$v.json 2> $O/${T}_bench_$v.err
  echo "$v $(python -c "import json;d=json.loads(open('$O/${T}_bench_$v.json').read().strip().splitlines()[-1]);k=d['kernels_ms_per_step'];print(d['ms_per_step'], k.get('moe_router_fwd'), k.get('moe_router_bwd'), k.get('moe_aux_loss_fwd'), k.get('moe_gate_prep_bwd'))")" | tee -a $S
done
