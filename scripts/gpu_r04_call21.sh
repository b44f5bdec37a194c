#!/bin/bash
# round 4, call 21: batched index / vector loads in the MoE combine kernels, same-box A/B (parity first)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c21; S=$O/${T}_summary.txt; : > $S
timeout 900 python -m pytest tests/test_backbone_gpu.py -m gpu -x -q > $O/${T}_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
for v in base new base new; do
  L=$R/sm3det_amd/csrc/libsm3det_hip.so; [ $v = base ] && L=$R/sm3det_amd/csrc/libsm3det_hip_base.so
  SM3DET_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-ops > $O/${T}_bench_$v.json 2> $O/${T}_bench_$v.err
  echo "$v $(python -c "import json;d=json.loads(open('$O/${T}_bench_$v.json').read().strip().splitlines()[-1]);k=d['kernels_ms_per_step'];print(d['ms_per_step'], k.get('moe_combine_bwd'), k.get('moe_combine_fwd'), k.get('moe_gather_add'))")" | tee -a $S
done
