#!/bin/bash
# HBM traffic of the AMP (fp16-operand) training step: two --pmc passes (FETCH_SIZE, WRITE_SIZE), --kernel-trace only
TAG=${1:-amp}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp SM3_WGRAD_STREAM=0
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmca_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmca_$C -o p -- python $R/bench.py --amp --steps 2 --warmup 1 --no-cpu-baseline --no-ops --no-graph > $O/${TAG}_pmc_$C.log 2>&1
  D=$(dirname $(find /tmp/pmca_$C -name "*counter_collection.csv" | head -1))
  python $R/scripts/pmc_summary.py $D > $O/${TAG}_pmc_${C}_top.txt 2>&1
  cp $D/summary.json $O/${TAG}_pmc_${C}_summary.json
done
python - <<EOF
import json
f=json.load(open("$O/${TAG}_pmc_FETCH_SIZE_summary.json")); w=json.load(open("$O/${TAG}_pmc_WRITE_SIZE_summary.json"))
def find(d,key):
    for k,v in d.items():
        if key in k: return k,v
    return None,None
for key in ('gemm_f32_kernel','splitk_reduce'):
    kf,vf=find(f,key); kw,vw=find(w,key)
    print(key, kf, vf, vw)
EOF
