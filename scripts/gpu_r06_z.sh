#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06z; mkdir -p $O
for P in 0 4 0 4; do
SM3_PAIR_DGRAD=$P SM3_BENCH_NATIVE=0 SM3_BENCH_OPS=slice python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>$O/slice_$P.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d.get('ops_us') or {}
print('slice PAIR=$P', d['ms_per_step'], {k:v for k,v in o.items() if 'slice' in k and not isinstance(v, dict)})" | tee -a $O/ab.txt
done
