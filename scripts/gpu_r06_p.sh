#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06p; O=gpurun_out/r06p
for LS in 0 2 0 2; do
  SM3_BENCH_NATIVE=0 SM3_LEVEL_STREAMS=$LS SM3_BENCH_OPS=full python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$O/full_$LS.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); fm=d.get('full_model') or {}
print('LS=$LS', {k:v for k,v in fm.items() if 'ms' in k})"
done
