#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06y; mkdir -p $O
full() {
env "${@:2}" SM3_BENCH_NATIVE=0 SM3_BENCH_OPS=full python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$O/full_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); fm=d.get('full_model') or {}
print('full $1', {k:v for k,v in fm.items() if 'ms' in k})" | tee -a $O/ab.txt
}
full p0 SM3_PAIR_DGRAD=0
full p4_13M SM3_PAIR_DGRAD=4 SM3_PAIR_MAX_OUTPUTS=13000000
full p4_7M SM3_PAIR_DGRAD=4 SM3_PAIR_MAX_OUTPUTS=7000000
full p2_13M SM3_PAIR_DGRAD=2 SM3_PAIR_MAX_OUTPUTS=13000000
full p2_all SM3_PAIR_DGRAD=2
full p0b SM3_PAIR_DGRAD=0
