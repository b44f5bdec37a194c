#!/bin/bash
# the 1-rank RCCL control-flow run of tests/test_dp_rccl_gpu.py by hand: prints loss / config for each setting given
# usage: scripts/dist1.sh "<ENV=.. ENV=..|-> [bench flags]" ...
R=$GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 SM3_BENCH_RES=512
for cfg in "$@"; do
  envs=$(echo "$cfg" | tr ' ' '\n' | grep '=' | tr '\n' ' ')
  flags=$(echo "$cfg" | tr ' ' '\n' | grep -v '=' | tr '\n' ' ')
  env $envs python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-ops --no-cpu-baseline $flags 2>/tmp/err.txt | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$cfg', '| loss', d['loss'], 'split', d['config'].get('split_backward'), 'graph', d['config'].get('hip_graph'), d['config'].get('dist_backend'))"
  grep -i "capture failed\|error\|trace" /tmp/err.txt | head -12
done
