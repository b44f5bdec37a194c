#!/bin/bash
# last GPU call of a round: artifact collection first, then as many of the parity tests of the touched kernels as the remaining
# budget allows.  usage: scripts/gpu_final.sh <tag> <budget seconds>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$1
T0=$(date +%s)
timeout 330 bash scripts/collect_artifacts.sh $1 2>&1 | tail -9
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2
LEFT=$(( $2 - ($(date +%s) - T0) ))
echo "[tests: $LEFT s left]"
[ $LEFT -gt 60 ] && timeout $LEFT python -m pytest tests/test_backbone_gpu.py tests/test_amp_gpu.py tests/test_graph_replay_gpu.py tests/test_fullsize_gpu.py -q -k "not fullsize or e8t2_b2 or base_b1" --durations=6 2>&1 | tail -14 | tee gpurun_out/$1/final_tests.txt
