#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
SM3_BENCH_NATIVE=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ops > $O/bench_kt.log 2>&1
python $R/scripts/graph_gaps.py /tmp/kt > $O/graph_gaps.txt 2>&1
head -40 $O/graph_gaps.txt
cd $R; python -m pytest tests/test_gfl_gpu.py -q -m gpu -x -k parallel_streams 2>&1 | tail -n 40
