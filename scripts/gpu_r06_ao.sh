#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ao; mkdir -p $O
for P in 0 4; do
  SM3_PAIR_DGRAD=$P SM3_BENCH_TRACE=1 SM3_BENCH_NATIVE=0 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-ops > $O/out_$P.json 2> $O/trace_$P.txt
  grep "\[trace\]" $O/trace_$P.txt | sed -n '1p;20p;40p;60p' | sed "s/^/PAIR=$P /"
  grep -c suspicious $O/trace_$P.txt
done
