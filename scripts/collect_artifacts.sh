#!/bin/bash
# Round artifacts on the GPU box: full bench line, rocprofv3 kernel-trace stats (kernels serialized so the per-kernel
# averages are those of each kernel alone, and the overlapped production run), PMC passes for HBM traffic and MFMA
# utilisation (separate --pmc passes, --kernel-trace only: MI355X_MICROARCH.md), the AMP bench line.
# usage: scripts/collect_artifacts.sh <tag>   -> gpurun_out/<tag>_*
set -u
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 300 $O/${TAG}_bench.json
python $R/bench.py --amp --no-cpu-baseline --no-ops > $O/${TAG}_bench_amp.json 2> $O/${TAG}_bench_amp.err
for MODE in serial overlap; do
  rm -rf /tmp/prof_$MODE
  if [ $MODE = serial ]; then export SM3_WGRAD_STREAM=0; else export SM3_WGRAD_STREAM=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$MODE -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ops > $O/${TAG}_rocprof_$MODE.log 2>&1
  find /tmp/prof_$MODE -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_$MODE.csv \;
  grep -h '^{' $O/${TAG}_rocprof_$MODE.log | tail -1 > $O/${TAG}_bench_under_rocprof_$MODE.json
done
export SM3_WGRAD_STREAM=0
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ops --no-graph > $O/${TAG}_pmc_$C.log 2>&1
  D=$(dirname $(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1))
  python $R/scripts/pmc_summary.py $D > $O/${TAG}_pmc_${C}_top.txt 2>&1
  cp $D/summary.json $O/${TAG}_pmc_${C}_summary.json
done
CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_m1
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_m1 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ops --no-graph > $O/${TAG}_mfma_bench.log 2>&1
D=$(dirname $(find /tmp/pmc_m1 -name "*counter_collection.csv" | head -1)); python $R/scripts/pmc_summary.py $D > $O/${TAG}_mfma_bench_top.txt 2>&1; cp $D/summary.json $O/${TAG}_mfma_bench_summary.json
python $R/scripts/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE_summary.json $O/${TAG}_pmc_WRITE_SIZE_summary.json $O/${TAG}_bench.json $O/${TAG}_mfma_bench_summary.json > $O/${TAG}_pmc_traffic.json 2> $O/${TAG}_pmc_traffic.err
echo done
