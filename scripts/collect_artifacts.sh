#!/bin/bash
# Round artifacts on the GPU box: the default bench line (with cpu_baseline and the operator micro-bench), the lines of
# the other BASELINE.json configs, rocprofv3 kernel-trace stats (kernels serialized so the per-kernel averages are those
# of each kernel alone, and the overlapped production run), PMC passes for HBM traffic and MFMA utilisation (separate
# --pmc passes, --kernel-trace only: MI355X_MICROARCH.md), the same for the AMP config.
# usage: scripts/collect_artifacts.sh <tag>   -> gpurun_out/<tag>_*
set -u
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T0=$(date +%s)
say() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/${TAG}_summary.txt; }
python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
say "bench default: $(head -c 420 $O/${TAG}_bench.json)"
export SM3_BENCH_NATIVE=0  # only the default line above carries the native-fp32-MFMA companion run
for CFG in e16t2 SM3Det_convnext_t SM3Det_convnext_b; do
  python $R/bench.py --config $CFG --no-cpu-baseline --no-ops > $O/${TAG}_bench_$CFG.json 2> $O/${TAG}_bench_$CFG.err
  say "bench $CFG: $(head -c 420 $O/${TAG}_bench_$CFG.json)"
done
python $R/bench.py --config SM3Det_convnext_b --fp32 --no-cpu-baseline --no-ops > $O/${TAG}_bench_SM3Det_convnext_b_fp32.json 2> $O/${TAG}_bench_SM3Det_convnext_b_fp32.err
say "bench convnext_b fp32: $(head -c 420 $O/${TAG}_bench_SM3Det_convnext_b_fp32.json)"

profile() {  # profile <suffix> <bench args...>: stats (serial [+ overlap]) and the PMC passes of one bench command
  local SFX=$1; shift
  local MODES="serial"; [ -z "$SFX" ] && MODES="serial overlap"
  for MODE in $MODES; do
    rm -rf /tmp/prof_$MODE
    # serial: no concurrent partners (per-kernel durations of each kernel ALONE); overlap: the production schedule (SM3_PAIR_DGRAD default)
    if [ $MODE = serial ]; then export SM3_PAIR_DGRAD=0; else unset SM3_PAIR_DGRAD; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$MODE -o p -- python $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-ops > $O/${TAG}_rocprof_$MODE$SFX.log 2>&1
    find /tmp/prof_$MODE -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_$MODE$SFX.csv \;
    grep -h '^{' $O/${TAG}_rocprof_$MODE$SFX.log | tail -1 > $O/${TAG}_bench_under_rocprof_$MODE$SFX.json
  done
  export SM3_PAIR_DGRAD=0
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-ops --no-graph > $O/${TAG}_pmc_$C$SFX.log 2>&1
    D=$(dirname $(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1))
    python $R/scripts/pmc_summary.py $D > $O/${TAG}_pmc_${C}_top$SFX.txt 2>&1
    cp $D/summary.json $O/${TAG}_pmc_${C}_summary$SFX.json
  done
  CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  rm -rf /tmp/pmc_m1
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_m1 -o p -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-ops --no-graph > $O/${TAG}_mfma_bench$SFX.log 2>&1
  D=$(dirname $(find /tmp/pmc_m1 -name "*counter_collection.csv" | head -1)); python $R/scripts/pmc_summary.py $D > $O/${TAG}_mfma_bench_top$SFX.txt 2>&1; cp $D/summary.json $O/${TAG}_mfma_bench_summary$SFX.json
  # L2 hit rate and memory-side requests per kernel (does the split-K slab round trip leave the L2?)
  rm -rf /tmp/pmc_tcc
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d /tmp/pmc_tcc -o p -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-ops --no-graph > $O/${TAG}_pmc_tcc$SFX.log 2>&1
  D=$(dirname $(find /tmp/pmc_tcc -name "*counter_collection.csv" | head -1)); python $R/scripts/pmc_summary.py $D > $O/${TAG}_pmc_tcc_top$SFX.txt 2>&1; cp $D/summary.json $O/${TAG}_pmc_tcc_summary$SFX.json
  unset SM3_PAIR_DGRAD
}
profile ""
# idle time / concurrency of one replayed step of the production schedule
rm -rf /tmp/kt; SM3_BENCH_NATIVE=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ops > $O/${TAG}_kt.log 2>&1
python $R/scripts/graph_gaps.py /tmp/kt > $O/${TAG}_graph_gaps.txt 2>&1
# the full detector alone (the third named workload): kernel statistics of SM3_BENCH_OPS=full
rm -rf /tmp/prof_full
SM3_BENCH_OPS=full rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_rocprof_full.log 2>&1
find /tmp/prof_full -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_full_model_kernel_stats.csv \;
say "full model stats: $(wc -l < $O/${TAG}_full_model_kernel_stats.csv) kernels"
python $R/scripts/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE_summary.json $O/${TAG}_pmc_WRITE_SIZE_summary.json $O/${TAG}_bench.json $O/${TAG}_mfma_bench_summary.json > $O/${TAG}_pmc_traffic.json 2> $O/${TAG}_pmc_traffic.err
say "fp32 profile done: $(grep -A3 gemm_family\" $O/${TAG}_pmc_traffic.json | tr -d '\n ')"
profile _amp --config SM3Det_convnext_t
python $R/scripts/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE_summary_amp.json $O/${TAG}_pmc_WRITE_SIZE_summary_amp.json $O/${TAG}_bench_SM3Det_convnext_t.json $O/${TAG}_mfma_bench_summary_amp.json > $O/${TAG}_pmc_traffic_amp.json 2> $O/${TAG}_pmc_traffic_amp.err
say "amp profile done: $(grep -A3 gemm_family\" $O/${TAG}_pmc_traffic_amp.json | tr -d '\n ')"
