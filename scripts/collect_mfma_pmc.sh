#!/bin/bash
# MFMA-utilisation PMC pass (SQ + GRBM counters only; no tracing besides --kernel-trace) on the serialized bench step and
# on the isolated GEMM sweep (calibration: a GEMM of known TFLOP/s).  usage: scripts/collect_mfma_pmc.sh <tag>
set -u
TAG=${1:-x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp SM3_WGRAD_STREAM=0
CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_m1 /tmp/pmc_m2
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_m1 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ops --no-graph > $O/${TAG}_mfma_bench.log 2>&1
D=$(dirname $(find /tmp/pmc_m1 -name "*counter_collection.csv" | head -1)); python $R/scripts/pmc_summary.py $D > $O/${TAG}_mfma_bench_top.txt 2>&1; cp $D/summary.json $O/${TAG}_mfma_bench_summary.json
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_m2 -o p -- python $R/scripts/gemm_one_shape.py > $O/${TAG}_mfma_gemm.log 2>&1
D=$(dirname $(find /tmp/pmc_m2 -name "*counter_collection.csv" | head -1)); python $R/scripts/pmc_summary.py $D > $O/${TAG}_mfma_gemm_top.txt 2>&1; cp $D/summary.json $O/${TAG}_mfma_gemm_summary.json
tail -3 $O/${TAG}_mfma_gemm.log
echo done
