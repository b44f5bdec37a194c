#!/bin/bash
# copy the judged summaries of one artifact collection (scripts/collect_artifacts.sh <tag>) from gpurun_out/ into profiles/<round>/
# usage: scripts/publish_profiles.sh <tag> <round>
set -eu
TAG=$1; RND=$2
cd "$(dirname "$0")/.."
P=profiles/$RND; G=gpurun_out
mkdir -p $P
for f in bench bench_e16t2 bench_SM3Det_convnext_t bench_SM3Det_convnext_b bench_SM3Det_convnext_b_fp32 \
         bench_under_rocprof_serial bench_under_rocprof_overlap bench_under_rocprof_serial_amp \
         pmc_traffic pmc_traffic_amp pmc_FETCH_SIZE_summary pmc_WRITE_SIZE_summary pmc_FETCH_SIZE_summary_amp \
         pmc_WRITE_SIZE_summary_amp mfma_bench_summary mfma_bench_summary_amp pmc_tcc_summary pmc_tcc_summary_amp; do
  [ -s $G/${TAG}_$f.json ] && cp $G/${TAG}_$f.json $P/$f.json
done
for f in kernel_stats_serial kernel_stats_overlap kernel_stats_serial_amp full_model_kernel_stats; do
  [ -s $G/${TAG}_$f.csv ] && cp $G/${TAG}_$f.csv $P/$f.csv
done
for f in pmc_FETCH_SIZE_top pmc_WRITE_SIZE_top mfma_bench_top pmc_FETCH_SIZE_top_amp pmc_WRITE_SIZE_top_amp mfma_bench_top_amp pmc_tcc_top pmc_tcc_top_amp summary graph_gaps; do
  [ -s $G/${TAG}_$f.txt ] && cp $G/${TAG}_$f.txt $P/$f.txt
done
for f in $G/fullsize_*.json; do [ -s $f ] && cp $f $P/; done
ls $P | wc -l
