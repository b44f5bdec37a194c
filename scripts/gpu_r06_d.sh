#!/bin/bash
# round 6, call D: SQ / LDS counters of one isolated bf16x3 GEMM shape (stage-2 dense FC1 forward)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM"
P3="SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
for sh in "nt 8192 1536 384" "nn 8192 384 1536" "tn 1536 384 8192"; do
  tag=$(echo $sh | tr ' ' '_')
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1)); rm -rf /tmp/pmc_$i
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python $R/scripts/gemm_one_shape2.py $sh > $O/${tag}_p$i.log 2>&1
    python $R/scripts/pmc_sum.py /tmp/pmc_$i > $O/${tag}_p$i.txt 2>&1
  done
  cat $O/${tag}_p1.txt $O/${tag}_p2.txt $O/${tag}_p3.txt; tail -n 1 $O/${tag}_p1.log
done
