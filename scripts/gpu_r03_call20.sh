#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/c20_ops; cd $R
for OP in box_iou_rotated_2000x512 box_iou_rotated_2000x64 nms_rotated_2000 nms_rotated_10000; do
  rm -rf /tmp/op_st
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/op_st -o p -- python $R/scripts/ops_profile.py $OP 5 > /dev/null 2>&1
  find /tmp/op_st -name "*kernel_stats.csv" -exec cp {} $O/c20_ops/${OP}_stats.csv \;
  rm -rf /tmp/op_pmc
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/op_pmc -o p -- python $R/scripts/ops_profile.py $OP 3 > /dev/null 2>&1
  D=$(dirname $(find /tmp/op_pmc -name "*counter_collection.csv" | head -1))
  python $R/scripts/pmc_summary.py $D > /dev/null 2>&1 && cp $D/summary.json $O/c20_ops/${OP}_sq.json
done
python scripts/ops_pmc_summary.py $O/c20_ops
