#!/bin/bash
# rocprofv3 evidence for the hot path (b) operators: per operator a kernel-trace + stats pass and two PMC passes
# (SQ instruction / cycle counters; FETCH_SIZE and WRITE_SIZE need a pass each: MI355X_MICROARCH.md PMC slots).
# usage: scripts/gpu_r03_ops_profile.sh <tag>  ->  gpurun_out/<tag>_ops/<op>_{stats.csv,sq.json,fetch.json,write.json}
set -u
TAG=${1:-ops}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_ops
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OPS="box_iou_rotated_2000x512 box_iou_rotated_2000x64 nms_rotated_2000 nms_rotated_10000 nms_8768 roi_align_rotated_fwd_nchw roi_align_rotated_fwd_nhwc roi_align_rotated_bwd_nchw roi_align_rotated_bwd_nhwc deform_conv2d_fwd deform_conv2d_bwd"
for OP in $OPS; do
  rm -rf /tmp/op_st
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/op_st -o p -- python $R/scripts/ops_profile.py $OP 5 > $O/${OP}_stats.log 2>&1
  find /tmp/op_st -name "*kernel_stats.csv" -exec cp {} $O/${OP}_stats.csv \;
  for P in "sq:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    N=${P%%:*}; C=${P#*:}
    rm -rf /tmp/op_pmc
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/op_pmc -o p -- python $R/scripts/ops_profile.py $OP 3 > $O/${OP}_$N.log 2>&1
    D=$(dirname $(find /tmp/op_pmc -name "*counter_collection.csv" | head -1))
    python $R/scripts/pmc_summary.py $D > /dev/null 2>&1 && cp $D/summary.json $O/${OP}_$N.json
  done
  echo "$OP done"
done
