#!/bin/bash
# round 4, call 18: XCD-aware tile order of the fused DeformConv2d forward -- parity, timing, counted traffic
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c18; S=$O/${T}_summary.txt; : > $S
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_deform_conv_gpu.py -m gpu -x -q > $O/${T}_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
timeout 300 python scripts/ops_quick.py 2>/dev/null | grep deform | tee -a $S
cd /tmp
for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  N=${P%%:*}; C=${P#*:}
  rm -rf /tmp/op_pmc
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/op_pmc -o p -- python $R/scripts/ops_profile.py deform_conv2d_fwd 3 > $O/${T}_$N.log 2>&1
  D=$(dirname $(find /tmp/op_pmc -name "*counter_collection.csv" | head -1))
  python $R/scripts/pmc_summary.py $D 2>/dev/null | grep deform_conv_fwd_fused | tee -a $S
done
