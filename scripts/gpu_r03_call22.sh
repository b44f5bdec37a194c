#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/c22_ops; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ref_wrappers.py tests/test_roi_head_gpu.py -m gpu -q > $O/c22_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/c22_pytest.log)"
for OP in roi_align_rotated_fwd_nchw roi_align_rotated_fwd_nhwc; do
  rm -rf /tmp/op_st
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/op_st -o p -- python $R/scripts/ops_profile.py $OP 5 > /dev/null 2>&1
  find /tmp/op_st -name "*kernel_stats.csv" -exec cp {} $O/c22_ops/${OP}_stats.csv \;
done
python scripts/ops_pmc_summary.py $O/c22_ops | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if k.startswith('_'): continue
    print(k, {n:x['us_per_call'] for n,x in v['kernels'].items()})
"
