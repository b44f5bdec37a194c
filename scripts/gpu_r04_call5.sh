#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c5; S=$O/${T}_summary.txt; : > $S
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_deform_conv_gpu.py tests/test_roi_head_gpu.py -m gpu -q > $O/${T}_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
grep -E "^FAILED|^ERROR" $O/${T}_pytest.log | head >> $S
timeout 300 python scripts/ops_quick.py > $O/${T}_ops_quick.txt 2>&1; echo "ops_quick rc=$?" | tee -a $S; cat $O/${T}_ops_quick.txt >> $S
