#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -k amp > $O/c11_pytest.log 2>&1; echo "pytest amp fullsize rc=$? $(tail -1 $O/c11_pytest.log)" | tee $O/c11_summary.txt
bash $R/scripts/gpu_r03_ops_profile.sh c11 >> $O/c11_summary.txt 2>&1
tail -3 $O/c11_summary.txt
