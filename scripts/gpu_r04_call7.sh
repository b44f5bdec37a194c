#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c7; S=$O/${T}_summary.txt; : > $S
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_roi_head_gpu.py tests/test_rpn_gpu.py -m gpu -q > $O/${T}_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/${T}_pytest.log)" | tee -a $S
grep -E "^FAILED|^ERROR" $O/${T}_pytest.log | head >> $S
timeout 300 python scripts/ops_quick.py > $O/${T}_ops_quick.txt 2>&1; echo "ops_quick rc=$?" | tee -a $S; cat $O/${T}_ops_quick.txt >> $S
rm -rf /tmp/prof_roi; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_roi -o roi -- python scripts/roi_bwd_probe.py tiled > $O/${T}_prof.log 2>&1; echo "rocprof rc=$?" | tee -a $S
find /tmp/prof_roi -name "*kernel_stats.csv" -exec cp {} $O/${T}_roi_kernel_stats.csv \;
head -9 $O/${T}_roi_kernel_stats.csv | cut -c1-160 >> $S
timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -k "amp" > $O/${T}_pytest_amp.log 2>&1; echo "pytest fullsize amp rc=$? $(tail -1 $O/${T}_pytest_amp.log)" | tee -a $S
