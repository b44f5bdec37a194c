#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python scripts/slice_glue_audit.py > $O/r04c17_glue.txt 2> $O/r04c17_glue.err; echo "rc=$?"; head -75 $O/r04c17_glue.txt
