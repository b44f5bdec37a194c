#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python scripts/slice_glue_audit2.py > $O/r04c19_glue.txt 2> $O/r04c19_glue.err; echo "rc=$?"; head -70 $O/r04c19_glue.txt
