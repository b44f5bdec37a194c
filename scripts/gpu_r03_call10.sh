#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
{
python scripts/amp_grad_debug.py full_e8t2_b2 amp forced 8
python scripts/amp_grad_debug.py full_base_b1 fp32 forced 6
python scripts/amp_grad_debug.py full_base_b1 amp forced 25
SM3_AMP_STORAGE=fp32 python scripts/amp_grad_debug.py full_base_b1 amp forced 12
SM3_DEBUG_LOSS_SCALE=1024 python scripts/amp_grad_debug.py full_base_b1 amp forced 12
} > $O/c10_debug.log 2>&1
tail -5 $O/c10_debug.log
