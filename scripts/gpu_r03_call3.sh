#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/c3_summary.txt
: > $S
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $S; }
timeout 300 python scripts/amp_debug.py > $O/c3_amp_debug.log 2>&1
stamp "amp debug rc=$?"
timeout 900 python -m pytest tests/test_losses_gpu.py tests/test_amp_gpu.py tests/test_deform_conv_gpu.py -m gpu -q -rf > $O/c3_pytest_a.log 2>&1
stamp "pytest A (losses, amp, deform) rc=$? $(tail -1 $O/c3_pytest_a.log)"
timeout 600 python - > $O/c3_deform.log 2>&1 <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
us = {}
import torch, time
from sm3det_amd.mmcv_deform_conv import deform_conv2d
xd = torch.randn(2, 256, 128, 128, device='cuda', requires_grad=True)
od = (torch.randn(2, 18, 128, 128, device='cuda') * 2).requires_grad_(True)
wd = (torch.randn(256, 256, 3, 3, device='cuda') * 0.02).requires_grad_(True)
yd = deform_conv2d(xd, od, wd, 1, 1, 1, 1, 1, False, 2)
gd = torch.randn_like(yd)
def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('fwd us', t(lambda: deform_conv2d(xd, od, wd, 1, 1, 1, 1, 1, False, 2)))
print('bwd us', t(lambda: torch.autograd.grad(yd, (xd, od, wd), gd, retain_graph=True)))
PY
stamp "deform timing: $(cat $O/c3_deform.log | tr '\n' ' ')"
cat $S
cd $R
timeout 900 python bench.py --no-cpu-baseline > $O/c3_bench.json 2> $O/c3_bench.err
echo "[$(( $(date +%s) - t0 )) s] bench rc=$? $(head -c 200 $O/c3_bench.json)" >> $S
grep -c "capture failed" $O/c3_bench.err >> $S
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -rf -k amp > $O/c3_pytest_b.log 2>&1
echo "[$(( $(date +%s) - t0 )) s] fullsize amp rc=$? $(tail -1 $O/c3_pytest_b.log)" >> $S
for C in SM3Det_convnext_t SM3Det_convnext_b; do
  timeout 600 python bench.py --config $C --no-ops --no-cpu-baseline > $O/c3_bench_$C.json 2> $O/c3_bench_$C.err
  echo "[$(( $(date +%s) - t0 )) s] bench $C $(head -c 230 $O/c3_bench_$C.json)" >> $S
done
