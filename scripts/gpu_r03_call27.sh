#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_rpn_gpu.py tests/test_detector_slice_gpu.py -m gpu -q -x 2>&1 | tail -3
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from sm3det_amd import _lib
L = _lib.lib()
for n in (196608, 49152, 12288):
    sc = torch.rand(n, device='cuda')
    k = 2000
    order = torch.empty(k, dtype=torch.long, device='cuda'); full = torch.empty(n, dtype=torch.long, device='cuda')
    nb = L.sm3_topk_desc_workspace_bytes(n); ws = _lib.workspace(nb, sc.device)
    nb2 = L.sm3_argsort_desc_workspace_bytes(n); ws2 = _lib.workspace(nb2, sc.device)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    a = t(lambda: L.sm3_topk_desc_f32(_lib.ptr(sc), n, k, _lib.ptr(order), _lib.ptr(ws), nb, _lib.stream_ptr()))
    b = t(lambda: L.sm3_argsort_desc_f32(_lib.ptr(sc), n, _lib.ptr(full), _lib.ptr(ws2), nb2, _lib.stream_ptr()))
    print(f'n={n}: topk {a:.1f} us, argsort {b:.1f} us')
PY
