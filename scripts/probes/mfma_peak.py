"""Run scripts/probes/mfma_peak.hip: the sustained fp32 MFMA rate of the chip with random / zero operands at 1-3 waves per
SIMD, and what s_memtime counts.    python scripts/probes/mfma_peak.py > gpurun_out/mfma_peak.txt"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'mfma_peak.so')
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, 'mfma_peak.hip')):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', SO,
                           os.path.join(HERE, 'mfma_peak.hip')])
L = ctypes.CDLL(SO)
L.mfma_peak_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda')
iters = 20000
for data in ('randn', 'zeros'):
    src = (torch.randn(1 << 20, device=dev) if data == 'randn' else torch.zeros(1 << 20, device=dev))
    for wgs_per_cu in (1, 2, 3):
        blocks = 256 * wgs_per_cu
        dst = torch.empty(blocks * 256, device=dev)
        st = torch.zeros(blocks * 4 * 4, dtype=torch.int64, device=dev)
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.mfma_peak_launch(src.data_ptr(), dst.data_ptr(), st.data_ptr(), blocks, iters, torch.cuda.current_stream().cuda_stream)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0
        ms = e0.elapsed_time(e1)
        flops = blocks * 4 * iters * 8 * (2.0 * 32 * 32 * 2)
        s = st.view(-1, 4).cpu().double()
        ticks, rt = (s[:, 1] - s[:, 0]), (s[:, 3] - s[:, 2])
        per_mfma = (ticks / (iters * 8)).median().item()
        mhz = (ticks / rt * 100.0).median().item()
        wall_loop_us = (rt / 100.0).median().item()
        print(f'{data:6s} {wgs_per_cu} WG/CU ({wgs_per_cu} waves/SIMD): launch {ms * 1e3:9.1f} us -> {flops / ms / 1e9:7.1f} TFLOP/s | '
              f's_memtime ticks per MFMA per wave {per_mfma:6.2f} | s_memtime rate {mhz:7.1f} MHz | median loop {wall_loop_us:9.1f} us '
              f'-> {iters * 8 * wgs_per_cu * 64 / wall_loop_us:7.1f} MFMA-cycles/us per SIMD (= shader MHz if issue-bound)', flush=True)
