"""Which Python lines launch the torch (non-library) kernels of the headline train step?

    python scripts/tail_sources.py [--config NAME] [--amp]        (GPU box)

One eager step of bench.py's headline workload under torch.profiler (with_stack), device kernels of the aten operators grouped
by the innermost frame inside this repository.  Used to hunt the copy / fill / elementwise launches of the step's tail
(profiles/r06/tail_sources.txt).
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default=bench.DEFAULT_CONFIG)
    ap.add_argument('--amp', action='store_true')
    args = ap.parse_args()
    from sm3det_amd import _lib
    from sm3det_amd.data_parallel import BucketedGradReducer
    from sm3det_amd.optim import build_optimizer
    _lib.lib()
    cfg_entry = bench.load_config(args.config)
    net = bench.build_model(args.config).cuda().train()
    if args.amp:
        from sm3det_amd import amp
        amp.wrap_fp16_model(net)
    params = [p for p in net.parameters() if p.requires_grad]
    reducer = BucketedGradReducer(params, bucket_mb=64.0)
    opt = build_optimizer({'backbone': net}, cfg_entry['optimizer'], cfg_entry['optimizer_config'],
                          loss_scale='dynamic' if args.amp else None)
    x = torch.randn(bench.BATCH, 3, bench.RES, bench.RES).cuda()
    proj = None

    def step():
        nonlocal proj
        reducer.zero_grad()
        outs, gl = net(x, ['single'])
        if proj is None:
            proj = [torch.randn_like(o) for o in outs]
        loss = bench.loss_fn(outs, gl, proj)
        opt.scale(loss).backward()
        reducer.finalize()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    groups = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith('aten::') or ev.self_device_time_total <= 0:
            continue
        where = 'autograd engine / no repo frame'
        for fr in ev.stack:
            if ROOT in fr and 'tail_sources' not in fr:
                where = fr.replace(ROOT + '/', '')
                break
        if where.startswith('autograd'):
            for fr in ev.stack:
                if 'tail_sources' in fr:
                    where = fr.replace(ROOT + '/', '')
                    break
        k = (ev.name, where)
        groups[k][0] += 1
        groups[k][1] += ev.self_device_time_total
    tot_n = tot_t = 0
    for (name, where), (n, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print(f'{n:4d} x {t / max(n, 1):7.1f} us = {t:8.1f} us  {name:28s} {where}')
        tot_n += n
        tot_t += t
    print(f'total {tot_n} aten device launches, {tot_t / 1e3:.3f} ms')


if __name__ == '__main__':
    main()
