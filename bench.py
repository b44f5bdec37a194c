#!/usr/bin/env python
"""bench.py -- headline benchmark of the SM3Det hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--amp]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W                  (N>1, one rank per GPU over RCCL)

`python bench.py --gpus N` with N > 1 and no torchrun environment launches its own N ranks (it re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, the launch line of the reference's
tools/dist_train.sh) and relays rank 0's JSON line, so both command forms work.  `--config` selects one of the BASELINE.json
configurations from sm3det_amd/configs/baseline_configs.json (generated from the reference's own config files by
scripts/make_bench_configs.py): main_SM3Det (#2, the default and the headline), SM3Det_convnext_t (#3, AMP), e16t2 (#4),
SM3Det_convnext_b (#5, AMP), simple_joint (#1, no MoE); a config whose file sets `fp16 = dict(loss_scale='dynamic')` runs
the AMP data path, `--amp` forces it on any config.

Metric (BASELINE.json): train imgs/sec, SM3Det ConvNeXt-T e8t2 @1024^2, bs2/GPU.
Workload timed here (config.workload): one TRAINING STEP of the hot path = the `main_SM3Det.py` backbone
(ConvNeXt_moe_MultiInput, ConvNeXt-T, 8 experts top-2, MoE in stages 1-3: 9 MoE + 9 dense blocks, drop_path 0.1,
noisy gating) forward + backward in fp32 on a synthetic (2,3,1024,1024) batch per GPU, gradient all-reduce across
ranks (bucketed) and the optimizer step (global-norm clip 35 + AdamW with one lr per parameter, one fused launch).
The workload definition is the same every round.  The pieces either side of the backbone that have been built since
(MultitaskFPN, Oriented-RPN tower + proposal glue, fused multi-level RoI extractor, Shared2FC head) and
the rotated-detection operators of hot path (b) are reported as per-op timings in `ops_us`, outside `value`.

Printed JSON line (rank 0): the contract fields + `roofline` (dominant kernel = fp32 MFMA GEMM family, per-launch
HIP events on the launch stream) + `cpu_baseline` (the CPU oracle timed on this host, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# ROCm 7.2's hipGraph "packet capture" fast path (linear graphs) does not keep memset nodes ordered with the kernels
# around them on replay: torch's multi-block reductions (semaphore memset) returned a stale loss and, before the library
# switched to a zero-fill kernel, gradient buffers came back unzeroed.  The flag is read when the HIP runtime loads, so it
# is set before torch is imported; replay speed is unchanged (20.18 vs 20.22 ms per step measured both ways).
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MI355X_FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
MI355X_HBM_PEAK_GBS = 8000.0  # same guide: HBM3E ~8 TB/s
MI355X_BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), measured 2495
# bf16x3 form of the fp32 GEMMs: six bf16 products per fp32 product -> fp32-equivalent ceiling of the matrix pipe
MI355X_BF16X3_PEAK_TFLOPS = MI355X_BF16_MFMA_PEAK_TFLOPS / 6.0
MI355X_F16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense fp16 MFMA rate = the bf16 one (v_mfma_f32_32x32x16_f16)
BATCH = 2
RES = int(os.environ.get('SM3_BENCH_RES', '1024'))  # 1024 = BASELINE config; the override is a debugging aid only
CONFIG_FILE = os.path.join(ROOT, 'sm3det_amd', 'configs', 'baseline_configs.json')
DEFAULT_CONFIG = 'main_SM3Det'  # BASELINE.json config #2: the configuration `metric` is quoted on


def load_config(name):
    """one entry of sm3det_amd/configs/baseline_configs.json: the `model` / `fp16` / `optimizer` dicts of a reference
    config file, exactly as mmcv.Config.fromfile yields them (scripts/make_bench_configs.py; the reference tree itself is
    absent on the GPU box)"""
    with open(CONFIG_FILE) as f:
        cfgs = json.load(f)
    if name not in cfgs:
        raise SystemExit(f'--config {name!r}: choose from {sorted(cfgs)}')
    return cfgs[name]


def backbone_cfg(name):
    bb = dict(load_config(name)['model']['backbone'])
    bb.pop('init_cfg', None)  # Pretrained checkpoint path: no checkpoints exist here, weights are random-init
    return bb


def build_model(name=DEFAULT_CONFIG):
    """the backbone of reference config `name`, built through the registry from the config file's own dict"""
    from sm3det_amd import convnext_moe  # noqa: F401  (registers the classes)
    from sm3det_amd.registry import MODELS
    torch.manual_seed(0)
    net = MODELS.build(backbone_cfg(name))
    with torch.no_grad():  # layer scale 1e-6 would hide the FFN/MoE branch numerically; use O(1) like a trained net
        for n, p in net.named_parameters():
            if n.endswith('gamma'):
                p.fill_(1.0)
    return net


def describe_backbone(net):
    blocks = [b for st in net.stages for b in st]
    n_moe = sum(1 for b in blocks if b.MoE_cfg is not None)
    arch = {tuple(v['channels']): k for k, v in type(net).arch_settings.items() if v['depths'] == list(net.depths)}
    moe = next((b.ffn for b in blocks if b.MoE_cfg is not None), None)
    return dict(arch=arch.get(tuple(net.channels), str(net.channels)), moe_blocks=n_moe, dense_blocks=len(blocks) - n_moe,
                experts=(moe.num_experts if moe is not None else 0), top_k=(moe.k if moe is not None else 0),
                params_m=round(sum(p.numel() for p in net.parameters()) / 1e6, 2))


def loss_fn(outs, gate_loss, proj):
    loss = gate_loss
    for o, r in zip(outs, proj):
        loss = loss + (o * r).sum() * 1e-4
    return loss


CPU_THREADS = (8, 16, 32, 64)


def _cpu_worker(kind, cfg_name=DEFAULT_CONFIG):
    """Runs in a SUBPROCESS with OMP_NUM_THREADS fixed before torch is imported (resizing torch's thread pool inside
    a live process stalled on the 128-thread host in round 1).  `step`: the CPU oracle (oracle/moe_oracle.py, the
    restatement of the reference module that tests pin to it) doing the bench's training step (fwd+bwd, fp32) on
    1x3x1024x1024: 1 warm-up at 512^2, then 3 timed full-size steps, no pixel scaling.  `ops`: the reference's OWN
    CPU operators (oracle/_ref, compiled from /root/reference by oracle/build_ref.py) on the shapes of `ops_us`."""
    out = {'threads': torch.get_num_threads()}
    if kind == 'step':
        from oracle import ref_moe
        if ref_moe.available():
            # the REFERENCE backbone module itself (mmrotate/models/backbones/convnext_moe.py:794-820, imported unmodified
            # by oracle/ref_moe.py: from /root/reference in the build container, from the bytecode oracle/build_ref.py
            # compiled into oracle/_ref/pyc on the GPU box) doing the same training step: kind = "reference"
            bcfg = backbone_cfg(cfg_name)
            bcfg.pop('type', None)
            torch.manual_seed(0)
            ref = ref_moe.build_reference_backbone(**bcfg)
            ref.train()  # (the reference's train() override returns None)
            g = torch.Generator().manual_seed(0)

            def run_ref(b, res):
                x = torch.randn(b, 3, res, res, generator=g)
                ref.zero_grad(set_to_none=True)
                t0 = time.perf_counter()
                r = ref(x, ['single'])
                outs, gl = r if isinstance(r, tuple) and len(r) == 2 and not torch.is_tensor(r[0]) else (r, 0.0)
                (sum((o * o).mean() for o in outs) + gl).backward()
                return time.perf_counter() - t0
            run_ref(1, 512)
            out['step_seconds'] = [run_ref(1, RES) for _ in range(3)]
            out['kind'] = 'reference'
            print('CPUWORKER ' + json.dumps(out), flush=True)
            return
        from oracle import moe_oracle as MO
        net = build_model(cfg_name)
        bcfg = backbone_cfg(cfg_name)
        inds = bcfg.get('MoE_Block_inds', [[], [], [], []])
        E, topk = bcfg.get('num_experts', 2), bcfg.get('top_k', 2)
        p = {k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.endswith(('.mean', '.std')))
             for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(0)
        kw = dict(arch=bcfg['arch'], moe_block_inds=inds, num_experts=E, top_k=topk, train=True)

        def run(b, res):
            x = torch.randn(b, 3, res, res, generator=g)
            toks, H = [], res // 4
            for i, ii in enumerate(inds):
                if i > 0:
                    H //= 2
                toks += [b * H * H] * len([q for q in ii if q < net.depths[i]])
            noise = [torch.randn(t, E, generator=g) for t in toks]
            for v in p.values():
                v.grad = None
            t0 = time.perf_counter()
            res_ = MO.backbone_forward(x, p, noise=noise, **kw)
            outs, gl = res_ if toks else (res_, 0.0)
            (sum((o * o).mean() for o in outs) + gl).backward()
            return time.perf_counter() - t0
        run(1, 512)
        out['step_seconds'] = [run(1, RES) for _ in range(3)]
    else:
        import numpy as np
        from oracle import build_ref
        from tests import synth
        ref = build_ref.load_ref()
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731

        def timeit(fn, n=3):
            fn()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            return (time.perf_counter() - t0) / n * 1e6
        us = {}
        for m in (64, 512):
            b1, b2 = T(synth.rotated_boxes(2000, 0)), T(synth.rotated_boxes(m, 1))
            o = torch.zeros(2000 * m)
            us[f'box_iou_rotated_2000x{m}'] = timeit(lambda: ref.box_iou_rotated(b1, b2, o, 0, False))
        d, sc = T(synth.rotated_boxes(2000, 2, cluster=True)), T(synth.unique_scores(2000, 3))
        us['nms_rotated_2000'] = timeit(lambda: ref.nms_rotated_cpu(d, sc, 0.1))
        d, sc = T(synth.rotated_boxes(10000, 7)), T(synth.unique_scores(10000, 8))
        us['nms_rotated_10000'] = timeit(lambda: ref.nms_rotated_cpu(d, sc, 0.1), n=1)
        hb, hs = T(synth.hboxes(8768, 4, cluster=True)), T(synth.unique_scores(8768, 5))
        us['nms_8768'] = timeit(lambda: ref.nms(hb, hs, 0.8, 0))
        x = torch.randn(1, 256, 256, 256)
        rois = T(synth.rois_for_level(512, 6, batch=1, extent=1024.0))
        y = torch.zeros(512, 256, 7, 7)
        us['roi_align_rotated_fwd_512x256x7x7'] = timeit(
            lambda: ref.roi_align_rotated_forward(x, rois, y, 7, 7, 0.25, 2, True, True))
        gi = torch.zeros_like(x)
        us['roi_align_rotated_bwd_512x256x7x7'] = timeit(
            lambda: ref.roi_align_rotated_backward(y, rois, gi, 7, 7, 0.25, 2, True, True))
        xd, off = torch.randn(2, 256, 128, 128), torch.randn(2, 18, 128, 128) * 2
        w, od = torch.randn(256, 256, 3, 3) * 0.02, torch.zeros(2, 256, 128, 128)
        e = torch.zeros(0)
        us['deform_conv2d_fwd_2x256x128x128'] = timeit(
            lambda: ref.deform_conv_forward(xd, w, off, od, e, e, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 2), n=1)
        out['ops_us'] = {k: round(v, 1) for k, v in us.items()}
    print('CPUWORKER ' + json.dumps(out), flush=True)


def cpu_baseline(cfg_name=DEFAULT_CONFIG):
    """`cpu_baseline` of the bench line: the same training step on this host's cores, one subprocess per thread count
    (SURVEY.md 8(d) protocol), best value reported with its thread count; plus the reference CPU ops per shape."""
    import subprocess

    def worker(kind, threads):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), CUDA_VISIBLE_DEVICES='',
                   HIP_VISIBLE_DEVICES='')
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-worker', kind, '--config', cfg_name], env=env,
                               capture_output=True, text=True, timeout=420)
            for line in r.stdout.splitlines():
                if line.startswith('CPUWORKER '):
                    return json.loads(line[len('CPUWORKER '):])
            return {'error': (r.stderr or r.stdout)[-300:]}
        except Exception as e:  # noqa: BLE001
            return {'error': f'{type(e).__name__}: {e}'}

    ncpu = os.cpu_count() or 8
    per_threads, best = {}, None
    # the headline config sweeps the thread counts; the others (bigger models, same protocol) use the one that won there
    counts = CPU_THREADS if cfg_name == DEFAULT_CONFIG else (16,)
    for t in [t for t in counts if t <= ncpu] or [ncpu]:
        w = worker('step', t)
        if 'step_seconds' in w:
            sec = sorted(w['step_seconds'])[1]  # median of the 3 timed steps
            per_threads[str(t)] = round(1.0 / sec, 4)
            if best is None or 1.0 / sec > best[0]:
                best = (1.0 / sec, t, w['step_seconds'], w.get('kind', 'port'))
        else:
            per_threads[str(t)] = w.get('error', 'failed')
    ops = worker('ops', best[1] if best else 8)
    if best is None:
        return dict(value=None, unit='imgs/sec', cores=0, kind='port', sample='cpu worker failed', detail=per_threads)
    kind = best[3]
    what = ('the reference backbone module itself (mmrotate/models/backbones/convnext_moe.py, unmodified; loaded by '
            'oracle/ref_moe.py from the bytecode oracle/build_ref.py compiled out of /root/reference, with stand-ins only for '
            'the timm / mmengine / mmcv container classes)' if kind == 'reference' else
            'oracle/moe_oracle.py (restatement of the reference backbone, pinned to it at this size by tests/test_oracle_fullsize.py)')
    return dict(value=round(best[0], 4), unit='imgs/sec', cores=best[1], kind=kind,
                sample=f'{what}: training step fwd+bwd fp32 on 1x3x{RES}x{RES}, 1 warm-up + 3 timed '
                       f'steps ({", ".join(f"{s:.1f}" for s in best[2])} s) per thread count, median, no scaling; best of '
                       f'OMP_NUM_THREADS in {list(per_threads)} on a {ncpu}-thread host',
                imgs_per_sec_by_threads=per_threads,
                ops_us_reference_cpu=ops.get('ops_us', ops.get('error')),
                ops_kind='reference (oracle/_ref = the reference CPU operators compiled from its own sources; '
                         'single-threaded by construction except DeformConv2d, whose GEMM uses the step\'s thread count)')


def ops_microbench():
    """per-op timings (microseconds) of hot path (b) at the SURVEY.md 8(d) shapes."""
    import numpy as np
    from sm3det_amd import mmcv_ops as ops
    from tests import synth
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731

    def timeit(fn, n=10):
        # median of three back-to-back batches of n calls (a batch that hits an allocator refill -- hipMalloc / hipFree of
        # a cached segment -- once doubled a number: 13.2 vs 7.5 ms for the same RPN tower)
        fn()
        fn()
        torch.cuda.synchronize()
        batches = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            batches.append((time.perf_counter() - t0) / n * 1e6)
        return sorted(batches)[1]

    # SM3_BENCH_OPS=slice: only the detector slice runs (profiling aid: rocprofv3 then sees the slice's kernels alone)
    timeit_real = timeit
    if os.environ.get('SM3_BENCH_OPS') == 'slice':
        timeit = lambda fn, n=10: 0.0  # noqa: E731

    out = {}
    b1, b2 = dev(synth.rotated_boxes(2000, 0)), dev(synth.rotated_boxes(512, 1))
    out['box_iou_rotated_2000x512'] = timeit(lambda: ops.box_iou_rotated(b1, b2))
    b64 = dev(synth.rotated_boxes(64, 1))
    out['box_iou_rotated_2000x64'] = timeit(lambda: ops.box_iou_rotated(b1, b64))
    d, s = dev(synth.rotated_boxes(2000, 2, cluster=True)), dev(synth.unique_scores(2000, 3))
    out['nms_rotated_2000'] = timeit(lambda: ops.nms_rotated(d, s, 0.1))
    hb, hs = dev(synth.hboxes(8768, 4, cluster=True)), dev(synth.unique_scores(8768, 5))
    out['nms_8768'] = timeit(lambda: ops.nms(hb, hs, iou_threshold=0.8))
    x = torch.randn(1, 256, 256, 256, device='cuda', requires_grad=True)
    rois = dev(synth.rois_for_level(512, 6, batch=1, extent=1024.0))
    layer = ops.RoIAlignRotated(output_size=7, spatial_scale=0.25, sampling_ratio=2, clockwise=True)
    out['roi_align_rotated_fwd_512x256x7x7'] = timeit(lambda: layer(x, rois))
    y = layer(x, rois)
    go = torch.randn_like(y)
    out['roi_align_rotated_bwd_512x256x7x7'] = timeit(lambda: torch.autograd.grad(y, x, go, retain_graph=True))
    xl = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)  # NHWC features (MI355X layout)
    out['roi_align_rotated_fwd_nhwc'] = timeit(lambda: layer(xl, rois))
    yl = layer(xl, rois)
    out['roi_align_rotated_bwd_nhwc'] = timeit(lambda: torch.autograd.grad(yl, xl, go, retain_graph=True))
    d10, s10 = dev(synth.rotated_boxes(10000, 7)), dev(synth.unique_scores(10000, 8))
    out['nms_rotated_10000'] = timeit(lambda: ops.nms_rotated(d10, s10, 0.1), n=3)
    # the neck of main_SM3Det.py on backbone-shaped NHWC inputs (bs 2 @ 1024^2, start_level 0)
    from sm3det_amd.fpn import MultitaskFPN
    fpn = MultitaskFPN(in_channels=[96, 192, 384, 768], out_channels=256, extra_level=1, add_extra_convs='on_output',
                       num_outs=5).cuda()
    feats = [torch.randn(BATCH, c, 256 >> i, 256 >> i, device='cuda').contiguous(memory_format=torch.channels_last)
             .requires_grad_(True) for i, c in enumerate([96, 192, 384, 768])]

    def fpn_step():
        for q in fpn.parameters():
            q.grad = None
        sum((o * o).mean() for o in fpn(feats)).backward()
    out['multitask_fpn_fwd_bwd_bs2_1024'] = timeit(fpn_step, n=5)
    # Oriented-RPN head on the 5 neck outputs (tower fwd+bwd) and the proposal glue of one image (nms_pre 2000)
    from sm3det_amd.rpn_head import OrientedRPNHead, grid_anchors
    rpn = OrientedRPNHead(in_channels=256, feat_channels=256, version='le90',
                          bbox_coder=dict(type='MidpointOffsetCoder', angle_range='le90', target_means=[0.0] * 6,
                                          target_stds=[1.0, 1.0, 1.0, 1.0, 0.5, 0.5]),
                          test_cfg=dict(nms_pre=2000, max_per_img=2000, nms=dict(type='nms', iou_threshold=0.8),
                                        min_bbox_size=0)).cuda()
    rpn.init_weights()
    lv = [torch.randn(BATCH, 256, 256 >> i, 256 >> i, device='cuda').contiguous(memory_format=torch.channels_last)
          .requires_grad_(True) for i in range(5)]

    def rpn_step():
        for q in rpn.parameters():
            q.grad = None
        cls, reg = rpn(lv)
        (sum((c * c).mean() for c in cls) + sum((r * r).mean() for r in reg)).backward()
    out['orpn_tower_fwd_bwd_bs2_5levels'] = timeit(rpn_step, n=5)
    with torch.no_grad():
        cls, reg = rpn(lv)
        anchors = grid_anchors([tuple(c.shape[-2:]) for c in cls], [4, 8, 16, 32, 64])
        c0, r0 = [c[0] for c in cls], [r[0] for r in reg]
        out['orpn_proposals_one_image'] = timeit(lambda: rpn._get_bboxes_single(c0, r0, anchors, (1024, 1024, 3)), n=5)
    # RoI path: fused multi-level extractor vs the reference's per-level loop over the single-level op, 1024 RoIs
    from sm3det_amd.roi_head import RotatedShared2FCBBoxHead, RotatedSingleRoIExtractor
    ext = RotatedSingleRoIExtractor(dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True), 256,
                                    [4, 8, 16, 32])
    gr = torch.Generator().manual_seed(12)
    rr = torch.zeros(1024, 6)
    rr[:, 0] = torch.randint(0, BATCH, (1024,), generator=gr).float()
    rr[:, 1:3] = torch.rand(1024, 2, generator=gr) * 1024
    rr[:, 3] = torch.exp(torch.rand(1024, generator=gr) * 4.2 + 2.0)
    rr[:, 4] = rr[:, 3] * (0.3 + torch.rand(1024, generator=gr))
    rr[:, 5] = (torch.rand(1024, generator=gr) - 0.5) * 3.1
    rr = rr.cuda()
    f4 = lv[:4]
    out['roi_extract_fused_1024rois_fwd'] = timeit(lambda: ext(f4, rr))
    ro = ext(f4, rr)
    gro = torch.randn_like(ro)
    out['roi_extract_fused_1024rois_bwd'] = timeit(lambda: torch.autograd.grad(ro, f4, gro, retain_graph=True), n=5)
    layers = [ops.RoIAlignRotated(output_size=7, spatial_scale=1.0 / s, sampling_ratio=2, clockwise=True)
              for s in (4, 8, 16, 32)]

    def per_level_loop():  # rotate_single_level_roi_extractor.py:128-140
        tl = ext.map_roi_levels(rr, 4)
        res = torch.zeros(1024, 256, 7, 7, device='cuda')
        for i in range(4):
            inds = (tl == i).nonzero(as_tuple=False).squeeze(1)
            if inds.numel() > 0:
                res[inds] = layers[i](f4[i].detach(), rr[inds])
        return res
    out['roi_extract_per_level_loop_1024rois_fwd'] = timeit(per_level_loop)
    head = RotatedShared2FCBBoxHead(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=26,
                                    reg_class_agnostic=True).cuda()
    xf = ro.detach().requires_grad_(True)

    def head_step():
        for q in head.parameters():
            q.grad = None
        a, b = head(xf)
        ((a * a).mean() + (b * b).mean()).backward()
    out['shared2fc_head_fwd_bwd_1024rois'] = timeit(head_step, n=5)
    # DeformConv2d at the SURVEY 8(d) shape: x (2,256,128,128), 3x3, 256 -> 256, offsets randn*2
    from sm3det_amd.mmcv_deform_conv import deform_conv2d
    xd = torch.randn(2, 256, 128, 128, device='cuda', requires_grad=True)
    od = (torch.randn(2, 18, 128, 128, device='cuda') * 2).requires_grad_(True)
    wd = (torch.randn(256, 256, 3, 3, device='cuda') * 0.02).requires_grad_(True)
    out['deform_conv2d_fwd_2x256x128x128'] = timeit(lambda: deform_conv2d(xd, od, wd, 1, 1, 1, 1, 1, False, 2), n=5)
    yd = deform_conv2d(xd, od, wd, 1, 1, 1, 1, 1, False, 2)
    gd = torch.randn_like(yd)
    out['deform_conv2d_bwd_2x256x128x128'] = timeit(
        lambda: torch.autograd.grad(yd, (xd, od, wd), gd, retain_graph=True), n=5)
    del xd, od, wd, yd, gd
    torch.cuda.synchronize()
    torch.cuda.empty_cache()  # the column buffers of DeformConv2d (GBs) otherwise distort the allocator for what follows
    # GFL head towers (SAR branch) on the 5 levels it sees at 1024^2 (strides 8..128), one image, forward + backward
    from sm3det_amd.gfl_head import GFLHead
    gfl = GFLHead(num_classes=26, in_channels=256, stacked_convs=4, feat_channels=256, reg_max=16).cuda()
    gl_feats = [torch.randn(1, 256, 128 >> i, 128 >> i, device='cuda').contiguous(memory_format=torch.channels_last)
                .requires_grad_(True) for i in range(5)]

    def gfl_step():
        for q in gfl.parameters():
            q.grad = None
        cs, rg = gfl(gl_feats)
        (sum((c * c).mean() for c in cs) + sum((r * r).mean() for r in rg)).backward()
    out['gfl_head_fwd_bwd_bs1_5levels'] = timeit(gfl_step, n=5)
    # target assignment on the real shapes: rpn = 261 888 anchors x 8 gts (horizontal), rcnn = 2000 proposals x 8 gts
    from sm3det_amd.assign import MaxIoUAssigner, RandomSampler
    from sm3det_amd.rpn_head import grid_anchors as _ga
    anc = torch.cat(_ga([(256 >> i, 256 >> i) for i in range(5)], [4, 8, 16, 32, 64], [8], [0.5, 1.0, 2.0], device='cuda'))
    ghb = dev(synth.hboxes(8, 31, extent=1024.0))
    rpn_asg = MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True)
    out['max_iou_assign_rpn_261888x8'] = timeit(lambda: rpn_asg.assign(anc, ghb))
    grb = dev(synth.rotated_boxes(8, 32))
    prb = dev(synth.rotated_boxes(2000, 33, cluster=True))
    rcnn_asg = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False,
                              iou_calculator=dict(type='RBboxOverlaps2D'))
    out['max_iou_assign_rcnn_2000x8_rotated'] = timeit(lambda: rcnn_asg.assign(prb, grb))
    # ---- second named workload (never part of `value`): the two-stage (RGB / IR) branch as the detector chains it ----
    # TriSourceDetector.forward_train for a batch of RGB images (trisource_H1stage_R2stage_detector.py:235-369): backbone ->
    # MultitaskFPN -> OrientedRPNHead.forward_train (tower, masked MaxIoU assignment of 261 888 anchors / image, sampler,
    # fused target-encode + BCE / SmoothL1 losses, fixed-size proposals: top-2000 per level, decode, NMS 0.8) ->
    # OrientedStandardRoIHead.forward_train (rotated MaxIoU assignment, add_gt_as_proposals, 512 RoIs / image, fused
    # multi-level RoIAlignRotated, Shared2FC head, fused target-encode + CE / SmoothL1 losses) -> backward -> grad-clip +
    # AdamW over every parameter.  All pieces are built from the reference config's own `model` dict (committed copy) with
    # the modality's train / test cfg, the losses are the REAL ones; sync-free and replayed from two hipGraphs.  Not in it:
    # the SAR branch (GFL head: its towers are `gfl_head_fwd_bwd_bs1_5levels` above; ATSS / QFL / DFL / GIoU are mmdet code
    # that is not built) and the data pipeline.
    from sm3det_amd.config import build_detector_pieces
    from sm3det_amd.optim import MultiTensorAdamW
    torch.manual_seed(1)
    pcs = build_detector_pieces(load_config(DEFAULT_CONFIG)['model'])
    fpn2, rpn2, roi2 = pcs['neck'].cuda(), pcs['rgb_rpn_head'].cuda(), pcs['rgb_roi_head'].cuda()
    rpn2.init_weights()
    roi2.init_weights()
    bb = build_model().cuda().train()
    img = torch.randn(BATCH, 3, RES, RES, device='cuda')
    mods = (bb, fpn2, rpn2, roi2)
    gts = [dev(synth.rotated_boxes(8, 40 + i)) for i in range(BATCH)]
    gls = [torch.randint(0, 26, (8,), generator=torch.Generator().manual_seed(50 + i)).cuda() for i in range(BATCH)]
    metas = [dict(img_shape=(RES, RES, 3), pad_shape=(RES, RES, 3)) for _ in range(BATCH)]
    prop_cfg = load_config(DEFAULT_CONFIG)['model']['rgb_train_cfg']['rpn_proposal']
    sl_params = [q for m in mods for q in m.parameters() if q.requires_grad]
    sl_opt = MultiTensorAdamW([dict(params=[q]) for q in sl_params], lr=1e-4, betas=(0.9, 0.999), weight_decay=0.05,
                              max_grad_norm=35.0)
    sl_losses = {}

    def slice_fwd_bwd():
        for q in sl_params:
            q.grad = None
        feats_, gl_ = bb(img, ['single'])
        pyr = fpn2(feats_)
        rpn_losses, props = rpn2.forward_train(pyr, metas, gts, proposal_cfg=prop_cfg)
        roi_losses = roi2.forward_train(pyr, metas, props, gts, gls)
        terms = dict(gate_loss=gl_, loss_rpn_cls=sum(rpn_losses['loss_rpn_cls']), loss_rpn_bbox=sum(rpn_losses['loss_rpn_bbox']),
                     loss_cls=roi_losses['loss_cls'], loss_bbox=roi_losses['loss_bbox'])
        loss = sum(terms.values())  # BaseDetector._parse_losses: every key containing 'loss'
        loss.backward()
        sl_losses.update({k: v.detach() for k, v in terms.items()}, acc=roi_losses['acc'].detach())
        return loss

    def slice_step():
        slice_fwd_bwd()
        sl_opt.step()

    # everything about the slice runs on a NON-default stream: autograd caches each parameter's AccumulateGrad node with
    # the stream of its first use, and a node bound to the legacy default stream drags that stream into a later capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out['detector_slice_eager_train_step_bs2_1024'] = timeit_real(slice_step, n=3)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    first = {k: float(v) for k, v in sl_losses.items()}
    try:
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            slice_fwd_bwd()
        keep = [q.grad for q in sl_params]  # noqa: F841  (the tensors the graph writes the gradients to)
        g1.replay()
        sl_opt.refresh_grad_pointers()
        with torch.cuda.graph(g2, pool=g1.pool()):
            sl_opt.step()

        def replay():
            g1.replay()
            g2.replay()
        out['detector_slice_train_step_bs2_1024'] = timeit_real(replay, n=5)
    except Exception as e:  # noqa: BLE001
        print(f'[bench] detector slice: hipGraph capture failed ({type(e).__name__}: {e})', file=sys.stderr)
        out['detector_slice_train_step_bs2_1024'] = out['detector_slice_eager_train_step_bs2_1024']
    torch.cuda.synchronize()
    extra = dict(losses_first_step={k: round(v, 5) for k, v in first.items()},
                 losses_last_step={k: round(float(v), 5) for k, v in sl_losses.items()},
                 params_m=round(sum(q.numel() for q in sl_params) / 1e6, 2))
    res = {k: round(v, 1) for k, v in out.items()}
    res['detector_slice'] = extra
    return res



def full_model_bench():
    """Third named workload (never part of `value`): ONE TRAINING STEP OF THE WHOLE DETECTOR of local_configs/main_SM3Det.py
    at the config's native modality mix -- 2 SAR + 1 RGB + 1 IR images of 1024^2 (source_ratio [2, 1, 1]) --
    `MODELS.build(cfg.model)` -> `TriSourceDetector.forward_train` (one MoE backbone call on the 4 concatenated images,
    MultitaskFPN x3, GFLHead with ATSS + QFL / DFL / GIoU for the SAR pair, OrientedRPNHead + OrientedStandardRoIHead with
    their real targets and losses for the RGB and the IR image) -> BaseDetector._parse_losses -> backward -> grad-clip(35)
    + AdamW over all 178 M parameters.  Inputs are device-resident; the step is sync-free and replayed from two hipGraphs
    when capture succeeds.  The SAR loss side (ATSS, QFL / DFL / GIoU) is restated from mmdet (not vendored: parity unpinned)
    and runs on csrc/gfl.hip like everything else."""
    import copy
    import numpy as np
    from sm3det_amd import detector  # noqa: F401
    from sm3det_amd.optim import MultiTensorAdamW
    from sm3det_amd.registry import MODELS
    from tests import synth
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    m = copy.deepcopy(load_config(DEFAULT_CONFIG)['model'])
    m['backbone'].pop('init_cfg', None)
    torch.manual_seed(2)
    det = MODELS.build(m).cuda().train()
    for h in (det.rgb_rpn_head, det.rgb_roi_head, det.ifr_rpn_head, det.ifr_roi_head):
        h.init_weights()
    with torch.no_grad():
        for n, p in det.backbone.named_parameters():
            if n.endswith('gamma'):
                p.fill_(1.0)
    mix = dict(sar=2, rgb=1, ifr=1)
    g = torch.Generator().manual_seed(3)
    img = {s: torch.randn(n, 3, RES, RES, generator=g).cuda() for s, n in mix.items()}
    metas = {s: [dict(img_shape=(RES, RES, 3), pad_shape=(RES, RES, 3)) for _ in range(n)] for s, n in mix.items()}
    gtb = {s: [dev(synth.hboxes(8, 80 + i, extent=float(RES))) if s == 'sar' else dev(synth.rotated_boxes(8, 90 + i + 5 * (s == 'ifr')))
               for i in range(n)] for s, n in mix.items()}
    gtl = {s: [torch.randint(0, 26, (8,), generator=g).cuda() for _ in range(n)] for s, n in mix.items()}
    params = [q for q in det.parameters() if q.requires_grad]
    opt = MultiTensorAdamW([dict(params=[q]) for q in params], lr=1e-4, betas=(0.9, 0.999), weight_decay=0.05, max_grad_norm=35.0)
    logs = {}
    # lr_config = dict(policy='dynamic', ...) of the config (main_SM3Det.py:291-300): DynamicLrUpdaterHook recomputes one lr
    # per parameter tensor from the 11 loss scalars every iteration, BEFORE that iteration's optimizer step (hook priority).
    # Here one kernel on the device-resident log_vars (sm3det_amd.optim.DeviceDynamicLr), inside the captured step.
    from sm3det_amd.optim import DeviceDynamicLr
    lrc = dict(load_config(DEFAULT_CONFIG).get('lr_config') or {})
    dla = None
    if lrc.pop('policy', None) == 'dynamic':
        names = [n for n, q in det.named_parameters() if q.requires_grad]
        for q in params:  # the tables need the gradient set the optimizer will see
            q.grad = torch.zeros_like(q)
        dla = DeviceDynamicLr(opt, names, **lrc)
        for q in params:
            q.grad = None

    def fwd_bwd():
        for q in params:
            q.grad = None
        losses = det.forward_train_gathered(img, metas, gtb, gtl)
        total, lv = det.parse_losses(losses)
        if dla is not None:
            dla.update(lv)  # after_train_iter of the lr hook: runs before the optimizer hook's backward + step
        total.backward()
        logs.update({k: v.detach() for k, v in lv.items()})

    def step():
        fwd_bwd()
        opt.step()

    def timeit(fn, n):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / n * 1e3)
        return sorted(ts)[1]

    out = dict(images_per_step=sum(mix.values()), mix=mix, params_m=round(sum(q.numel() for q in params) / 1e6, 2))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
        torch.cuda.synchronize()
        first = {k: round(float(v), 5) for k, v in logs.items()}
        if dla is not None:  # time the steady state of the policy (softmax weights + sigmoid_kl), not the linear warm-up
            dla.fast_forward(int(lrc.get('warmup_iters') or 0))
        out['ms_per_step_eager'] = round(timeit(step, 3), 3)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ms = out['ms_per_step_eager']
    try:
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            fwd_bwd()
        keep = [q.grad for q in params]  # noqa: F841
        g1.replay()
        opt.refresh_grad_pointers()
        with torch.cuda.graph(g2, pool=g1.pool()):
            opt.step()

        def replay():
            g1.replay()
            g2.replay()
        ms = out['ms_per_step_graph'] = round(timeit(replay, 5), 3)
        out['hip_graph'] = True
    except Exception as e:  # noqa: BLE001
        print(f'[bench] full model: hipGraph capture failed ({type(e).__name__}: {e})', file=sys.stderr)
        out['hip_graph'] = False
    torch.cuda.synchronize()
    # roofline of the full step ("ceilings for backbone-only and full-model separately"): one more eager step with
    # HIP events around every C-ABI launch; the contraction family = backbone / head GEMMs + the implicit-GEMM 3x3 convolutions
    try:
        from sm3det_amd import _lib_backbone as LB
        best = None
        for _ in range(2):
            LB.PROFILE = []
            step()
            torch.cuda.synchronize()
            rec, LB.PROFILE = LB.PROFILE, None
            tot = sum(e0.elapsed_time(e1) for _n, _f, _b, e0, e1 in rec)
            if best is None or tot < best[0]:
                best = (tot, rec)
        fam = [(n, f, e0.elapsed_time(e1)) for n, f, _b, e0, e1 in best[1] if n.startswith('gemm_f') or n.startswith('conv3x3_')]
        g_ms, g_fl = sum(t for _, _, t in fam), sum(f for _, f, _ in fam)
        other = {}
        for n, _f, _b, e0, e1 in best[1]:
            if not (n.startswith('gemm_f') or n.startswith('conv3x3_')):
                other[n.split(' ')[0]] = other.get(n.split(' ')[0], 0.0) + e0.elapsed_time(e1)
        b3 = LB.ARITH32 == 2
        peak = MI355X_BF16X3_PEAK_TFLOPS if b3 else MI355X_FP32_MFMA_PEAK_TFLOPS
        ach = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        out['roofline'] = dict(
            bound='mfma', kernel='gemm_f32_kernel family (backbone / head GEMMs + implicit-GEMM 3x3 convolutions), ' +
            ('bf16x3 form' if b3 else 'native fp32 MFMA'), achieved=round(ach, 2), peak=round(peak, 1), unit='TFLOP/s',
            frac=round(ach / peak, 4), algorithmic_gflop_per_step=round(g_fl / 1e9, 1), launches_per_step=len(fam),
            gemm_ms_per_step=round(g_ms, 3), conv3x3_ms_per_step=round(sum(t for n, _, t in fam if n.startswith('conv3x3_')), 3),
            other_library_kernels_ms_per_step=round(sum(other.values()), 3),
            torch_glue_and_gaps_ms_per_step=round(max(ms - g_ms - sum(other.values()), 0.0), 3),
            top_other_ms={k: round(v, 3) for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:10]},
            note='per-launch HIP-event brackets of one eager step (they add launch gaps); ms_per_step_graph is the replayed step')
    except Exception as e:  # noqa: BLE001
        out['roofline'] = dict(error=f'{type(e).__name__}: {e}'[:200])
    out['losses_first_step'] = first
    out['losses_last_step'] = {k: round(float(v), 5) for k, v in logs.items()}
    out['imgs_per_sec'] = round(sum(mix.values()) / (ms * 1e-3), 2)
    if dla is not None:
        emas, updates, iters = dla.state_host()
        lrv = opt._lr.cpu()
        out['dynamic_lr'] = dict(policy='dynamic (DynamicLrUpdaterHook on the device: sm3_dla_lr inside the captured step)',
                                 extra_args=lrc.get('extra_args'), iterations=iters, lr_min=float(lrv.min()),
                                 lr_max=float(lrv.max()), loss_emas=[round(e, 5) for e in emas])
    del det, opt, params
    torch.cuda.empty_cache()
    return out


def full_model_dp_main(args, world, rank, multi, backend, force_dist):
    """`--workload full_model`: the data-parallel training step of the WHOLE detector (BASELINE configs #3 / #5 are detector
    DDP runs): every rank holds the 178 M-parameter TriSourceDetector and its own synthetic native mix (2 SAR + 1 RGB + 1 IR
    @1024^2); graph 1 = forward + all losses + backward, gradients landing in / packed into 64 MiB buckets; then ONE
    all-reduce of the stacked log_vars (mmdet's `_parse_losses` issues one blocking all-reduce + .item() per loss: ~15) and
    the bucket all-reduces (RCCL, ReduceOp.AVG); graph 2 = the dynamic-lr policy on the REDUCED losses (as the reference's hook
    sees them) + grad-clip + AdamW.  Same timing contract as the headline; prints the one JSON line."""
    import copy
    import numpy as np
    import torch.distributed as dist
    from sm3det_amd import detector  # noqa: F401
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd.data_parallel import BucketedGradReducer
    from sm3det_amd.optim import DeviceDynamicLr, MultiTensorAdamW
    from sm3det_amd.registry import MODELS
    from tests import synth
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    cfg_entry = load_config(DEFAULT_CONFIG)
    m = copy.deepcopy(cfg_entry['model'])
    m['backbone'].pop('init_cfg', None)
    torch.manual_seed(2)
    det = MODELS.build(m).cuda().train()
    for h in (det.rgb_rpn_head, det.rgb_roi_head, det.ifr_rpn_head, det.ifr_roi_head):
        h.init_weights()
    with torch.no_grad():
        for n, q in det.backbone.named_parameters():
            if n.endswith('gamma'):
                q.fill_(1.0)
    mix = dict(sar=2, rgb=1, ifr=1)
    g = torch.Generator().manual_seed(3 + rank)  # every rank its own shard
    img = {s_: torch.randn(n, 3, RES, RES, generator=g).cuda() for s_, n in mix.items()}
    metas = {s_: [dict(img_shape=(RES, RES, 3), pad_shape=(RES, RES, 3)) for _ in range(n)] for s_, n in mix.items()}
    gtb = {s_: [dev(synth.hboxes(8, 80 + i + 100 * rank, extent=float(RES))) if s_ == 'sar' else
                dev(synth.rotated_boxes(8, 90 + i + 5 * (s_ == 'ifr') + 100 * rank)) for i in range(n)] for s_, n in mix.items()}
    gtl = {s_: [torch.randint(0, 26, (8,), generator=g).cuda() for _ in range(n)] for s_, n in mix.items()}
    named = [(n, q) for n, q in det.named_parameters() if q.requires_grad]
    params = [q for _, q in named]
    reducer = BucketedGradReducer(params, bucket_mb=64.0, force_comm=force_dist)
    reducer.broadcast_parameters(0, module=det)
    reducer.overlap = False
    opt = MultiTensorAdamW([dict(params=[q]) for q in params], lr=1e-4, betas=(0.9, 0.999), weight_decay=0.05, max_grad_norm=35.0)
    lrc = dict(cfg_entry.get('lr_config') or {})
    lrc.pop('policy', None)
    for q in params:
        q.grad = torch.zeros_like(q)
    dla = DeviceDynamicLr(opt, [n for n, _ in named], **lrc)
    state = {}

    def fwd_bwd():
        reducer.zero_grad()
        losses = det.forward_train_gathered(img, metas, gtb, gtl)
        total, lv = det.parse_losses(losses)
        total.backward()
        reducer.pack_all()
        state['keys'] = list(lv)
        state['lv'] = torch.stack([v.detach().float().reshape(()) for v in lv.values()])  # one vector: one collective

    def reduce_and_step():
        lv = state['lv']
        if multi and world > 1:
            dist.all_reduce(lv, op=dist.ReduceOp.SUM)
            lv = lv / world
        state['lv_mean'] = lv
        reducer.finalize(repack=False)
        dla.update({k: lv[i] for i, k in enumerate(state['keys'])})
        opt.step()

    def step():
        fwd_bwd()
        reduce_and_step()

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
        torch.cuda.synchronize()
        dla.fast_forward(int(lrc.get('warmup_iters') or 0))
        step()
    torch.cuda.current_stream().wait_stream(side)
    fence()
    run, use_graph = step, False
    if not args.no_graph:
        cap_kw = dict(capture_error_mode='thread_local') if multi else {}
        try:
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, **cap_kw):
                fwd_bwd()
            keep = [q.grad for q in params]  # noqa: F841
            g1.replay()
            reducer.finalize(repack=False)
            opt.refresh_grad_pointers()
            lv_buf = state['lv']
            with torch.cuda.graph(g2, pool=g1.pool(), **cap_kw):
                dla.update({k: lv_buf[i] for i, k in enumerate(state['keys'])})
                opt.step()

            def run():
                g1.replay()
                if multi and world > 1:  # the stacked log_vars: summed in place, the policy divides by reading the mean
                    dist.all_reduce(lv_buf, op=dist.ReduceOp.SUM)
                    lv_buf.div_(world)
                reducer.finalize(repack=False)
                g2.replay()
            use_graph = True
        except Exception as e:  # noqa: BLE001
            print(f'[bench] full model DP: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr)
            torch.cuda.synchronize()
            run = step
    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    spread = 0.0
    if multi:
        chk = torch.stack([q.detach().double().sum() for q in params]).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = float((hi - lo).item())
    n_img = sum(mix.values())
    if rank == 0:
        lv = state['lv'].detach().cpu().tolist()
        result = {
            'metric': f'train imgs/sec SM3Det ConvNeXt-T e8t2 @{RES}^2, whole detector (TriSourceDetector of main_SM3Det.py), '
                      f'{n_img} images/GPU native mix',
            'value': round(world * n_img * args.steps / dt, 3), 'unit': 'imgs/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'full_model: ' + full_model_dp_main.__doc__.split('\n')[0].strip(),
                       'name': DEFAULT_CONFIG, 'global_batch': world * n_img, 'resolution': RES, 'parallelism': f'dp{world}',
                       'params_m': round(sum(q.numel() for q in params) / 1e6, 2), 'grad_buckets': reducer.num_buckets,
                       'hip_graph': use_graph, 'dist_backend': (backend if multi else None),
                       'collective_avg': bool(reducer._avg), 'log_var_collectives_per_step': 1 if (multi and world > 1) else 0,
                       'grad_bytes_in_place_frac': round(reducer.pack_stats['in_place_bytes'] /
                                                         max(reducer.pack_stats['in_place_bytes'] + reducer.pack_stats['copied_bytes'], 1), 4)
                       if multi else None,
                       'replica_checksum_spread': spread, 'gemm_arith': LB.gemm_arith(),
                       'dynamic_lr': 'DeviceDynamicLr on the rank-averaged losses (sm3_dla_lr)'},
            'loss_terms': dict(zip(state['keys'], [round(v, 5) for v in lv])),
            'roofline': None, 'cpu_baseline': None,
        }
        print(json.dumps(result), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command line under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1 -- tools/dist_train.sh:8-19 of the reference does the same with
    torch.distributed.launch) and relay what they print; rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sock:  # a free port for this job (several benches may share a host)
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '8'))
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


MI355X_VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz: fp32 lane-instructions / s (= 157.3 TF / 2)


def ops_roofline(us):
    """achieved-vs-roofline of the hot path (b) operators from their `ops_us` timings.  Algorithmic work per call is
    SURVEY.md 8(d)'s: RoIAlignRotated = HBM gather / scatter (output + each touched feature map once, backward zero-fill +
    accumulate), box_iou_rotated / NMS = VALU (pair tests; the instruction count per pair is measured with rocprofv3
    SQ_INSTS_VALU and committed under profiles/r04/ops_pmc.json), DeformConv2d forward = MFMA GEMM + HBM-bound sampling."""
    pmc = {}
    for rnd in ('r04', 'r03'):  # the newest committed per-operator counter passes
        try:
            with open(os.path.join(ROOT, 'profiles', rnd, 'ops_pmc.json')) as f:
                pmc = json.load(f)
            break
        except Exception:
            continue
    out = {}

    def hbm(name, key, nbytes):
        if us.get(key):  # (absent or 0 in the SM3_BENCH_OPS=slice / =full profiling modes)
            gbs = nbytes / (us[key] * 1e-6) / 1e9
            out[name] = dict(bound='hbm', algorithmic_mb=round(nbytes / 1e6, 1), us=us[key], achieved=round(gbs, 1),
                             peak=MI355X_HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / MI355X_HBM_PEAK_GBS, 4),
                             traffic=pmc.get(name, {}).get('hbm_bytes'))

    def valu(name, key, pairs, what):
        if us.get(key):
            rate = pairs / (us[key] * 1e-6)
            ipp = pmc.get(name, {}).get('valu_lane_instr_per_pair')
            peak = MI355X_VALU_LANE_OPS / ipp if ipp else None
            out[name] = dict(bound='valu', work=what, pairs=int(pairs), us=us[key], achieved=round(rate / 1e9, 3),
                             unit='G pairs/s', valu_lane_instr_per_pair=ipp,
                             peak=round(peak / 1e9, 3) if peak else None, frac=round(rate / peak, 4) if peak else None,
                             serial_floor_us=pmc.get(name, {}).get('serial_floor_us'))
            # the pair-test kernel alone (rocprofv3 duration from the committed per-operator pass): `us` above is the
            # whole operator -- argsort, mask kernel, serial sweep, 5-9 launches
            kern = pmc.get(name, {}).get('kernels', {})
            pk = next((kern[k]['us_per_call'] for k in ('box_iou_rotated_kernel', 'nms_rotated_mask_kernel', 'nms_mask_kernel')
                       if k in kern), None)
            if pk and peak:
                out[name].update(pair_kernel_us=pk, pair_kernel_frac=round(pairs / (pk * 1e-6) / peak, 4))
    n, C, HW = 512, 256, 256 * 256
    fmap, roi_out = 1 * C * HW * 4, n * C * 49 * 4
    hbm('roi_align_rotated_fwd_nchw', 'roi_align_rotated_fwd_512x256x7x7', roi_out + fmap + n * 24)
    hbm('roi_align_rotated_fwd_nhwc', 'roi_align_rotated_fwd_nhwc', roi_out + fmap + n * 24)
    hbm('roi_align_rotated_bwd_nchw', 'roi_align_rotated_bwd_512x256x7x7', roi_out + 2 * fmap)
    hbm('roi_align_rotated_bwd_nhwc', 'roi_align_rotated_bwd_nhwc', roi_out + 2 * fmap)
    valu('box_iou_rotated_2000x512', 'box_iou_rotated_2000x512', 2000 * 512, 'N*M rotated pair IoUs')
    valu('box_iou_rotated_2000x64', 'box_iou_rotated_2000x64', 2000 * 64, 'N*M rotated pair IoUs')
    valu('nms_rotated_10000', 'nms_rotated_10000', 10000 * 9999 / 2, 'N(N-1)/2 rotated pair tests (mask) + serial sweep')
    valu('nms_rotated_2000', 'nms_rotated_2000', 2000 * 1999 / 2, 'N(N-1)/2 rotated pair tests (mask) + serial sweep')
    valu('nms_8768', 'nms_8768', 8768 * 8767 / 2, 'N(N-1)/2 horizontal pair tests (mask) + serial sweep')
    if us.get('deform_conv2d_fwd_2x256x128x128'):
        fl = 2.0 * 2 * 128 * 128 * 256 * 256 * 9
        t = us['deform_conv2d_fwd_2x256x128x128'] * 1e-6
        out['deform_conv2d_fwd'] = dict(bound='mfma', algorithmic_gflop=round(fl / 1e9, 1), us=round(t * 1e6, 1),
                                        achieved=round(fl / t / 1e12, 2), peak=MI355X_FP32_MFMA_PEAK_TFLOPS,
                                        unit='TFLOP/s', frac=round(fl / t / 1e12 / MI355X_FP32_MFMA_PEAK_TFLOPS, 4),
                                        algorithmic_mb=round(2 * 128 * 128 * (256 + 256 + 18) * 4 / 1e6, 1))
    if us.get('deform_conv2d_bwd_2x256x128x128'):
        fl = 2.0 * 2.0 * 2 * 128 * 128 * 256 * 256 * 9  # input-gradient GEMM + weight-gradient GEMM
        t = us['deform_conv2d_bwd_2x256x128x128'] * 1e-6
        out['deform_conv2d_bwd'] = dict(bound='mfma', algorithmic_gflop=round(fl / 1e9, 1), us=round(t * 1e6, 1),
                                        achieved=round(fl / t / 1e12, 2), peak=MI355X_FP32_MFMA_PEAK_TFLOPS,
                                        unit='TFLOP/s', frac=round(fl / t / 1e12 / MI355X_FP32_MFMA_PEAK_TFLOPS, 4))
    return out


def native_f32_mfma_line(args):
    """The headline step once more in a child process with SM3_GEMM_ARITH=f32 (every fp32 GEMM on v_mfma_f32_32x32x2_f32):
    reported beside `value` so that the gain of the bf16x3 form is measured on the same box in the same run."""
    import subprocess
    env = dict(os.environ, SM3_GEMM_ARITH='f32', SM3_BENCH_NATIVE='0')
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--config', args.config, '--no-ops', '--no-cpu-baseline'] + (['--fp32'] if args.fp32 else [])
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600).stdout
        line = json.loads([ln for ln in out.splitlines() if ln.startswith('{')][-1])
        return dict(value_native_f32_mfma=line['value'], ms_per_step_native_f32_mfma=line['ms_per_step'],
                    roofline_native_f32_mfma=dict(achieved=line['roofline']['achieved'], peak=line['roofline']['peak'],
                                                  frac=line['roofline']['frac'], unit='TFLOP/s',
                                                  gemm_ms_per_step=line['roofline']['gemm_ms_per_step']))
    except Exception as e:  # noqa: BLE001
        return dict(value_native_f32_mfma=None, native_f32_mfma_error=f'{type(e).__name__}: {e}'[:200])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-ops', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--workload', choices=['backbone', 'full_model'], default='backbone',
                    help="'backbone' (default): the headline hot path; 'full_model': the whole detector's data-parallel training step")
    ap.add_argument('--config', default=DEFAULT_CONFIG, help='BASELINE.json configuration (a key of '
                    'sm3det_amd/configs/baseline_configs.json): main_SM3Det (#2, default), SM3Det_convnext_t (#3), e16t2 (#4), '
                    'SM3Det_convnext_b (#5), simple_joint (#1)')
    ap.add_argument('--amp', action='store_true', help='mixed precision (the `fp16 = dict(loss_scale="dynamic")` switch of '
                    'configs #3 / #5; implied by --config SM3Det_convnext_{t,b}): fp16 activations and GEMM operands, fp32 '
                    'accumulation / master weights / LayerNorm / router / combine, dynamic loss scale; reported as its own '
                    'dtype, never mixed with the fp32 line')
    ap.add_argument('--fp32', action='store_true', help='run the model of an AMP config in fp32 (diagnostic line: the MFMA '
                    'roofline of the ConvNeXt-B shapes); the config label says so')
    ap.add_argument('--cpu-worker', choices=['step', 'ops'], help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return _cpu_worker(args.cpu_worker, args.config)
    cfg_entry = load_config(args.config)
    args.amp = bool(args.amp or cfg_entry.get('fp16')) and not args.fp32

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return _self_launch(args.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    # SM3_BENCH_BACKEND=gloo lets two ranks share ONE GPU (the 1-GPU dev box) to exercise the N>1 control flow --
    # graph A + bucket pack, eager all-reduce, graph B on the reduced buckets -- without RCCL; never used for numbers.
    backend = os.environ.get('SM3_BENCH_BACKEND', 'nccl')
    if os.environ.get('SM3_BENCH_WATCHDOG'):  # debugging aid: dump all Python stacks and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['SM3_BENCH_WATCHDOG']), exit=True)
    local_rank = local_rank % torch.cuda.device_count() if backend != 'nccl' else local_rank
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    # SM3_BENCH_FORCE_DIST=1: initialise the process group and run the bucket / collective / split-backward flow even
    # with ONE rank (the all-reduces are identities) -- the multi-GPU code path on a 1-GPU box (tests/test_dp_rccl_gpu.py)
    force_dist = os.environ.get('SM3_BENCH_FORCE_DIST') == '1' and world == 1
    multi = world > 1 or force_dist
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from sm3det_amd import _lib, _lib_backbone as LB
    from sm3det_amd.data_parallel import BucketedGradReducer
    _lib.lib()  # fail loudly if the HIP extension is missing
    if args.workload == 'full_model':
        return full_model_dp_main(args, world, rank, multi, backend, force_dist)

    net = build_model(args.config).cuda().train()
    desc = describe_backbone(net)
    if args.amp:
        from sm3det_amd import amp
        amp.wrap_fp16_model(net)  # what Fp16OptimizerHook.before_run does (mmcv/mmcv/runner/hooks/optimizer.py:245-249)
    params = [p for p in net.parameters() if p.requires_grad]
    # N>1: the backward is replayed in two segments -- stages 3+2 (93 % of the gradient bytes) first, whose buckets are
    # all-reduced over xGMI while the backward of stages 1+0 still runs (SM3_BENCH_SPLIT=0/1 overrides; N=1 default off)
    split = os.environ.get('SM3_BENCH_SPLIT', '1' if multi else '0') == '1'
    # ... in SM3_BENCH_SEGMENTS pieces (default 4: stage 3 | stage 2 | stage 1 | stage 0 + stem): the buckets of a segment are
    # all-reduced while the NEXT segment's backward replays.  Gradient bytes by stage (ConvNeXt-T e8t2): 57 % / 37 % / 4 % / 2 %,
    # backward time 18 % / 52 % / 19 % / 11 % -- with two segments (3+2 | 1+0) 93 % of the bytes had 30 % of the backward to
    # hide behind; with four, stage 3's 0.32 GB travel during stage 2's backward (half of the pass).
    n_seg = max(2, min(4, int(os.environ.get('SM3_BENCH_SEGMENTS', '4')))) if split else 1
    seg_stages = {2: [(3, 2), (1, 0)], 3: [(3,), (2,), (1, 0)], 4: [(3,), (2,), (1,), (0,)]}.get(n_seg, [(3, 2, 1, 0)])
    named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]

    def stage_of(name):
        for st in (3, 2, 1):
            if name.startswith((f'stages.{st}.', f'downsample_layers.{st}.', f'norm{st}.')):
                return st
        return 0  # stage 0, its norm, the stem
    seg_params = [[p for n, p in reversed(named) if stage_of(n) in sts] for sts in seg_stages]
    reducer = BucketedGradReducer(params, bucket_mb=64.0, groups=seg_params if split else None, force_comm=force_dist)
    reducer.broadcast_parameters(0, module=net)
    # optimizer of local_configs/main_SM3Det.py: AdamW(lr 1e-4, betas (0.9, 0.999), wd 0.05), one param group per
    # parameter (paramwise_cfg / dynamic-lr hook), grad_clip max_norm 35 -- here one fused launch with a per-tensor lr vector
    # built from the config's own `optimizer` / `optimizer_config` dicts by the rules of mmcv's
    # DefaultOptimizerConstructor (sm3det_amd.optim.build_optimizer, pinned on that class by
    # tests/test_optim_constructor_cpu.py)
    from sm3det_amd.optim import build_optimizer
    opt = build_optimizer({'backbone': net}, cfg_entry['optimizer'], cfg_entry['optimizer_config'],
                          loss_scale='dynamic' if args.amp else None)
    assert [id(q) for g_ in opt.param_groups for q in g_['params']] == [id(q) for q in params]

    g = torch.Generator(device='cpu').manual_seed(rank)  # rank r draws its own synthetic shard
    x = torch.randn(BATCH, 3, RES, RES, generator=g).cuda()
    proj = None

    zero = torch.zeros((), device='cuda')

    def forward():
        r = net(x, ['single'])
        return r if desc['moe_blocks'] else (r, zero)  # config #1 has no MoE block: no gate loss

    def step():
        nonlocal proj
        reducer.zero_grad()
        outs, gl = forward()
        if proj is None:
            gp = torch.Generator(device='cpu').manual_seed(1000 + rank)
            proj = [torch.randn(o.shape, generator=gp).cuda().contiguous(memory_format=torch.channels_last)
                    for o in outs]
        loss = loss_fn(outs, gl, proj)
        opt.scale(loss).backward()  # identity without --amp
        reducer.finalize()
        opt.step()
        return loss

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # The whole step is sync-free (routing tables are built on the device), so it is captured once into hipGraphs and
    # replayed: ~800 launches per step would otherwise leave the GPU idle behind the Python/ctypes launch path.
    # Graph A = forward + backward (gradients adopted by autograd: no zero-fill, no accumulate launches) and, for N>1,
    # the pack of the gradients into the flat xGMI buckets; then the RCCL all-reduces of the buckets (eager, back to
    # back on RCCL's stream); then graph B = grad-clip + AdamW reading the reduced buckets (N>1) or the gradients
    # themselves (N=1).  (`--no-graph` runs eagerly with the all-reduces overlapped with backward.)
    use_graph = not args.no_graph
    run = step
    graph_loss = None
    if use_graph:
        reducer.overlap = False

        def fwd_bwd():
            reducer.zero_grad()
            outs, gl = forward()
            l = loss_fn(outs, gl, proj)
            opt.scale(l).backward()
            reducer.pack_all()
            return l

        def seg_losses(outs):
            """the step's loss split by segment: gate-loss terms and output projections of the stages a segment owns"""
            terms = net._gate_loss_terms
            nt = max(len(terms), 1)
            ls = []
            for sts in seg_stages:
                l = sum((g for i, g in terms if i in sts), zero) / nt
                for i, (o, r) in enumerate(zip(outs, proj)):
                    if i in sts:
                        l = l + (o * r).sum() * 1e-4
                ls.append(l)
            return ls

        def seg_first():
            """forward + backward of the top segment (everything above the tokens leaving stage seg_stages[1][0])"""
            reducer.zero_grad()
            outs, _ = forward()
            ls = seg_losses(outs)
            toks = net._boundary_tokens  # {stage: tokens leaving it}
            below = toks[seg_stages[1][0]]
            # autograd.grad, not backward(inputs=[..., tok]): the latter would EXECUTE the token's producer node (and free its
            # saved tensors) instead of just capturing the gradient that arrives at it
            grads = torch.autograd.grad(opt.scale(ls[0]), [below] + seg_params[0], allow_unused=True)
            for p, g in zip(seg_params[0], grads[1:]):
                p.grad = g
            reducer.pack_group(0)
            return ls, toks, grads[0]

        def seg_next(k, ls, toks, g_in):
            """backward of segment k: its own loss terms plus the gradient arriving at the tokens that leave its top stage"""
            top = toks[seg_stages[k][0]]
            last = k == len(seg_stages) - 1
            if last:
                torch.autograd.backward([opt.scale(ls[k]), top], grad_tensors=[torch.ones_like(ls[k]), g_in],
                                        inputs=seg_params[k])
                reducer.pack_group(k)
                return None
            below = toks[seg_stages[k + 1][0]]
            grads = torch.autograd.grad([opt.scale(ls[k]), top], [below] + seg_params[k],
                                        grad_outputs=[torch.ones_like(ls[k]), g_in], allow_unused=True)
            for p, g in zip(seg_params[k], grads[1:]):
                p.grad = g
            reducer.pack_group(k)
            return grads[0]

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        fence()
        # N>1: RCCL's watchdog thread may touch the HIP runtime while this thread captures -> thread-local capture mode
        cap_kw = dict(capture_error_mode='thread_local') if multi else {}
        try:
            g_fb, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            if split:
                net.stage_boundary = [sts[0] for sts in seg_stages[1:]]  # the tokens leaving the top stage of each lower segment
                g_segs = [g_fb] + [torch.cuda.CUDAGraph() for _ in seg_stages[1:]]
                with torch.cuda.graph(g_fb, **cap_kw):
                    ls, toks, g_in = seg_first()
                for k in range(1, len(seg_stages)):
                    with torch.cuda.graph(g_segs[k], pool=g_fb.pool(), **cap_kw):
                        g_in = seg_next(k, ls, toks, g_in)
                        if k == len(seg_stages) - 1:
                            graph_loss = sum(ls[1:], ls[0]).detach()
                net.stage_boundary = None
            else:
                with torch.cuda.graph(g_fb, **cap_kw):
                    graph_loss = fwd_bwd()
            keep_alive = [p.grad for p in params]  # the tensors the graphs write the gradients to  # noqa: F841
            g_fb.replay()
            if split:
                for gk in g_segs[1:]:
                    gk.replay()
            reducer.finalize(repack=False)  # N>1: p.grad -> slices of the reduced buckets
            opt.refresh_grad_pointers()
            with torch.cuda.graph(g_opt, pool=g_fb.pool(), **cap_kw):
                opt.step()

            if split:
                def run():
                    for k, gk in enumerate(g_segs):       # forward + backward of the top segment, then segment by segment:
                        gk.replay()                        # ... the backward of segment k (+ the pack of its buckets)
                        reducer.allreduce_group_async(k)   # RCCL stream: overlaps with the replay of segment k + 1
                    reducer.finalize(repack=False)
                    g_opt.replay()
                    return graph_loss
            else:
                def run():
                    g_fb.replay()
                    reducer.finalize(repack=False)  # world 1: no-op; world > 1: bucketed all-reduce + mean
                    g_opt.replay()
                    return graph_loss
        except Exception as e:  # keep the measurement alive: eager launches, all-reduces overlapped with backward
            print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr)
            use_graph = False
            net.stage_boundary = None
            reducer.overlap = True
            torch.cuda.synchronize()
            run = step

    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    trace = os.environ.get('SM3_BENCH_TRACE') == '1'  # debugging aid: loss and a parameter checksum after every step
    for _ in range(args.steps):
        loss = run()
        if trace:
            torch.cuda.synchronize()
            chk = float(torch.stack([p.detach().double().abs().sum() for p in params]).sum())
            gchk = float(torch.stack([p.grad.detach().double().abs().sum() for p in params if p.grad is not None]).sum())
            print(f'[trace] loss {float(loss):.6f} |params| {chk:.6f} |grads| {gchk:.6f}', file=sys.stderr)
            bad = [(n, float(q.grad.abs().max())) for n, q in net.named_parameters()
                   if q.grad is not None and not (float(q.grad.abs().max()) < 1e6)]
            if bad:
                print(f'[trace] suspicious gradients: {bad[:6]}', file=sys.stderr)
    fence()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * BATCH * args.steps / dt
    # replicas must stay bit-identical (same averaged gradients on every rank): spread of a parameter checksum
    replica_spread = 0.0
    if multi:
        chk = torch.stack([p.detach().double().sum() for p in params]).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replica_spread = float((hi - lo).item())

    # ---- roofline of the dominant kernel: one more identical step with HIP events around every launch ---------
    roofline = None
    kernels = {}
    # every rank runs the step (it contains the bucket all-reduces); only rank 0 records the per-launch events.
    # The weight-gradient side stream is switched off for this pass: per-kernel durations must be those of the kernel
    # alone (two kernels sharing the chip would each look slower), which is also what SM3_WGRAD_STREAM=0 + rocprofv3
    # --kernel-trace reports (profiles/).
    from sm3det_amd import backbone_ops as _bops
    overlap_was, pair_was = _bops.OVERLAP_WGRAD, _bops.PAIR_DGRAD
    _bops.OVERLAP_WGRAD, _bops.PAIR_DGRAD = False, 0
    # Three such steps, the one with the smallest summed bracket time is reported: a bracket also holds the host's launch
    # latency whenever the stream runs dry between e0 and the kernel, which a busy host inflates (observed once: 113 us
    # per GEMM launch instead of 103, with an unchanged 19.95 ms graph-replayed step).
    prof = None
    for _ in range(3):
        if rank == 0:
            LB.PROFILE = []
        step()
        torch.cuda.synchronize()
        if rank == 0:
            cand, LB.PROFILE = LB.PROFILE, None
            tot = sum(e0.elapsed_time(e1) for _n, _f, _b, e0, e1 in cand)
            if prof is None or tot < prof[0]:
                prof = (tot, cand)
    _bops.OVERLAP_WGRAD, _bops.PAIR_DGRAD = overlap_was, pair_was
    if rank == 0:
        prof = prof[1]
        for name, flops, nbytes, e0, e1 in prof:
            k = kernels.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            k['launches'] += 1
            k['ms'] += e0.elapsed_time(e1)
            k['flops'] += flops
            k['bytes'] += nbytes
        gem = [v for n, v in kernels.items() if n.startswith('gemm_f')]
        g_ms = sum(v['ms'] for v in gem)
        g_fl = sum(v['flops'] for v in gem)
        g_n = sum(v['launches'] for v in gem)
        achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        g_by = sum(v['bytes'] for v in gem)
        # HBM bytes per launch from the committed PMC passes of this command (rocprofv3 cannot run inside bench.py); the
        # passes exist for the headline workload (fp32 and AMP), other configs report null
        traffic = traffic_file = None
        if args.config == DEFAULT_CONFIG or (args.amp and args.config == 'SM3Det_convnext_t'):  # the same backbone
            for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):
                cand = os.path.join('profiles', rnd, 'pmc_traffic_amp.json' if args.amp else 'pmc_traffic.json')
                try:
                    with open(os.path.join(ROOT, cand)) as f:
                        traffic = round(json.load(f)['gemm_family']['hbm_bytes_per_launch'])
                    traffic_file = cand
                    break
                except Exception:
                    continue
        roofline = dict(bound='mfma', kernel='gemm_f32_kernel (NT/NN/TN, fp32 v_mfma_f32_32x32x2_f32)',
                        achieved=round(achieved, 2), peak=MI355X_FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                        frac=round(achieved / MI355X_FP32_MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                        traffic_source=None if traffic_file is None else
                        f'{traffic_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH x2 + WRITE, '
                        'per launch incl. the slice-reduce pass of the TN launches; scripts/collect_artifacts.sh)',
                        algorithmic_bytes_per_launch=round(g_by / max(g_n, 1)),
                        launches_per_step=g_n, avg_launch_us=round(g_ms / max(g_n, 1) * 1e3, 2),
                        algorithmic_gflop_per_step=round(g_fl / 1e9, 1),
                        gemm_ms_per_step=round(g_ms, 3),
                        other_kernels_ms_per_step=round(sum(v['ms'] for n, v in kernels.items()
                                                            if not n.startswith('gemm_f')), 3))
        if not args.amp and LB.ARITH32 == 2:
            # bf16x3: the fp32 contractions run as six v_mfma_f32_32x32x16_bf16 products per fp32 product (operands split exactly
            # into three bf16 pieces in the loader, fp32 accumulation): the matrix-pipe ceiling is 2.5 PF / 6 fp32-equivalent
            gbs = g_by / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
            roofline.update(kernel='gemm_f32_kernel<.., F16 = 2> (bf16x3: fp32 tensors, 3 exact bf16 pieces per operand element, '
                            '6 x v_mfma_f32_32x32x16_bf16 per 16 k, fp32 accumulate)',
                            peak=round(MI355X_BF16X3_PEAK_TFLOPS, 1), frac=round(achieved / MI355X_BF16X3_PEAK_TFLOPS, 4),
                            peak_note='dense bf16 MFMA peak 2500 TF/s / 6 products per fp32 product; achieved = fp32 FLOP / time',
                            x_native_fp32_mfma_peak=round(achieved / MI355X_FP32_MFMA_PEAK_TFLOPS, 3),
                            hbm_gbs=round(gbs, 1), hbm_frac=round(gbs / MI355X_HBM_PEAK_GBS, 4))
            if traffic_file is not None and not any(r in traffic_file for r in ('r05', 'r06')):
                roofline.update(traffic=None, traffic_source=None)  # the older passes measured the native-fp32 kernels
        if args.amp:  # fp16 operands: 16x the matrix rate of fp32 -> priced against BOTH ceilings; the nearer one is `bound`,
            # and `regime` says "latency" while neither fraction reaches 0.5 (launches of ~30 us on a 256-CU part)
            gbs = g_by / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
            hbm_frac, mfma_frac = gbs / MI355X_HBM_PEAK_GBS, achieved / MI355X_F16_MFMA_PEAK_TFLOPS
            roofline.update(regime='latency' if max(hbm_frac, mfma_frac) < 0.5 else ('hbm' if hbm_frac >= mfma_frac else 'mfma'),
                            hbm_frac=round(hbm_frac, 4), mfma_frac=round(mfma_frac, 4),
                            mfma_peak_tflops=MI355X_F16_MFMA_PEAK_TFLOPS)
            roofline.update(bound='hbm', kernel='gemm_f32_kernel<.., F16, IO> (v_mfma_f32_32x32x16_f16; activations stored '
                            'fp16 in HBM, weights / residual stream / C-wide gradients fp32 and rounded in the loader)',
                            achieved=round(gbs, 1), peak=MI355X_HBM_PEAK_GBS, unit='GB/s',
                            frac=round(gbs / MI355X_HBM_PEAK_GBS, 4), mfma_tflops=round(achieved, 1))

    result = None
    if rank == 0:
        arch_name = {'tiny': 'ConvNeXt-T', 'base': 'ConvNeXt-B', 'small': 'ConvNeXt-S'}.get(desc['arch'], desc['arch'])
        moe_name = f"e{desc['experts']}t{desc['top_k']}" if desc['moe_blocks'] else 'dense'
        result = {
            'metric': f'train imgs/sec SM3Det {arch_name} {moe_name} @{RES}^2 bs{BATCH}/GPU (hot path: MoE backbone train step)',
            'value': round(value, 3), 'unit': 'imgs/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16' if args.amp else 'f32', 'data': 'synthetic',
            'config': {'workload': (f"{cfg_entry['file']} (BASELINE.json config #{cfg_entry['baseline_config']}) backbone, built "
                                    f"from the config file's own dict: ConvNeXt_moe_MultiInput {desc['arch']}, "
                                    f"{desc['experts']} experts top-{desc['top_k']}, {desc['moe_blocks']} MoE + "
                                    f"{desc['dense_blocks']} dense blocks, {desc['params_m']} M parameters; " +
                                    ('AMP (fp16 = dict(loss_scale="dynamic")): fp16 activations / GEMM operands, fp32 accumulate / '
                                     'master weights / LayerNorm / router / combine, dynamic loss scale; ' if args.amp else
                                     ('run in fp32 although the config file enables AMP (--fp32, diagnostic); '
                                      if cfg_entry.get('fp16') else '')) +
                                    'fwd+bwd + bucketed grad all-reduce + grad-clip(35)+AdamW (per-parameter lr); synthetic '
                                    f'randn({BATCH},3,{RES},{RES}) per GPU, random-init weights; neck/heads timed separately in ops_us'),
                       'name': args.config, 'baseline_config': cfg_entry['baseline_config'], 'backbone': desc,
                       'gemm_arith': ('fp16 operands, fp32 accumulate (AMP)' if args.amp else
                                      ('bf16x3: fp32 tensors; each operand element split exactly into three bf16 pieces, six bf16 MFMA '
                                       'products (i+j<=2) accumulated in fp32 -- fp32-equivalent (error vs fp64 <= the native fp32 '
                                       "kernel's x1.5 per shape: profiles/r06/gemm_b3_sweep.txt); SM3_GEMM_ARITH=f32 runs the native "
                                       'v_mfma_f32_32x32x2_f32 form (value_native_f32_mfma)' if LB.ARITH32 == 2 else
                                       'native fp32 MFMA (v_mfma_f32_32x32x2_f32)')),
                       'global_batch': world * BATCH, 'resolution': RES, 'parallelism': f'dp{world}',
                       'grad_buckets': reducer.num_buckets, 'hip_graph': bool(use_graph),
                       'wgrad_side_stream': bool(overlap_was), 'split_backward': bool(split and use_graph),
                       # concurrent backward partners (backbone_ops._paired): input-gradient GEMM on the main stream, its layer's
                       # weight gradient right behind it on a side stream; level 4 = FC2, FC1, depthwise and gate pairs; the side
                       # stream is joined once per backward pass.  `roofline` is measured with the pairs OFF (each kernel alone).
                       'backward_pairs_level': int(pair_was), 'side_stream_joined_per': ('pass' if _bops.DEFER_JOIN else 'block'),
                       'gemm_eq_prio': os.environ.get('SM3_EQ_PRIO', '2 (auto: single-round launches)'),
                       'backward_segments': (n_seg if (split and use_graph) else 1),
                       'dist_backend': (backend if multi else None), 'collective_avg': bool(reducer._avg),
                       # share of the gradient bytes the backward kernels wrote straight into the bucket slices (no pack copy)
                       'grad_bytes_in_place_frac': (round(reducer.pack_stats['in_place_bytes'] /
                                                          max(reducer.pack_stats['in_place_bytes'] +
                                                              reducer.pack_stats['copied_bytes'], 1), 4) if multi else None),
                       'replica_checksum_spread': replica_spread},
            'loss': float(loss.detach()),
            'roofline': roofline,
            'kernels_ms_per_step': {n: round(v['ms'], 3) for n, v in sorted(kernels.items(), key=lambda kv: -kv[1]['ms'])},
            'kernels_launches_per_step': {n: v['launches'] for n, v in kernels.items()},
            'kernels_algorithmic_mb_per_step': {n: round(v['bytes'] / 1e6, 1) for n, v in kernels.items() if v['bytes'] > 0},
            # achieved GB/s of the HBM-bound kernels: ALGORITHMIC bytes (each operand once) / time
            'kernels_algorithmic_gbs': {n: round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) for n, v in
                                        sorted(kernels.items(), key=lambda kv: -kv[1]['ms'])
                                        if v['bytes'] > 0 and v['ms'] > 0 and not n.startswith('gemm_f')},
        }
        # SM3_BENCH_OPS=full: only the full-model workload runs after the headline (profiling aid, like =slice)
        only_full = os.environ.get('SM3_BENCH_OPS') == 'full'
        if world == 1 and not args.no_ops:
            result['ops_us'] = {} if only_full else ops_microbench()
            result['ops_roofline'] = ops_roofline(result['ops_us'])
            t_slice = result['ops_us'].get('detector_slice_train_step_bs2_1024')
            if t_slice:  # second named value, never mixed into `value`: see ops_microbench()
                result['full_slice_imgs_per_sec'] = round(BATCH / (t_slice * 1e-6), 2)
                result['full_slice_workload'] = (
                    'two-stage (RGB) branch of main_SM3Det.py as one training step at bs2 1024^2: backbone + MultitaskFPN + '
                    'OrientedRPNHead.forward_train (real targets + losses + proposals) + OrientedStandardRoIHead.forward_train '
                    '(real targets + losses) + backward + clip + AdamW; excludes the SAR GFL branch and the data pipeline')
            if os.environ.get('SM3_BENCH_OPS') != 'slice':
                try:
                    fm = full_model_bench()
                    result['full_model'] = fm
                    result['full_model_imgs_per_sec'] = fm['imgs_per_sec']
                    result['full_model_workload'] = (
                        'TriSourceDetector of main_SM3Det.py built from the config dict, one training step at the native mix 2 SAR + '
                        '1 RGB + 1 IR @1024^2: backbone (4 images) + MultitaskFPN x3 + GFLHead (ATSS, QFL/DFL/GIoU: device kernels, '
                        'restated from mmdet, unpinned) + 2 x (OrientedRPNHead + OrientedStandardRoIHead) with real targets / losses + the dynamic-lr policy of '
                        "the config's lr_config (one per-tensor lr from the 11 loss EMAs, on the device) + backward + clip + "
                        'AdamW over 178 M parameters; device-resident synthetic inputs, no data pipeline')
                except Exception as e:  # noqa: BLE001  (never lose the headline line to the extra workload)
                    result['full_model'] = dict(error=f'{type(e).__name__}: {e}'[:300])
        if world == 1 and not args.amp and LB.ARITH32 == 2 and os.environ.get('SM3_BENCH_NATIVE', '1') == '1':
            result.update(native_f32_mfma_line(args))  # the same step with the native fp32 matrix instruction, beside `value`
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.config)
        else:
            result['cpu_baseline'] = None
        print(json.dumps(result), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
