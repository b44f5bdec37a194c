"""GPU tier: MultiTensorAdamW (one launch, per-tensor lr vector, fused global-norm clipping) vs torch.optim.AdamW +
torch.nn.utils.clip_grad_norm_ -- the reference's OptimizerHook arithmetic.  Tolerance 1e-6 abs/rel (same fp32 ops,
different evaluation order inside the update)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(96, 3, 4, 4), (96,), (8, 384, 96), (1,), (33000,), (17, 5), (768, 3072)]
    return [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]


@pytest.mark.parametrize('max_norm', [None, 35.0, 0.5])
def test_matches_torch_adamw_with_per_tensor_lr_and_clip(max_norm):
    from sm3det_amd.optim import MultiTensorAdamW
    a, b = _params(0), _params(0)
    groups_a = [dict(params=[p], lr=1e-3 * (1 + i), weight_decay=0.05 if p.dim() > 1 else 0.0) for i, p in enumerate(a)]
    groups_b = [dict(params=[p], lr=1e-3 * (1 + i), weight_decay=0.05 if p.dim() > 1 else 0.0) for i, p in enumerate(b)]
    ref = torch.optim.AdamW(groups_a, betas=(0.9, 0.999), eps=1e-8)
    mine = MultiTensorAdamW(groups_b, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=max_norm)
    g = torch.Generator().manual_seed(1)
    for step in range(4):
        grads = [torch.randn(p.shape, generator=g).cuda() * (3.0 if step == 2 else 0.3) for p in a]
        for p, q, gr in zip(a, b, grads):
            p.grad = gr.clone()
            q.grad = gr.clone()
        if step == 2:  # the dynamic-lr hook rewrites every group's lr each step
            for ga, gb in zip(ref.param_groups, mine.param_groups):
                ga['lr'] *= 0.5
                gb['lr'] *= 0.5
        if max_norm:
            tn = torch.nn.utils.clip_grad_norm_(a, max_norm)
        ref.step()
        mine.step()
        if max_norm:
            assert abs(float(mine.grad_norm) - float(tn)) <= 1e-4 * float(tn)
        for p, q in zip(a, b):
            assert torch.allclose(p, q, rtol=2e-6, atol=2e-6), (step, p.shape, (p - q).abs().max())


def test_graph_capture_and_dynamic_lr_policy():
    from sm3det_amd.optim import DynamicLrPolicy, MultiTensorAdamW
    ps = _params(3)
    for p in ps:
        p.grad = torch.randn_like(p)
    opt = MultiTensorAdamW([dict(params=[p]) for p in ps], lr=1e-3, weight_decay=0.05, max_grad_norm=35.0)
    opt.step()  # builds the tables eagerly
    before = [p.detach().clone() for p in ps]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        opt.step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        opt.step()
    g.replay()
    torch.cuda.synchronize()
    assert float(opt._step) == 3.0  # eager + side-stream + one replay (capture itself does not execute)
    assert all(not torch.equal(a, b) for a, b in zip(before, ps))
    pol = DynamicLrPolicy(warmup_iters=0)
    names = ['backbone.stages.0.0.gamma', 'sar_bbox_head.gfl_cls.weight', 'rgb_roi_head.bbox_head.fc_cls.weight']
    lv = {'sar_loss_cls': 1.0, 'sar_loss_bbox': 2.0, 'sar_loss_dfl': 0.5, 'rgb_loss_rpn_cls': 0.3, 'rgb_loss_rpn_bbox': 0.2,
          'rgb_loss_cls': 0.7, 'rgb_loss_bbox': 0.6, 'ifr_loss_rpn_cls': 0.3, 'ifr_loss_rpn_bbox': 0.2,
          'ifr_loss_cls': 0.7, 'ifr_loss_bbox': 0.6, 'gate_loss': 0.01}
    m1 = pol.multipliers(lv, names)
    m2 = pol.multipliers({k: v * (2.0 if k.startswith('sar') else 1.0) for k, v in lv.items()}, names)
    assert set(m1) == set(names) and all(v > 0 for v in m2.values())
    assert m2['backbone.stages.0.0.gamma'] == min(m2.values())  # backbone_policy='min'


@pytest.mark.parametrize('backbone_policy,head_policy', [('sigmoid_kl', 'normal'), ('min', 'reverse'), ('kl', 'None'),
                                                         ('avg', 'normal'), ('max', 'normal')])
@pytest.mark.parametrize('as_run', [True, False])  # warm-up as a reference run executes it (default) / mmcv's documented ramp
def test_device_dynamic_lr_matches_the_host_flow_pinned_on_the_reference_hook(backbone_policy, head_policy, as_run):
    """``DeviceDynamicLr`` (sm3_dla_lr: one launch, no host read) against ``dynamic_lr_after_train_iter`` -- the host form
    that tests/test_dla_cpu.py pins on the reference's own ``DynamicLrUpdaterHook.after_train_iter`` -- over a recorded loss
    sequence that crosses the linear warm-up, both step-decay milestones and has list-valued losses (the GFL head's
    per-level lists are summed): the lr vector the AdamW launch reads equals the hook's within 1e-6, the EMAs within 1e-12."""
    from sm3det_amd.optim import (DeviceDynamicLr, DynamicLrPolicy, MultiTensorAdamW, dynamic_lr_after_train_iter)
    names = ['backbone.stages.0.0.gamma', 'backbone.stages.2.1.ffn.w1', 'neck.lateral_convs.0.conv.weight',
             'sar_bbox_head.gfl_cls.weight', 'rgb_rpn_head.rpn_conv.weight', 'rgb_roi_head.bbox_head.fc_cls.weight',
             'ifr_rpn_head.rpn_reg.bias', 'ifr_roi_head.bbox_head.shared_fcs.0.weight']
    ps = [torch.nn.Parameter(torch.randn(8, 4, device='cuda')) for _ in names]
    for p in ps:
        p.grad = torch.randn_like(p)
    base = [1e-4 * (1 + 0.1 * i) for i in range(len(ps))]
    opt = MultiTensorAdamW([dict(params=[p], lr=b) for p, b in zip(ps, base)], lr=1e-4, weight_decay=0.05)
    extra = {'T': 3, 'b': 0.4, 'ema': 0.001, 'backbone_policy': backbone_policy, 'head_policy': head_policy}
    W, steps_at = 5, [9, 12]
    dla = DeviceDynamicLr(opt, names, step=steps_at, gamma=0.1, extra_args=extra, warmup='linear', warmup_iters=W,
                          warmup_ratio=1.0 / 3, warmup_as_run=as_run)
    pol = DynamicLrPolicy(T=3, b=0.4, ema=0.001, backbone_policy=backbone_policy, head_policy=head_policy, warmup_iters=W)
    keys = ['sar_loss_cls', 'sar_loss_bbox', 'sar_loss_dfl', 'rgb_loss_rpn_cls', 'rgb_loss_rpn_bbox', 'rgb_loss_cls',
            'rgb_loss_bbox', 'ifr_loss_rpn_cls', 'ifr_loss_rpn_bbox', 'ifr_loss_cls', 'ifr_loss_bbox']
    for it in range(15):
        g = torch.Generator().manual_seed(100 + it)
        v = (torch.rand(len(keys) + 4, generator=g) * 2 + 0.05)
        dev = {'gate_loss': v[-1].cuda(), 'loss': v.sum().cuda()}  # not reweight keys: ignored
        host = {}
        for i, k in enumerate(keys):
            if k == 'sar_loss_bbox':  # a per-level list, as GFLHead returns it
                parts = [v[i] * 0.5, v[-2] * 0.25, v[-3] * 0.25]
                dev[k] = [q.cuda() for q in parts]
                host[k] = [float(q) for q in parts]
            else:
                dev[k] = v[i].cuda()
                host[k] = float(v[i])
        dla.set_iter(it)
        dla.update(dev)
        want = dynamic_lr_after_train_iter(pol, host, names, base, it, steps_at, 0.1, W, 1.0 / 3, warmup='linear',
                                           as_run=as_run)
        got = opt._lr.cpu().tolist()
        for a, b, n in zip(got, want, names):
            assert abs(a - b) <= 1e-6 * abs(b) + 1e-12, (it, n, a, b)
        opt.step()  # the optimizer must keep the device lr vector (not rewrite it from its host groups)
        assert opt._lr.cpu().tolist() == got
    emas, updates, iters = dla.state_host()
    assert updates == iters == 15
    for e, m in zip(emas, pol.history):
        assert abs(e - m.get()) <= 1e-6 * abs(m.get())  # (the list-valued loss is summed in fp32 on the device, as mmdet does)
    # the launch is capturable: no host read of a loss
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            dla.update(dev)
    torch.cuda.current_stream().wait_stream(s)
    gr.replay()
    torch.cuda.synchronize()
    assert dla.state_host()[1] == 16
