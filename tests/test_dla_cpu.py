"""CPU tier: ``sm3det_amd.optim.DynamicLrPolicy`` against the REFERENCE hook's own
``get_dynamic_lr`` (mmrotate/core/hook/dynamic_lr.py:107-175), imported live when /root/reference exists, over a
sequence of steps (the EMA history matters), for every backbone / head policy the policy object implements."""
import pytest
import torch

PARAMS = ['backbone.stages.0.0.gamma', 'backbone.stages.2.1.ffn.w1', 'neck.lateral_convs.0.conv.weight',
          'sar_bbox_head.gfl_cls.weight', 'rgb_rpn_head.rpn_conv.weight', 'rgb_roi_head.bbox_head.fc_cls.weight',
          'ifr_rpn_head.rpn_reg.bias', 'ifr_roi_head.bbox_head.shared_fcs.0.weight']


def _losses(step):
    g = torch.Generator().manual_seed(step)
    names = ['sar_loss_cls', 'sar_loss_bbox', 'sar_loss_dfl', 'rgb_loss_rpn_cls', 'rgb_loss_rpn_bbox', 'rgb_loss_cls',
             'rgb_loss_bbox', 'ifr_loss_rpn_cls', 'ifr_loss_rpn_bbox', 'ifr_loss_cls', 'ifr_loss_bbox']
    v = (torch.rand(len(names), generator=g) * 2 + 0.05).tolist()
    d = dict(zip(names, v))
    d['loss'] = sum(v)          # not a reweight key: ignored by both
    d['gate_loss'] = 0.01
    return d


@pytest.mark.parametrize('backbone_policy', ['min', 'avg', 'max', 'kl', 'sigmoid_kl'])
@pytest.mark.parametrize('head_policy', ['normal', 'reverse', 'None'])
@pytest.mark.parametrize('warmup_iters', [0, 3])
def test_policy_matches_reference_hook(backbone_policy, head_policy, warmup_iters):
    from oracle import ref_dla
    if not ref_dla.available():
        pytest.skip('/root/reference not present (GPU box)')
    from sm3det_amd.optim import DynamicLrPolicy
    mod = ref_dla.load()
    extra = {'T': 5, 'b': 0.5, 'ema': 0.005, 'backbone_policy': backbone_policy, 'head_policy': head_policy}
    hook = mod.DynamicLrUpdaterHook(step=[10 ** 9], extra_args=extra, by_epoch=False, warmup_iters=warmup_iters)
    hook.base_lr = [1.0] * len(PARAMS)
    hook.param_groups_param_names_mapping = dict(enumerate(PARAMS))
    pol = DynamicLrPolicy(T=5, b=0.5, ema=0.005, backbone_policy=backbone_policy, head_policy=head_policy,
                          warmup_iters=warmup_iters)

    class Runner:
        iter, epoch = 0, 0
    r = Runner()
    for step in range(8):
        lv = _losses(step)
        r.outputs = {'log_vars': lv}
        r.iter = step
        ref = hook.get_dynamic_lr(r)
        got = pol.multipliers(lv, PARAMS)
        for i, n in enumerate(PARAMS):
            assert abs(got[n] - ref[i]) <= 1e-6 * max(1.0, abs(ref[i])), (step, n, got[n], ref[i])
    # shared (backbone / neck) parameters follow the backbone policy, head parameters their own sub-network
    assert got[PARAMS[0]] == got[PARAMS[1]] == got[PARAMS[2]]


def test_policy_shapes_without_reference():
    from sm3det_amd.optim import DynamicLrPolicy
    pol = DynamicLrPolicy(warmup_iters=0)
    m = pol.multipliers(_losses(0), PARAMS)
    assert set(m) == set(PARAMS) and all(v > 0 for v in m.values())
    # 11 equal losses with equal history -> every multiplier 1
    pol2 = DynamicLrPolicy()
    eq = {k: 1.0 for k in _losses(0) if k.endswith(('cls', 'bbox', 'dfl'))}
    pol2.multipliers(eq, PARAMS)
    m2 = pol2.multipliers(eq, PARAMS)
    assert all(abs(v - 1.0) < 1e-9 for v in m2.values())


@pytest.mark.parametrize('backbone_policy,head_policy', [('sigmoid_kl', 'normal'), ('min', 'reverse')])
def test_after_train_iter_flow_with_linear_warmup_matches_reference_hook(backbone_policy, head_policy):
    """The whole per-iteration flow of the hook -- `after_train_iter` (dynamic_lr.py:192-217): linear warm-up with the EMAs
    updating underneath, then `get_dynamic_lr` with the step decay of `get_lr` -- run live against
    `sm3det_amd.optim.dynamic_lr_after_train_iter` (the host form the device kernel is compared with on the GPU), with the
    extra_args of local_configs/main_SM3Det.py:291-300 (T 3, b 0.4, ema 0.001)."""
    from oracle import ref_dla
    if not ref_dla.available():
        pytest.skip('/root/reference not present (GPU box)')
    from sm3det_amd.optim import DynamicLrPolicy, dynamic_lr_after_train_iter
    mod = ref_dla.load()
    extra = {'T': 3, 'b': 0.4, 'ema': 0.001, 'backbone_policy': backbone_policy, 'head_policy': head_policy}
    W, steps_at = 5, [9, 12]
    hook = mod.DynamicLrUpdaterHook(step=steps_at, gamma=0.1, extra_args=extra, by_epoch=False, warmup='linear',
                                    warmup_iters=W, warmup_ratio=1.0 / 3)
    base = [1e-4 * (1 + 0.1 * i) for i in range(len(PARAMS))]
    hook.base_lr = list(base)
    hook.param_groups_param_names_mapping = dict(enumerate(PARAMS))
    pol = DynamicLrPolicy(T=3, b=0.4, ema=0.001, backbone_policy=backbone_policy, head_policy=head_policy, warmup_iters=W)

    class Runner:
        iter, epoch = 0, 0
    r = Runner()
    for it in range(15):
        lv = _losses(100 + it)
        r.outputs = {'log_vars': lv}
        r.iter = it
        # mmcv sets regular_lr in before_train_epoch from get_lr; for an iteration-based run it is the decayed base lr
        hook.regular_lr = [hook.get_lr(r, b) for b in hook.base_lr]
        hook.after_train_iter(r)
        ref = hook.last_set
        got = dynamic_lr_after_train_iter(pol, lv, PARAMS, base, it, steps_at, 0.1, W, 1.0 / 3)
        for i, n in enumerate(PARAMS):
            assert abs(got[i] - ref[i]) <= 1e-6 * abs(ref[i]), (it, n, got[i], ref[i])
