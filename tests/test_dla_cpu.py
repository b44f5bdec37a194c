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


class _Model(torch.nn.Module):
    """one distinct-valued parameter per name of PARAMS (the hook's before_run maps optimizer groups to names by VALUE equality,
    dynamic_lr.py:178-184)"""

    def __init__(self):
        super().__init__()
        self._names = {}
        for i, n in enumerate(PARAMS):
            key = f'p{i}'
            self.register_parameter(key, torch.nn.Parameter(torch.full((3,), float(i + 1))))
            self._names[key] = n

    def named_parameters(self, *a, **k):
        for key, p in super().named_parameters(*a, **k):
            yield self._names[key], p


def _runner_and_hook(mod, base, **hook_kw):
    """an IterBasedRunner-shaped object: one optimizer group per parameter (the dynamic hook asserts that), and the hook
    taken through the calls the runner issues before the first iteration: before_run, then `before_epoch` -- NOT
    before_train_epoch (mmcv/mmcv/runner/iter_based_runner.py run(): call_hook('before_run'), call_hook('before_epoch'))"""
    model = _Model()
    opt = torch.optim.SGD([dict(params=[p], lr=b) for (_, p), b in zip(model.named_parameters(), base)], lr=1.0)

    class Runner:
        iter, epoch = 0, 0
    r = Runner()
    r.model, r.optimizer = model, opt
    hook = mod.DynamicLrUpdaterHook(by_epoch=False, **hook_kw)  # IterBasedRunner sets by_epoch=False on the lr config
    hook.before_run(r)
    assert hook.param_groups_param_names_mapping == dict(enumerate(PARAMS))
    getattr(hook, 'before_epoch', lambda _r: None)(r)
    return r, hook, opt


@pytest.mark.parametrize('backbone_policy,head_policy', [('sigmoid_kl', 'normal'), ('min', 'reverse')])
def test_after_train_iter_flow_as_the_runner_drives_it(backbone_policy, head_policy):
    """The whole per-iteration flow of the hook as an IterBasedRunner run executes it -- the reference hook on the
    reference's OWN mmcv LrUpdaterHook (oracle/ref_dla.py loads both files unmodified), driven through before_run /
    before_epoch / before_train_iter / after_train_iter with NO field set by hand -- against
    `sm3det_amd.optim.dynamic_lr_after_train_iter` (the host form the device kernel is compared with on the GPU), with the
    extra_args of local_configs/main_SM3Det.py:291-300 (T 3, b 0.4, ema 0.001, warmup='linear', ratio 1/3).
    As run, the warm-up leaves the lr at its initial value (regular_lr is never installed: see the function's docstring)."""
    from oracle import ref_dla
    if not ref_dla.available():
        pytest.skip('/root/reference not present (GPU box)')
    from sm3det_amd.optim import DynamicLrPolicy, dynamic_lr_after_train_iter
    mod = ref_dla.load()
    extra = {'T': 3, 'b': 0.4, 'ema': 0.001, 'backbone_policy': backbone_policy, 'head_policy': head_policy}
    W, steps_at = 5, [9, 12]
    base = [1e-4 * (1 + 0.1 * i) for i in range(len(PARAMS))]
    r, hook, opt = _runner_and_hook(mod, base, step=steps_at, gamma=0.1, extra_args=extra, warmup='linear',
                                    warmup_iters=W, warmup_ratio=1.0 / 3)
    pol = DynamicLrPolicy(T=3, b=0.4, ema=0.001, backbone_policy=backbone_policy, head_policy=head_policy, warmup_iters=W)
    saw_full_lr_in_warmup = False
    for it in range(15):
        lv = _losses(100 + it)
        r.iter = it
        hook.before_train_iter(r)
        r.outputs = {'log_vars': lv}
        hook.after_train_iter(r)
        ref = [g['lr'] for g in opt.param_groups]
        got = dynamic_lr_after_train_iter(pol, lv, PARAMS, base, it, steps_at, 0.1, W, 1.0 / 3, warmup='linear')
        for i, n in enumerate(PARAMS):
            assert abs(got[i] - ref[i]) <= 1e-6 * abs(ref[i]), (it, n, got[i], ref[i])
        if it < W:
            saw_full_lr_in_warmup = ref == base
            assert saw_full_lr_in_warmup, 'the reference run trains its warm-up iterations at the initial lr'
    assert hook.regular_lr == []  # the mechanism: never installed under an iteration-based runner


def test_warmup_iters_gate_without_warmup():
    """warmup=None with warmup_iters = 3: no warm-up branch, but get_dynamic_lr's `history.steps < warmup_iters` gate keeps
    the head weights at 1 for the first three EMA updates (dynamic_lr.py:124)"""
    from oracle import ref_dla
    if not ref_dla.available():
        pytest.skip('/root/reference not present (GPU box)')
    from sm3det_amd.optim import DynamicLrPolicy, dynamic_lr_after_train_iter
    mod = ref_dla.load()
    extra = {'T': 3, 'b': 0.4, 'ema': 0.001, 'backbone_policy': 'min', 'head_policy': 'normal'}
    base = [1e-4] * len(PARAMS)
    r, hook, opt = _runner_and_hook(mod, base, step=[100], gamma=0.1, extra_args=extra, warmup=None, warmup_iters=3)
    pol = DynamicLrPolicy(T=3, b=0.4, ema=0.001, backbone_policy='min', head_policy='normal', warmup_iters=3)
    for it in range(6):
        lv = _losses(200 + it)
        r.iter = it
        hook.before_train_iter(r)
        r.outputs = {'log_vars': lv}
        hook.after_train_iter(r)
        ref = [g['lr'] for g in opt.param_groups]
        got = dynamic_lr_after_train_iter(pol, lv, PARAMS, base, it, [100], 0.1, 3, 0.1, warmup=None)
        for i, n in enumerate(PARAMS):
            assert abs(got[i] - ref[i]) <= 1e-6 * abs(ref[i]), (it, n, got[i], ref[i])
        if it < 3:
            assert all(abs(x - 1e-4) < 1e-12 for x in ref[3:])  # head sub-networks: weight 1 while the gate holds


def test_documented_linear_ramp_option():
    """as_run=False: mmcv's documented ramp (lr_updater.py:75-92), checked against the restated base class of round 5"""
    from oracle import ref_dla
    if not ref_dla.available():
        pytest.skip('/root/reference not present (GPU box)')
    from sm3det_amd.optim import DynamicLrPolicy, dynamic_lr_after_train_iter
    mod = ref_dla.load(real_base=False)
    extra = {'T': 3, 'b': 0.4, 'ema': 0.001, 'backbone_policy': 'sigmoid_kl', 'head_policy': 'normal'}
    W, steps_at = 5, [9, 12]
    hook = mod.DynamicLrUpdaterHook(step=steps_at, gamma=0.1, extra_args=extra, by_epoch=False, warmup='linear',
                                    warmup_iters=W, warmup_ratio=1.0 / 3)
    base = [1e-4 * (1 + 0.1 * i) for i in range(len(PARAMS))]
    hook.base_lr = list(base)
    hook.param_groups_param_names_mapping = dict(enumerate(PARAMS))
    pol = DynamicLrPolicy(T=3, b=0.4, ema=0.001, backbone_policy='sigmoid_kl', head_policy='normal', warmup_iters=W)

    class Runner:
        iter, epoch = 0, 0
    r = Runner()
    for it in range(15):
        lv = _losses(100 + it)
        r.outputs = {'log_vars': lv}
        r.iter = it
        hook.regular_lr = [hook.get_lr(r, b) for b in hook.base_lr]  # what the documented flow would have installed
        hook.after_train_iter(r)
        ref = hook.last_set
        got = dynamic_lr_after_train_iter(pol, lv, PARAMS, base, it, steps_at, 0.1, W, 1.0 / 3, warmup='linear', as_run=False)
        for i, n in enumerate(PARAMS):
            assert abs(got[i] - ref[i]) <= 1e-6 * abs(ref[i]), (it, n, got[i], ref[i])
