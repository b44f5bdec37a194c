"""Optimizer step of the SM3Det training loop on MI355X.

``MultiTensorAdamW`` -- drop-in for ``torch.optim.AdamW`` as the reference configures it (AdamW, betas (0.9, 0.999),
weight_decay 0.05, ONE param group per parameter: mmcv/mmcv/runner/optimizer/default_constructor.py:180-227) fused
with the gradient clipping of ``OptimizerHook`` (``grad_clip=dict(max_norm=35, norm_type=2)``,
mmcv/mmcv/runner/hooks/optimizer.py:55-73).  All tensors are updated by ONE kernel launch that reads each group's
``lr`` / ``weight_decay`` from a device vector, so the per-parameter learning rates that the dynamic-lr hook writes
every step cost nothing; step counter, gradient norm and clip coefficient stay on the device (no host sync, hipGraph
capturable).

Differences from ``clip_grad_norm_`` + ``AdamW`` worth knowing: the clip coefficient is applied INSIDE the update, so
``p.grad`` keeps the unclipped (and, under a loss scale, still scaled) values after ``step()``; parameters without a
gradient are skipped like torch does (the tables are rebuilt when that set changes, outside captures only).

``DynamicLrPolicy`` -- the scalar arithmetic of ``DynamicLrUpdaterHook.get_dynamic_lr``
(mmrotate/core/hook/dynamic_lr.py:107-175; default ``head_policy='normal'``, ``backbone_policy`` min/avg/max) as a
plain object: feed it the step's loss scalars, get one lr multiplier per parameter name.
"""
import math

import torch

from . import _lib
from . import _lib_backbone as LB


class MultiTensorAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None,
                 loss_scale=None):
        """loss_scale: None (fp32 training) | 'dynamic' | float (static) | dict of torch GradScaler arguments
        (init_scale, growth_factor, backoff_factor, growth_interval) -- the `loss_scale` argument of the reference's
        Fp16OptimizerHook (mmcv/mmcv/runner/hooks/optimizer.py:198-243).  With a scale, train with
        ``opt.scale(loss).backward(); opt.step()``: unscaling, the overflow check, the skipped step and the scale update
        all happen on the device inside ``step()``."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
        self._built = False
        self._n_steps_host = 0  # steps taken (host-side count; only used to flag late-joining parameters)
        self._last_hyper = None
        self._scaler = None
        self._scaler_cfg = None
        if loss_scale is not None:
            cfg = dict(init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000)  # GradScaler()
            if isinstance(loss_scale, dict):
                cfg.update(loss_scale)
            elif isinstance(loss_scale, (int, float)):
                cfg.update(init_scale=float(loss_scale), growth_interval=0)  # static (hook re-sets it every iteration)
            elif loss_scale != 'dynamic':
                raise ValueError('loss_scale must be of type float, dict, or "dynamic", got ' + repr(loss_scale))
            self._scaler_cfg = cfg

    # ------------------------------------------------------------------------------------------- loss scaling
    def _ensure_scaler(self, device):
        if self._scaler is None and self._scaler_cfg is not None:
            self._scaler = torch.tensor([self._scaler_cfg['init_scale'], 0.0, 0.0], dtype=torch.float32, device=device)
        return self._scaler

    def scale(self, loss):
        """GradScaler.scale: loss * current scale (a device scalar; no sync)."""
        if self._scaler_cfg is None:
            return loss
        return loss * self._ensure_scaler(loss.device)[0]

    @property
    def loss_scale(self):
        """current scale as a python float (synchronises; logging / checkpoints only)"""
        return None if self._scaler is None else float(self._scaler[0])

    @property
    def found_inf(self):
        """1.0 if the last step() was skipped because a gradient overflowed (device tensor view)"""
        return None if self._scaler is None else self._scaler[2]

    def scaler_state_dict(self):
        if self._scaler is None:
            return dict(self._scaler_cfg or {})
        sc, tr, _ = self._scaler.tolist()
        return dict(self._scaler_cfg, scale=sc, _growth_tracker=int(tr))

    def load_scaler_state_dict(self, sd):
        if self._scaler_cfg is None:
            raise _lib.SM3Error('this optimizer was built without loss_scale')
        self._scaler_cfg.update({k: sd[k] for k in ('growth_factor', 'backoff_factor', 'growth_interval') if k in sd})
        if self._scaler is not None and 'scale' in sd:
            self._scaler.copy_(torch.tensor([sd['scale'], float(sd.get('_growth_tracker', 0)), 0.0]))
        elif 'scale' in sd:
            self._scaler_cfg['init_scale'] = sd['scale']

    # ------------------------------------------------------------------------- reference optimizer state
    def load_reference_state(self, module, ref_state_dict, ref_param_keys=None):
        """`resume_from` of a reference checkpoint: translate the state of the reference's ``torch.optim.AdamW`` (one
        group per parameter in ``named_parameters()`` order -- mmcv ``DefaultOptimizerConstructor.add_params``,
        optimizer/default_constructor.py:180-227; moments in the REFERENCE layouts: one tensor per expert Linear,
        depthwise weights (C,1,7,7), 3x3 conv weights OIHW) into this optimizer's per-parameter state.

        The moments have the shapes of the parameters they belong to, so they are pushed through the module's own
        ``load_state_dict`` hooks (the ones that turn a reference ``state_dict`` into the fused / tap-major storage) on a
        CPU shadow copy of ``module``.  ``ref_param_keys``: the reference's parameter names in optimizer order; default =
        the parameter keys of ``module.state_dict()``, which are emitted in the reference's order.  The step counter is
        one device scalar here: the reference's per-parameter ``step`` values must agree (they do: AdamW steps every
        parameter that has a gradient, and the schedule needs one value)."""
        import copy
        buf = {n for n, _ in module.named_buffers()}
        if ref_param_keys is None:
            ref_param_keys = [k for k in module.state_dict().keys() if k not in buf]
        state = ref_state_dict['state']
        order = [i for g in ref_state_dict['param_groups'] for i in g['params']]
        if len(order) != len(ref_param_keys):
            raise ValueError(f'reference optimizer state has {len(order)} parameters, the model has '
                             f'{len(ref_param_keys)} under the reference key schema')
        shadow = copy.deepcopy(module).to('cpu')
        by_name = dict(module.named_parameters())
        mine = {id(p) for g in self.param_groups for p in g['params']}
        steps = set()
        for moment in ('exp_avg', 'exp_avg_sq'):
            sd = {}
            for key, idx in zip(ref_param_keys, order):
                st = state.get(idx, state.get(str(idx)))
                if st is None:
                    continue  # never stepped (frozen / unused): stays at its zero-initialised state
                sd[key] = st[moment].detach().to('cpu', torch.float32)
                steps.add(float(st['step']))
            with torch.no_grad():
                for q in shadow.parameters():
                    q.fill_(float('nan'))  # marks what the load below does not touch
            res = shadow.load_state_dict(sd, strict=False)
            if res.unexpected_keys:  # e.g. only some experts of a fused tensor: the hooks fuse all or nothing
                raise ValueError(f'reference optimizer state names unknown parameters: {res.unexpected_keys[:4]}')
            for name, q in shadow.named_parameters():
                p = by_name[name]
                if id(p) not in mine or bool(torch.isnan(q).any()):
                    continue
                self.state[p][moment] = q.detach().clone().to(p.device).contiguous()
        if len(steps) > 1:
            raise ValueError(f'reference optimizer state carries different step counts {sorted(steps)[:4]}')
        if steps:
            dev = next(iter(by_name.values())).device
            self._step = torch.full((1,), steps.pop(), dtype=torch.float32, device=dev)
        self._built = False  # pointer tables are rebuilt on the next step()

    # ------------------------------------------------------------------------------------------- tables
    def _grad_set(self):
        return frozenset(id(p) for g in self.param_groups for p in g['params'] if p.grad is not None)

    def _build(self):
        ps = []
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise _lib.SM3Error('MultiTensorAdamW: parameters must be contiguous float32 GPU tensors')
                ps.append((gi, p))
        if not ps:
            raise _lib.SM3Error('MultiTensorAdamW: no parameter has a gradient')
        dev = ps[0][1].device
        self._params = ps
        chunk = _lib.lib().sm3_optim_chunk_elems()
        tab = []
        for ti, (_, p) in enumerate(ps):
            st = self.state[p]
            if 'exp_avg' not in st:
                if hasattr(self, '_step') and self._n_steps_host > 0:
                    # ONE step counter serves every tensor (one bias correction per launch).  torch.optim.AdamW -- what
                    # the reference uses -- keeps a step per parameter, so a parameter that receives its first gradient
                    # after others have already stepped would start its bias correction at 1 there and at the global step
                    # here.  No SM3Det config does that (all trainable parameters get a gradient every step, DDP demands
                    # it); say so instead of diverging silently.
                    import warnings
                    warnings.warn('MultiTensorAdamW: a parameter received its first gradient after the optimizer had '
                                  'already stepped; its Adam bias correction uses the global step (torch.optim.AdamW would '
                                  'restart it at 1)')
                st['exp_avg'] = torch.zeros_like(p)
                st['exp_avg_sq'] = torch.zeros_like(p)
            for c in range((p.numel() + chunk - 1) // chunk):
                tab.append((ti, c))
        as_u64 = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)  # noqa: E731
        self._p_ptrs = as_u64([p for _, p in ps])
        self._m_ptrs = as_u64([self.state[p]['exp_avg'] for _, p in ps])
        self._v_ptrs = as_u64([self.state[p]['exp_avg_sq'] for _, p in ps])
        self._g_ptrs = as_u64([p.grad for _, p in ps])
        self._g_addr = [p.grad.data_ptr() for _, p in ps]
        # fp16 shadows of AMP operand weights (backbone_ops._shadow hangs them on the parameters): the update keeps them
        # current, so the forward never casts.  0 = the parameter has none.
        self._h_addr = self._shadow_addrs()
        self._h_ptrs = torch.tensor(self._h_addr, dtype=torch.int64, device=dev)
        self._numel = torch.tensor([p.numel() for _, p in ps], dtype=torch.int64, device=dev)
        self._chunks = torch.tensor(tab, dtype=torch.int32, device=dev).contiguous()
        self._n_chunks = len(tab)
        self._total_numel = sum(p.numel() for _, p in ps)
        self._lr = torch.empty(len(ps), dtype=torch.float32, device=dev)
        self._wd = torch.empty(len(ps), dtype=torch.float32, device=dev)
        if not hasattr(self, '_step'):  # a rebuild keeps the step counter
            self._step = torch.zeros(1, dtype=torch.float32, device=dev)
        self._coef = torch.ones(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._partials = torch.empty(self._n_chunks, dtype=torch.float32, device=dev)
        self._built = True
        self._built_set = self._grad_set()
        # persistent pinned staging (a non_blocking H2D copy out of a temporary pageable tensor may read freed memory), TWO
        # slots used alternately, each guarded by its own event: a per-iteration lr schedule (warm-up, DynamicLrUpdaterHook)
        # refills the slot whose copy was issued two updates ago -- long done -- instead of draining the stream every step
        self._hyper_pin = torch.empty(2, 2, len(ps), dtype=torch.float32, pin_memory=True)
        self._hyper_ev = [None, None]
        self._hyper_slot = 0
        self.update_hyperparams(force=True)

    def _shadow_addrs(self):
        out = []
        for _, p in self._params:
            s = getattr(p, '_sm3_shadow', None)
            # only a shadow that is current (cast from this value of p) may be maintained incrementally
            ok = s is not None and getattr(p, '_sm3_shadow_version', None) == p._version and s.is_contiguous()
            out.append(s.data_ptr() if ok else 0)
        return out

    def refresh_grad_pointers(self):
        """Re-read the addresses of ``p.grad`` (they move when gradients are dropped with set_to_none and re-created
        by autograd).  ``step()`` does it itself when not capturing; call it once, outside the capture, before capturing
        a graph that contains ``step()`` so the graph reads the gradients where the captured backward writes them."""
        if not self._built:
            return
        if self._grad_set() != self._built_set:
            # torch.optim.AdamW skips parameters without a gradient and picks them up when one appears: rebuild the
            # tables for the current set (optimizer state is kept per parameter; never inside a capture)
            if torch.cuda.is_current_stream_capturing():
                raise _lib.SM3Error('MultiTensorAdamW: the set of parameters with gradients changed inside a capture')
            self._build()
            return
        addr = [p.grad.data_ptr() for _, p in self._params]
        if addr != self._g_addr:
            self._g_ptrs.copy_(torch.tensor(addr, dtype=torch.int64))
            self._g_addr = addr
        haddr = self._shadow_addrs()  # shadows appear with the first AMP forward, and go stale when a torch op writes p
        if haddr != self._h_addr:
            self._h_ptrs.copy_(torch.tensor(haddr, dtype=torch.int64))
            self._h_addr = haddr

    def update_hyperparams(self, force=False):
        """Push the groups' lr / weight_decay to the device vectors (call before replaying a captured graph whenever a
        scheduler changed them; ``step()`` does it itself when not capturing)."""
        if not self._built:
            return
        lr = [self.param_groups[gi]['lr'] for gi, _ in self._params]
        wd = [self.param_groups[gi]['weight_decay'] for gi, _ in self._params]
        if getattr(self, '_dla_owns_lr', False) and not force:
            return  # DeviceDynamicLr writes the lr vector on the device every iteration
        if force or (lr, wd) != self._last_hyper:
            slot = self._hyper_slot
            self._hyper_slot ^= 1
            if self._hyper_ev[slot] is not None:
                self._hyper_ev[slot].synchronize()  # the copy that last read this slot (two updates ago): no stream drain
            pin = self._hyper_pin[slot]
            pin[0].copy_(torch.tensor(lr, dtype=torch.float32))
            pin[1].copy_(torch.tensor(wd, dtype=torch.float32))
            self._lr.copy_(pin[0], non_blocking=True)
            self._wd.copy_(pin[1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._hyper_ev[slot] = ev
            self._last_hyper = (lr, wd)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._built:
            self._build()
        elif not torch.cuda.is_current_stream_capturing():
            self.refresh_grad_pointers()
            self.update_hyperparams()
        g0 = self.param_groups[0]
        b1, b2 = g0['betas']
        self._n_steps_host += 1
        sc = self._ensure_scaler(self._step.device)
        cfg = self._scaler_cfg or dict(growth_factor=2.0, backoff_factor=0.5, growth_interval=0)
        LB.call('adamw_multi', self._p_ptrs, self._g_ptrs, self._m_ptrs, self._v_ptrs, self._h_ptrs, self._numel, self._chunks,
                self._n_chunks, self._lr, self._wd, float(b1), float(b2), float(g0['eps']), float(self.max_grad_norm),
                self._step, self._coef, self.grad_norm, self._partials, sc, float(cfg['growth_factor']),
                float(cfg['backoff_factor']), int(cfg['growth_interval']),
                nbytes=(28.0 + (4.0 if (self.max_grad_norm > 0 or sc is not None) else 0.0)) * self._total_numel)
        return loss


class _EMA:
    """EMA_meter of dynamic_lr.py:27-43"""

    def __init__(self, alpha):
        self.alpha, self.ema, self.steps = alpha, None, 0

    def update(self, v):
        self.ema = v if self.ema is None else self.alpha * v + (1 - self.alpha) * self.ema
        self.steps += 1

    def get(self):
        return self.ema if self.ema is not None else 1e-3


DEFAULT_REWEIGHT = {  # dynamic_lr.py:64-67
    'sar_loss_cls': 'sar_bbox_head', 'sar_loss_bbox': 'sar_bbox_head', 'sar_loss_dfl': 'sar_bbox_head',
    'rgb_loss_rpn_cls': 'rgb_rpn_head', 'rgb_loss_rpn_bbox': 'rgb_rpn_head', 'rgb_loss_cls': 'rgb_roi_head',
    'rgb_loss_bbox': 'rgb_roi_head', 'ifr_loss_rpn_cls': 'ifr_rpn_head', 'ifr_loss_rpn_bbox': 'ifr_rpn_head',
    'ifr_loss_cls': 'ifr_roi_head', 'ifr_loss_bbox': 'ifr_roi_head'}


class DynamicLrPolicy:
    """Per-parameter lr multipliers from the step's loss scalars (dynamic_lr.py:107-175)."""

    def __init__(self, T=5, b=0.5, ema=0.005, backbone_policy='min', head_policy='normal', warmup_iters=0,
                 reweight_losses=None):
        self.T, self.b = T, b
        self.backbone_policy, self.head_policy = backbone_policy, head_policy
        self.warmup_iters = warmup_iters
        self.reweight_losses = dict(reweight_losses or DEFAULT_REWEIGHT)
        self.history = [_EMA(ema) for _ in self.reweight_losses]

    def multipliers(self, log_vars, param_names):
        """log_vars: {loss name: float}; returns {param name: multiplier}."""
        names = [k for k in log_vars if k in self.reweight_losses]
        cur = [float(sum(log_vars[k]) if isinstance(log_vars[k], list) else log_vars[k]) for k in names]
        n = len(cur)
        if self.history[0].steps < self.warmup_iters or self.head_policy == 'None':
            bw = [1.0] * n
        else:
            hist = [m.get() for m in self.history[:n]]
            w = [(c / h) if self.head_policy == 'reverse' else (h / c) for c, h in zip(cur, hist)]
            mx = max(x / self.T for x in w)
            ex = [math.exp(x / self.T - mx) for x in w]
            bw = [n * e / sum(ex) for e in ex]
        subnet = {}
        for s in set(self.reweight_losses.values()):
            ws = [bw[i] for i, k in enumerate(names) if self.reweight_losses[k] == s]
            subnet[s] = sum(ws) / len(ws) if ws else 1.0
        vals = list(subnet.values())
        if self.backbone_policy in ('kl', 'sigmoid_kl'):
            # F.kl_div(softmax(cur).log(), softmax(history), reduction='batchmean') of 1-D tensors = sum / n (:156-165)
            hist = [m.get() for m in self.history[:n]]
            lh, lc = _log_softmax(hist), _log_softmax(cur)
            kl = sum(math.exp(a) * (a - c) for a, c in zip(lh, lc)) / n
            shared = (1 + (1 - kl) / math.sqrt(self.T)) if self.backbone_policy == 'kl' else \
                2.0 / (1.0 + math.exp(-((1 - kl - self.b) * self.T)))
        else:
            shared = {'min': min(vals), 'avg': sum(vals) / len(vals), 'max': max(vals)}.get(self.backbone_policy, 1.0)
        for i, c in enumerate(cur):
            self.history[i].update(c)
        out = {}
        for pn in param_names:
            mult = shared
            for s, v in subnet.items():
                if s in pn:
                    mult = v
                    break
            out[pn] = mult
        return out


def dynamic_lr_after_train_iter(policy, log_vars, param_names, base_lrs, it, step, gamma=0.1, warmup_iters=0,
                                warmup_ratio=0.1, warmup='linear', as_run=True):
    """Host form of ``DynamicLrUpdaterHook.after_train_iter`` (dynamic_lr.py:192-217) for iteration index `it`.  `policy` is
    a ``DynamicLrPolicy`` (it carries the EMAs; its ``warmup_iters`` is the hook's, which gates the head weights whether or
    not a warm-up is configured).  -> list of lrs, one per name.

    Warm-up (``warmup='linear'``, ``it < warmup_iters``): the loss EMAs update and

    * ``as_run=True`` (default): the lr is left alone, i.e. every group keeps its INITIAL lr.  This is what the reference
      does when run: the hook overrides ``before_train_iter`` with ``pass`` (:189-190), mmcv's ``before_train_epoch`` returns
      before it sets ``regular_lr`` when ``by_epoch=False`` (lr_updater.py:134-135; IterBasedRunner forces it and calls
      ``before_epoch`` anyway), so ``regular_lr`` stays ``[]``, ``get_warmup_lr`` returns ``[]`` and ``_set_lr`` writes
      nothing -- the first `warmup_iters` iterations train at the full base lr.
    * ``as_run=False``: the ramp mmcv documents, ``regular_lr * (1 - (1 - it / warmup_iters) * (1 - warmup_ratio))``
      (lr_updater.py:75-92) -- what the config's author presumably intended; kept as an option.

    Pinned on the reference hook over the reference's own ``LrUpdaterHook`` driven the way IterBasedRunner drives it by
    tests/test_dla_cpu.py; the device kernel (``DeviceDynamicLr``) is compared with this function on the GPU."""
    e = (it // step) if isinstance(step, int) else next((i for i, s_ in enumerate(step) if it < s_), len(step))
    regular = [b * gamma ** e for b in base_lrs]
    if warmup is not None and warmup_iters and it < warmup_iters:
        names = [k for k in log_vars if k in policy.reweight_losses]
        for i, k in enumerate(names):
            v = log_vars[k]
            policy.history[i].update(float(sum(v) if isinstance(v, list) else v))
        if as_run:
            return list(base_lrs)
        k_ = (1 - it / warmup_iters) * (1 - warmup_ratio)
        return [r * (1 - k_) for r in regular]
    mult = policy.multipliers(log_vars, param_names)
    return [r * mult[n] for r, n in zip(regular, param_names)]


def _log_softmax(v):
    mx = max(v)
    ls = math.log(sum(math.exp(x - mx) for x in v))
    return [x - mx - ls for x in v]


_HEAD_POLICY = {'normal': 0, 'reverse': 1, 'None': 2}
_BACKBONE_POLICY = {'min': 0, 'avg': 1, 'max': 2, 'kl': 3, 'sigmoid_kl': 4}


class DeviceDynamicLr:
    """``DynamicLrUpdaterHook`` (mmrotate/core/hook/dynamic_lr.py:45-217; ``lr_config = dict(policy='dynamic', ...)`` of
    local_configs/main_SM3Det.py:291-300) as one kernel launch per iteration on device-resident loss scalars
    (``sm3_dla_lr``): the reference reads its 11 losses with ``.item()`` every iteration; here the loss EMAs, the update and
    iteration counters and the per-tensor lr vector never leave the device, so the step stays capturable in a hipGraph.

    ``update(losses)`` is the hook's ``after_train_iter`` -- it runs BEFORE the optimizer step of the same iteration (the
    lr hook has priority VERY_HIGH, the optimizer hook ABOVE_NORMAL), so the lr it writes is the one that iteration's
    AdamW update uses.  ``losses``: {name: 0-d device tensor (or list of them: summed)} -- the step's log_vars; only the
    keys of ``reweight_losses`` count, in the dict's order (as the reference walks ``log_vars``)."""

    def __init__(self, optimizer, param_names, step, gamma=0.1, min_lr=None, extra_args=None, reweight_losses=None,
                 warmup=None, warmup_iters=0, warmup_ratio=0.1, by_epoch=False, warmup_as_run=True, **_unused):
        if by_epoch:
            raise NotImplementedError('the dynamic policy asserts by_epoch=False (dynamic_lr.py:217)')
        if warmup not in (None, 'linear'):
            raise NotImplementedError(f"warmup={warmup!r}: the SM3Det configs use 'linear'")
        ea = dict(T=5, b=0.5, ema=0.005, backbone_policy='min', head_policy='normal')
        ea.update(extra_args or {})
        self.extra, self.reweight = ea, dict(reweight_losses or DEFAULT_REWEIGHT)
        self.step_at = [step] if isinstance(step, int) else list(step)
        self.step_is_int = isinstance(step, int)
        self.gamma, self.min_lr = gamma, min_lr
        if min_lr is not None:
            raise NotImplementedError('min_lr clipping is per-tensor host arithmetic in the reference; no SM3Det config sets it')
        # warmup_iters is kept even without a warm-up: the hook's `history.steps < warmup_iters` gate on the head weights
        # (dynamic_lr.py:124) reads it either way.  warmup_as_run (default): during the warm-up the lr stays the initial lr,
        # as in a reference run (see dynamic_lr_after_train_iter) -- passed to the kernel as warm-up ratio 1; False: mmcv's
        # documented linear ramp from `warmup_ratio`.
        self.warmup = warmup
        self.warmup_iters = int(warmup_iters or 0)
        self.warmup_as_run = bool(warmup_as_run)
        self.warmup_ratio = 1.0 if (warmup and self.warmup_as_run) else float(warmup_ratio)
        self.opt = optimizer
        if not optimizer._built:
            optimizer._build()
        names = list(param_names)
        assert len(names) == len(optimizer._params), 'one name per optimizer tensor, in the optimizer\'s order'
        self.subnets = sorted(set(self.reweight.values()))
        dev = optimizer._lr.device
        ps = []
        for n in names:  # dynamic_lr.py:166-174: the first sub-network whose name occurs in the parameter's name, else shared
            ps.append(next((i for i, sname in enumerate(self.subnets) if sname in n), -1))
        self._param_subnet = torch.tensor(ps, dtype=torch.int32, device=dev)
        self._base_lr = torch.tensor([optimizer.param_groups[gi].get('initial_lr', optimizer.param_groups[gi]['lr'])
                                      for gi, _ in optimizer._params], dtype=torch.float32, device=dev)
        self._sched = torch.ones(1, dtype=torch.float32, device=dev)
        self._sched_host = 1.0
        self._state = None
        self._keys = None
        self.iter = 0

    def _decay(self, it):
        if self.step_is_int:
            e = it // self.step_at[0]
        else:
            e = next((i for i, s_ in enumerate(self.step_at) if it < s_), len(self.step_at))
        return self.gamma ** e

    def set_iter(self, it):
        """host bookkeeping of get_lr's step decay (:92-105): call with the runner's iteration index when not replaying a
        fixed graph; the factor changes at the few `step` milestones only (a tiny H2D copy then)"""
        f = self._decay(it)
        if f != self._sched_host:
            self._sched.fill_(f)
            self._sched_host = f

    def update(self, losses):
        keys = [k for k in losses if k in self.reweight]
        if self._keys is None:
            self._keys = keys
            dev = self._base_lr.device
            self._loss_subnet = torch.tensor([self.subnets.index(self.reweight[k]) for k in keys], dtype=torch.int32, device=dev)
            self._state = torch.zeros(len(keys) + 2, dtype=torch.float64, device=dev)
        elif keys != self._keys:
            raise _lib.SM3Error('DeviceDynamicLr: the set / order of reweighted losses changed between iterations '
                                f'({self._keys} -> {keys}); the device tables are built for one set')
        vals = [sum(losses[k]) if isinstance(losses[k], (list, tuple)) else losses[k] for k in keys]
        cur = torch.stack([v.detach().float().reshape(()) for v in vals])
        ea = self.extra
        LB.call('dla_lr', cur, len(keys), self._loss_subnet, len(self.subnets), self._param_subnet, self._base_lr,
                self._base_lr.numel(), self._sched, self._state, _HEAD_POLICY[str(ea['head_policy'])],
                _BACKBONE_POLICY.get(ea['backbone_policy'], 5), self.warmup_iters if self.warmup else -self.warmup_iters,
                self.warmup_ratio, float(ea['T']),
                float(ea['b']), float(ea['ema']), self.opt._lr)
        self.opt._dla_owns_lr = True  # the optimizer must not overwrite the device lr vector from its host-side groups

    def fast_forward(self, iters):
        """set the update / iteration counters (e.g. past the warm-up, to time or test the steady-state branch); the EMAs keep
        the values they have"""
        if self._state is None:
            raise _lib.SM3Error('DeviceDynamicLr.fast_forward: call update() once first (the state is sized by the loss set)')
        self._state[-2:] = float(iters)

    def state_host(self):
        """(EMAs, updates, iteration) read back -- tests / checkpointing only"""
        s_ = self._state.cpu().tolist()
        return s_[:-2], int(s_[-2]), int(s_[-1])


# ------------------------------------------------------------------------------------------------------------------------
# mmcv DefaultOptimizerConstructor (mmcv/mmcv/runner/optimizer/default_constructor.py:95-260) on this package's modules
def _reference_keys(module):
    """the parameter names of `module` under the REFERENCE key schema, in the reference's `named_parameters()` order
    (``state_dict()`` is emitted in that schema and order by the modules' hooks; buffers dropped)"""
    buffers = {n for n, _ in module.named_buffers()}
    return [k for k in module.state_dict().keys() if k not in buffers]


def _module_kind(module, key):
    """'norm' | 'dwconv' | 'other' of the module that owns reference parameter `key` -- what
    `DefaultOptimizerConstructor.add_params` tests with isinstance (:167-172): norm layers (BatchNorm / InstanceNorm /
    GroupNorm / LayerNorm and subclasses) and depth-wise convolutions (Conv2d with in_channels == groups).  A key whose
    module path does not exist here belongs to a tensor this package stores fused (expert Linear layers): 'other'."""
    import torch.nn as nn
    path = key.rsplit('.', 1)[0] if '.' in key else ''
    try:
        m = module.get_submodule(path) if path else module
    except AttributeError:
        return 'other'
    if isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm, nn.GroupNorm, nn.LayerNorm)):
        return 'norm'
    if type(m).__name__ == 'DepthwiseConv7x7' or (isinstance(m, nn.Conv2d) and m.in_channels == m.groups):
        return 'dwconv'
    return 'other'


def reference_param_options(module, optimizer_cfg, prefix=''):
    """{reference key: dict(lr=, weight_decay=)} for every parameter of `module` (mounted at `prefix` in the detector,
    e.g. 'backbone'), by the rules of `DefaultOptimizerConstructor.add_params` (:140-238): a parameter whose dotted name
    contains a `custom_keys` entry takes that entry's `lr_mult` / `decay_mult` (longest key first, then alphabetical) and
    nothing else; otherwise `bias_lr_mult` for biases outside norm layers, and `norm_decay_mult` / `dwconv_decay_mult` /
    `bias_decay_mult` in that precedence.  (`dcn_offset_lr_mult`: no DCN layer exists in the SM3Det models.)"""
    cfg = dict(optimizer_cfg)
    pw = dict(cfg.get('paramwise_cfg') or {})
    base_lr, base_wd = cfg.get('lr'), cfg.get('weight_decay')
    custom = pw.get('custom_keys', {})
    if not isinstance(custom, dict):
        raise TypeError(f'If specified, custom_keys must be a dict, but got {type(custom)}')
    if base_wd is None and (any('decay_mult' in v for v in custom.values()) or
                            any(k in pw for k in ('bias_decay_mult', 'norm_decay_mult', 'dwconv_decay_mult'))):
        raise ValueError('base_wd should not be None')
    sorted_keys = sorted(sorted(custom.keys()), key=len, reverse=True)
    bias_lr_mult, bias_decay_mult = pw.get('bias_lr_mult', 1.), pw.get('bias_decay_mult', 1.)
    norm_decay_mult, dwconv_decay_mult = pw.get('norm_decay_mult', 1.), pw.get('dwconv_decay_mult', 1.)
    out = {}
    for key in _reference_keys(module):
        full = f'{prefix}.{key}' if prefix else (key if '.' in key else f'.{key}')  # f'{prefix}.{name}' of add_params
        leaf = key.rsplit('.', 1)[-1]
        kind = _module_kind(module, key)
        lr, wd = base_lr, base_wd
        for ck in sorted_keys:
            if ck in full:
                lr = base_lr * custom[ck].get('lr_mult', 1.)
                if base_wd is not None:
                    wd = base_wd * custom[ck].get('decay_mult', 1.)
                break
        else:
            if pw:
                if leaf == 'bias' and kind != 'norm':
                    lr = base_lr * bias_lr_mult
                if base_wd is not None:
                    if kind == 'norm':
                        wd = base_wd * norm_decay_mult
                    elif kind == 'dwconv':
                        wd = base_wd * dwconv_decay_mult
                    elif leaf == 'bias':
                        wd = base_wd * bias_decay_mult
        out[key] = dict(lr=lr, weight_decay=wd)
    return out


def param_groups_from_cfg(modules, optimizer_cfg):
    """One param group per trainable parameter of `modules` ({detector attribute name: module}, e.g.
    {'backbone': net, 'neck': fpn}) with the `lr` / `weight_decay` the reference's constructor would give it.  The options
    are computed per REFERENCE parameter and carried onto this package's storage by loading them through the modules' own
    `load_state_dict` hooks on a CPU shadow (the hooks that fuse the expert Linears and re-lay the convolutions): a fused
    tensor whose reference parameters disagree (e.g. a custom key that names one expert) is refused."""
    import copy
    groups = []
    for prefix, module in modules.items():
        opts = reference_param_options(module, optimizer_cfg, prefix)
        shadow = copy.deepcopy(module).to('cpu')
        ref_sd = {k: v for k, v in shadow.state_dict().items() if k in opts}
        per_field = {}
        for field in ('lr', 'weight_decay'):
            vals = sorted({o[field] for o in opts.values() if o[field] is not None})
            code = {v: float(i + 1) for i, v in enumerate(vals)}  # exact small integers survive any layout hook
            with torch.no_grad():
                for q in shadow.parameters():
                    q.fill_(float('nan'))
            sd = {k: torch.full_like(v, code[opts[k][field]] if opts[k][field] is not None else 0.0, dtype=torch.float32)
                  for k, v in ref_sd.items()}
            res = shadow.load_state_dict(sd, strict=False)
            if res.unexpected_keys:
                raise ValueError(f'unknown reference parameters {res.unexpected_keys[:4]}')
            back = {c: v for v, c in code.items()}
            back[0.0] = None
            per_field[field] = {}
            for name, q in shadow.named_parameters():
                if bool(torch.isnan(q.detach()).any()):
                    # still holds the NaN fill: no reference key reached it through load_state_dict (a parameter this
                    # package adds, or a key the hooks do not map) -- say so instead of reporting mixed options
                    raise KeyError(f'{prefix}.{name}: no reference parameter covers it (state_dict schema mismatch)')
                lo, hi = float(q.detach().min()), float(q.detach().max())
                if lo != hi or lo not in back:
                    raise NotImplementedError(f'{prefix}.{name}: its reference parameters carry different `{field}` options; '
                                              'this package stores them in one tensor')
                per_field[field][name] = back[lo]
        for name, p in module.named_parameters():
            if not p.requires_grad:
                continue
            g = dict(params=[p], name=f'{prefix}.{name}' if prefix else name)
            if per_field['lr'][name] is not None:
                g['lr'] = per_field['lr'][name]
            if per_field['weight_decay'][name] is not None:
                g['weight_decay'] = per_field['weight_decay'][name]
            groups.append(g)
    return groups


def build_optimizer(modules, optimizer_cfg, optimizer_config=None, loss_scale=None):
    """`build_optimizer(model, cfg.optimizer)` + `OptimizerHook(**cfg.optimizer_config)` of the reference's training
    script (mmrotate/apis/train.py) for `type='AdamW'`: `MultiTensorAdamW` over `param_groups_from_cfg`, `grad_clip`
    from `optimizer_config`, `loss_scale` as `cfg.fp16['loss_scale']`."""
    cfg = dict(optimizer_cfg)
    if cfg.get('type') != 'AdamW':
        raise NotImplementedError(f"optimizer type {cfg.get('type')!r}: the SM3Det configs use AdamW")
    clip = (optimizer_config or {}).get('grad_clip') or {}
    if clip and clip.get('norm_type', 2) != 2:
        raise NotImplementedError('grad_clip norm_type != 2')
    return MultiTensorAdamW(param_groups_from_cfg(modules, cfg), lr=cfg['lr'], betas=tuple(cfg.get('betas', (0.9, 0.999))),
                            eps=cfg.get('eps', 1e-8), weight_decay=cfg.get('weight_decay', 1e-2),
                            max_grad_norm=clip.get('max_norm'), loss_scale=loss_scale)
