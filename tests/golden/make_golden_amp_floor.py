"""Generate tests/golden/amp_floor_<case>.json: how far the REFERENCE backbone module, run under ITS OWN mixed precision, lands
from ITS OWN fp32 result -- the `amp_floor` of the AMP parity tests, as `fp32_floor` is the floor of the fp32 ones.

    python tests/golden/make_golden_amp_floor.py [case ...]        (build container only: needs /root/reference)

What runs: /root/reference/mmrotate/models/backbones/convnext_moe.py, imported unmodified through oracle/ref_moe.py, with
the weights / inputs / gate noise / drop-path masks of the fp32 fixture `tests/golden/<case>.pt`, train-mode forward +
backward under ``torch.autocast`` -- what the reference's ``fp16 = dict(loss_scale='dynamic')`` configs do
(mmcv/mmcv/runner/fp16_utils.py:71-149: under torch >= 1.6 ``auto_fp16`` runs the decorated forward under
``torch.cuda.amp.autocast``; mmcv/mmcv/runner/hooks/optimizer.py:198-306: the loss is multiplied by the dynamic scale, 2^16
initially, halved while a gradient is non-finite, and the gradients are unscaled before the step).

There is no GPU here, so autocast runs on the CPU with dtype float16 -- and CPU autocast has a DIFFERENT cast policy from
the CUDA one the reference trains under (it only lowers convolutions / matmuls and lets everything else follow its input's
dtype, where CUDA autocast runs layer_norm, softmax, norm, exp, pow, softplus, sum, cumsum ... in float32).  The operators of
THAT list which the reference backbone executes are patched to the CUDA behaviour for the duration of the run
(`cuda_cast_policy`: layer_norm, softmax, norm, exp, softplus, pow, sum, cumsum, log, reciprocal, rsqrt, prod -- torch's
"fp32" and "fp32_set_opt_dtype" lists, aten/src/ATen/autocast_mode.cpp).  That emulation, not a CUDA run, is what this
floor is; the file says so in its header field.

Routing is teacher-forced to the fp32 reference's expert sets (gate values, thresholds and load terms still come from the
run's own fp16-path logits), exactly like the `forced` half of tests/test_fullsize_gpu.py's AMP cases, so the floor measures
arithmetic, not routing flips.  Stored per output / per gradient tensor: the distance to the fp32 fixture in the metrics the
test applies (element-wise quantiles and max-norm for outputs; max-norm and projection error for gradients).
"""
import contextlib
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_moe  # noqa: E402
from tests import fullsize_common as FC  # noqa: E402
from tests.golden.make_golden_fullsize import _FixedDrop  # noqa: E402

AMP_CASES = ['full_e8t2_b2', 'full_base_b1', 'full_base_b2']  # BASELINE configs #3 and #5 (batch 1 and 2)


@contextlib.contextmanager
def cuda_cast_policy():
    """inside CPU autocast: run the CUDA-autocast fp32-list operators the reference backbone uses in float32"""
    saved = []

    def up(t):
        return t.float() if torch.is_tensor(t) and t.dtype == torch.float16 else t

    def patch(owner, name, keep_out_fp32=True):
        orig = getattr(owner, name)

        def wrapped(*a, **kw):
            if not torch.is_autocast_enabled('cpu'):
                return orig(*a, **kw)
            return orig(*[up(x) for x in a], **{k: up(v) for k, v in kw.items()})
        saved.append((owner, name, orig))
        setattr(owner, name, wrapped)

    for owner, names in ((F, ['layer_norm', 'softmax', 'softplus', 'log_softmax']),
                         (torch, ['softmax', 'exp', 'pow', 'sum', 'cumsum', 'log', 'reciprocal', 'rsqrt', 'prod', 'norm']),
                         (torch.Tensor, ['softmax', 'exp', 'pow', '__pow__', 'sum', 'cumsum', 'log', 'reciprocal', 'rsqrt',
                                         'prod', 'norm'])):
        for n in names:
            patch(owner, n)
    try:
        yield
    finally:
        for owner, name, orig in saved:
            setattr(owner, name, orig)


class _ForcedRouting:
    """injects the fixture's gate noise (reference :203 `torch.randn_like`) and teacher-forces the top-(k+1) selection of
    every MoE block (:207 `logits.topk`) to the fp32 reference's expert sets: values gathered from THIS run's logits
    (differentiable, like topk's), the k forced experts ordered by them, then the best of the rest"""

    def __init__(self, noise, routing, E, k):
        self.noise, self.routing, self.E, self.k = iter(noise), iter(routing), E, k

    def __enter__(self):
        self._topk, self._randn_like = torch.Tensor.topk, torch.randn_like
        me = self

        def topk(t, kk, *a, **kw):
            if not (t.dim() == 2 and t.shape[1] == me.E and kk == min(me.k + 1, me.E)):
                return me._topk(t, kk, *a, **kw)
            forced = next(me.routing)['topk'].long()                       # (T, k) expert sets of the fp32 reference
            vin = t.detach().float().gather(1, forced)
            top = forced.gather(1, vin.argsort(1, descending=True))
            rest = t.detach().float().scatter(1, forced, float('-inf'))
            idx = torch.cat([top, rest.argmax(1, keepdim=True)], 1)[:, :kk]
            return t.gather(1, idx), idx
        torch.Tensor.topk = topk
        torch.randn_like = lambda t, *a, **kw: next(me.noise).to(t.dtype)
        return self

    def __exit__(self, *exc):
        torch.Tensor.topk, torch.randn_like = self._topk, self._randn_like
        return False


def make(case):
    fx = FC.load(case)
    cfg, seed = fx['cfg'], fx['seed']
    E, k = cfg['num_experts'], cfg['top_k']
    torch.manual_seed(0)
    net = ref_moe.build_reference_backbone(**cfg)
    net.load_state_dict(FC.seeded_state_dict(net.state_dict(), seed), strict=True)
    net.train()
    x, noise, drop = FC.make_inputs(case, noise_seed=fx['noise_seed'])
    for b, rs in zip([b for st in net.stages for b in st], drop):
        b.drop_path = _FixedDrop(rs)
    scale = 65536.0
    t0 = time.time()
    while True:
        net.zero_grad(set_to_none=True)
        with torch.autocast('cpu', dtype=torch.float16), cuda_cast_policy(), _ForcedRouting(noise, fx['routing'], E, k):
            outs, gl = net(x, ['single'])
            outs = [o.float() for o in outs]
            L = FC.loss_of(outs, gl.float(), seed)
        (L * scale).backward()
        if all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None):
            break
        scale /= 2.0  # Fp16OptimizerHook / GradScaler: skip the step, halve the scale
        assert scale >= 1.0
    dt = time.time() - t0
    rep = dict(case=case, what='reference module under torch.autocast(cpu, float16) with the CUDA cast policy patched in for '
               'layer_norm / softmax / norm / exp / softplus / pow / sum / cumsum / log / reciprocal / rsqrt / prod, routing '
               'teacher-forced to its fp32 run; distances to its own fp32 fixture', loss_scale=scale,
               torch_version=torch.__version__, seconds=round(dt, 1), out_dtypes=[str(o.dtype) for o in outs],
               gate_loss_rel=abs(float(gl) - fx['gate_loss']) / abs(fx['gate_loss']))
    for i, (o, ref) in enumerate(zip(outs, fx['outs'])):
        st = {}
        for name, e in FC.compare_output(i, o, ref).items():
            st[name] = dict(max=float(e.max()), p99=float(torch.quantile(e, 0.99)), p95=float(torch.quantile(e, 0.95)),
                            median=float(e.median()))
        got_s = FC.summarise_output(i, o)['samples'].double()
        st['max_norm_rel'] = float((got_s - ref['samples'].double()).abs().max() / ref['max_abs'])
        rep[f'out{i}'] = st
    grads = {kk: p.grad.float() / scale for kk, p in net.named_parameters() if p.grad is not None}
    table = fx['grads']['table']
    rep['grads'] = {}
    for key in table:
        mx, proj = FC.compare_grad_maxnorm(key, grads[key], fx['grads'])
        rep['grads'][key] = (mx, proj)
    worst = max(rep['grads'].items(), key=lambda kv: kv[1][0])
    temps = {kk: v for kk, v in rep['grads'].items() if kk.endswith('temperature')}
    print(f'{case}: {dt:.0f} s, loss scale {scale:g}; outputs max-norm {[round(rep[f"out{i}"]["max_norm_rel"], 5) for i in range(len(outs))]}; '
          f'worst gradient {worst}; temperature gradients {[(round(a, 4), round(b, 4)) for a, b in temps.values()]}', flush=True)
    path = os.path.join(FC.GOLDEN, f'amp_floor_{case}.json')
    with open(path, 'w') as f:
        json.dump(rep, f, indent=0)
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    assert ref_moe.available(), 'needs /root/reference'
    torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', '8')))
    for name in (sys.argv[1:] or AMP_CASES):
        make(name)
