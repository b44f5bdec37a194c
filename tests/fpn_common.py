"""Shared helpers of the FPN tests: fixture loading and the seeded tensors of tests/golden/make_golden_fpn.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden_fpn as MG  # noqa: E402

CASES = list(MG.CASES)


def load(name):
    return torch.load(os.path.join(ROOT, 'tests', 'golden', name + '.pt'), weights_only=False)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def check_run(fx, sl, outs, grads, dinputs, tol_fwd, tol_bwd):
    """outs: tuple of NCHW tensors; grads: {reference key: tensor in the reference shape}; dinputs: list."""
    run = fx['runs'][sl]
    assert len(outs) == len(run['outs'])
    for o, r in zip(outs, run['outs']):
        assert tuple(o.shape) == tuple(r.shape)
        assert rel_err(o, r) < tol_fwd, rel_err(o, r)
    for k, r in run['grads'].items():
        assert k in grads, f'no gradient for {k}'
        assert rel_err(MG.sample(grads[k].detach().cpu().contiguous()), r) < tol_bwd, (k, rel_err(MG.sample(grads[k].detach().cpu().contiguous()), r))
    for d, r in zip(dinputs, run['dinputs']):
        if r is None:
            assert d is None or float(d.abs().max()) == 0.0
        else:
            assert rel_err(d, r) < tol_bwd
