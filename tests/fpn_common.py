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


def coder_round_trip_cases(n=3000, seed=3):
    """(gt, anchors, rois): oriented boxes in regular le90 form (w >= h, away from squares and from axis alignment, where
    the midpoint-offset representation is ambiguous by the reference's own 0.1 px vertex tolerance), horizontal anchors
    around their hull, perturbed oriented RoIs -- for the encode -> decode round-trip properties."""
    import math
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(n, 5, generator=g)
    gt[:, :2] = gt[:, :2] * 800 + 100
    gt[:, 2] = gt[:, 2] * 200 + 40
    gt[:, 3] = gt[:, 2] * (0.3 + 0.6 * torch.rand(n, generator=g))
    sign = torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0)
    gt[:, 4] = sign * (0.15 + torch.rand(n, generator=g) * (math.pi / 2 - 0.3))
    c, s = torch.cos(gt[:, 4]).abs(), torch.sin(gt[:, 4]).abs()
    xb, yb = gt[:, 2] / 2 * c + gt[:, 3] / 2 * s, gt[:, 2] / 2 * s + gt[:, 3] / 2 * c
    hull = torch.stack([gt[:, 0] - xb, gt[:, 1] - yb, gt[:, 0] + xb, gt[:, 1] + yb], 1)
    anchors = hull + torch.randn(n, 4, generator=g) * 8
    rois = gt.clone()
    rois[:, :2] += torch.randn(n, 2, generator=g) * 10
    rois[:, 2:4] *= 0.7 + 0.6 * torch.rand(n, 2, generator=g)
    rois[:, 4] = (torch.rand(n, generator=g) - 0.5) * math.pi * 0.98
    return gt, anchors, rois


def assert_boxes_close(a, b, tol_xywh, tol_angle):
    import math
    assert float((a[:, :4] - b[:, :4]).abs().max()) < tol_xywh
    d = (a[:, 4] - b[:, 4] + math.pi / 2) % math.pi - math.pi / 2
    assert float(d.abs().max()) < tol_angle
