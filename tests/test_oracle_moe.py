"""CPU tier: pins oracle/moe_oracle.py (torch-CPU restatement of the reference backbone) against the committed
fixtures generated from the REFERENCE module (tests/golden/make_golden_moe.py), and -- when /root/reference is
present -- against the live reference module.  Tolerances: forward 1e-5 rel, gradients 1e-4 rel (both are fp32
CPU computations that differ only in summation order)."""
import pytest
import torch

from oracle import moe_oracle as MO
from oracle import ref_moe
from tests.moe_common import load_fixture, loss_of, oracle_kwargs, rel_err


@pytest.mark.parametrize('name', ['moe_e4k2', 'moe_e8k3', 'moe_lin_e4k2'])
def test_oracle_eval_matches_reference_fixture(name):
    fx = load_fixture(name)
    with torch.no_grad():
        outs, gl = MO.backbone_forward(fx['x'], fx['state_dict'], train=False, **oracle_kwargs(fx['cfg']))
    for o, r in zip(outs, fx['eval']['outs']):
        assert o.shape == r.shape and rel_err(o, r) < 1e-5
    assert rel_err(gl, fx['eval']['gate_loss']) < 1e-5


@pytest.mark.parametrize('name', ['moe_e4k2', 'moe_e8k3', 'moe_lin_e4k2'])
def test_oracle_train_fwd_bwd_matches_reference_fixture(name):
    fx = load_fixture(name)
    p = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('.mean', '.std')))
         for k, v in fx['state_dict'].items()}
    outs, gl = MO.backbone_forward(fx['x'], p, train=True, noise=fx['noise'], drop_scale=fx['drop_scale'],
                                   **oracle_kwargs(fx['cfg']))
    for o, r in zip(outs, fx['train']['outs']):
        assert rel_err(o, r) < 1e-5
    assert rel_err(gl, fx['train']['gate_loss']) < 1e-5
    loss_of(outs, gl).backward()
    checked = 0
    for k, g in fx['train']['grads'].items():
        assert p[k].grad is not None, k
        assert rel_err(p[k].grad, g) < 1e-4, (k, rel_err(p[k].grad, g))
        checked += 1
    assert checked > 30


@pytest.mark.skipif(not ref_moe.available(), reason='/root/reference not present (GPU box)')
def test_oracle_vs_live_reference_tiny_arch():
    torch.manual_seed(3)
    kw = dict(arch='tiny', MoE_Block_inds=[[], [0, 2], [0, 2, 4, 6, 8], [0, 2]], num_experts=8, top_k=2)
    net = ref_moe.build_reference_backbone(**kw)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith('gamma'):
                p.fill_(1.0)
    net.eval()
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        outs, gl = net(x, ['single'])
        o2, g2 = MO.backbone_forward(x, net.state_dict(), arch='tiny', moe_block_inds=kw['MoE_Block_inds'],
                                     num_experts=8, top_k=2)
    for a, b in zip(o2, outs):
        assert rel_err(a, b) < 1e-5
    assert rel_err(g2, gl) < 1e-5
    # no-MoE configuration returns a plain tuple (config #1 plumbing, :818-819)
    net2 = ref_moe.build_reference_backbone(arch='tiny')
    net2.eval()
    with torch.no_grad():
        r = net2(x, ['single'])
        o = MO.backbone_forward(x, net2.state_dict(), arch='tiny')
    assert isinstance(r, tuple) and isinstance(o, tuple) and len(o) == 4
    for a, b in zip(o, r):
        assert rel_err(a, b) < 1e-5
