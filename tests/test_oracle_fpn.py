"""CPU tier: the FPN oracle (oracle/fpn_oracle.py) is pinned to the fixtures produced by the reference MultitaskFPN
(and, when /root/reference exists, to the live reference module); the MI355X module exposes the reference's
state_dict schema."""
import pytest
import torch

from oracle import fpn_oracle as FO
from tests.fpn_common import CASES, MG, check_run, load


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_fixture(name):
    fx = load(name)
    kw = fx['cfg']
    for sl in fx['runs']:
        p = {k: v.requires_grad_(True) for k, v in MG.seeded_state_dict(fx['shapes']).items()}
        xs = [x.requires_grad_(True) for x in MG.seeded_inputs(kw, fx['s0'], fx['batch'])]
        outs = FO.fpn_forward(xs, p, num_ins=len(kw['in_channels']), num_outs=kw['num_outs'], start_level=sl,
                              add_extra_convs=kw.get('add_extra_convs', False))
        sum((o * q).sum() for o, q in zip(outs, MG.seeded_proj(outs, sl))).backward()
        check_run(fx, sl, outs, {k: v.grad for k, v in p.items() if v.grad is not None}, [x.grad for x in xs],
                  1e-6, 1e-5)


def test_oracle_matches_live_reference():
    from oracle import ref_fpn
    if not ref_fpn.available():
        pytest.skip('/root/reference not present (GPU box)')
    mod = ref_fpn.load_reference_module()
    kw = dict(in_channels=[32, 64, 96, 128], out_channels=64, extra_level=1, add_extra_convs='on_lateral', num_outs=5)
    net = mod.MultitaskFPN(**kw)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(MG.seeded_state_dict(shapes))
    xs = MG.seeded_inputs(kw, 16, 2)
    ref = net(xs, start_level=1, add_extra_convs='on_lateral')
    got = FO.fpn_forward(xs, MG.seeded_state_dict(shapes), num_outs=5, start_level=1, add_extra_convs='on_lateral')
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('name', CASES)
def test_module_state_dict_schema_equals_reference(name):
    from sm3det_amd.fpn import MultitaskFPN
    fx = load(name)
    net = MultitaskFPN(**fx['cfg'])
    sd = net.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == fx['shapes']
    assert list(sd) == list(fx['shapes'])  # same order as the reference module
    ref = MG.seeded_state_dict(fx['shapes'])
    missing, unexpected = net.load_state_dict(ref)
    assert not missing and not unexpected
    for k, v in net.state_dict().items():  # round trip through the kernel layouts is exact
        assert torch.equal(v, ref[k]), k
    c = net.fpn_convs[0].conv
    assert tuple(c.weight.shape) == (c.out_channels, 3, 3, c.in_channels)
    net.init_weights()
    assert float(net.lateral_convs[0].conv.bias.abs().max()) == 0.0


def test_unsupported_options_raise():
    from sm3det_amd.fpn import MultitaskFPN
    with pytest.raises(NotImplementedError):
        MultitaskFPN([32, 64], 64, 2, norm_cfg=dict(type='BN'))
    with pytest.raises(NotImplementedError):
        MultitaskFPN([32, 64], 64, 2, upsample_cfg=dict(mode='bilinear'))
    net = MultitaskFPN([32, 64], 64, 2)
    with pytest.raises(Exception):
        net([torch.zeros(1, 32, 8, 8), torch.zeros(1, 64, 4, 4)])  # CPU tensors: no fallback
