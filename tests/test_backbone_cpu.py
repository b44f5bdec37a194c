"""CPU tier (no kernels run): registry surface, constructor kwargs and the state_dict key schema of the backbone
mirror must equal the reference's (fixture keys come from the reference module itself)."""
import pytest
import torch

from tests.moe_common import load_fixture


@pytest.mark.parametrize('name', ['moe_e4k2', 'moe_e8k3'])
def test_state_dict_schema_equals_reference(name):
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    fx = load_fixture(name)
    net = ConvNeXt_moe_MultiInput(**fx['cfg'])
    mine = net.state_dict()
    assert set(mine.keys()) == set(fx['state_dict'].keys())
    for k, v in fx['state_dict'].items():
        assert tuple(mine[k].shape) == tuple(v.shape), k
    net.load_state_dict(fx['state_dict'], strict=True)
    again = net.state_dict()
    for k, v in fx['state_dict'].items():
        assert torch.equal(again[k], v), k
    # fused expert storage really holds the per-expert tensors
    for n, m in net.named_modules():
        if m.__class__.__name__ == 'MoE_layer':
            assert torch.equal(m.w1[1], fx['state_dict'][f'{n}.experts.1.pointwise_conv1.weight'])


def test_registry_builds_main_sm3det_backbone_cfg():
    from sm3det_amd.registry import ROTATED_BACKBONES
    import sm3det_amd.convnext_moe  # noqa: F401  (registers the classes)
    cfg = dict(type='ConvNeXt_moe_MultiInput', MoE_Block_inds=[[], [0, 2], [i * 2 for i in range(5)], [0, 2]],
               datasets=None, num_experts=8, top_k=2, arch='tiny', drop_path_rate=0.1,
               init_cfg=dict(type='Pretrained', prefix='backbone', checkpoint='../data/pretrained/convnext-tiny.pth'))
    net = ROTATED_BACKBONES.build(cfg)  # local_configs/main_SM3Det.py:13-21
    n_params = sum(p.numel() for p in net.parameters())
    assert abs(n_params / 1e6 - 140.28) < 0.01  # 140.28 M
    n_moe = sum(1 for m in net.modules() if m.__class__.__name__ == 'MoE_layer')
    assert n_moe == 9
    assert 'ConvNeXt_moe' in ROTATED_BACKBONES and len(ROTATED_BACKBONES) >= 2


def test_pretrained_key_remap_clones_dense_ffn_into_experts():
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    net = ConvNeXt_moe_MultiInput(arch=dict(depths=[1, 1, 1, 1], channels=[32, 32, 32, 32]),
                                  MoE_Block_inds=[[], [0], [], []], num_experts=4, top_k=2)
    ck = {'backbone.downsample_layers.0.0.weight': torch.randn(32, 3, 4, 4),
          'backbone.downsample_layers.0.1.weight': torch.randn(32),
          'backbone.stages.1.0.pointwise_conv1.weight': torch.randn(128, 32),
          'backbone.stages.0.0.pointwise_conv1.weight': torch.randn(128, 32),
          'head.fc.weight': torch.randn(3)}
    sd = net.remap_pretrained_state_dict(ck)
    assert 'dataset_stems.single.weight' in sd and 'downsample_layers.0.0.weight' in sd
    assert all(f'stages.1.0.ffn.experts.{e}.pointwise_conv1.weight' in sd for e in range(4))
    assert 'stages.0.0.ffn.pointwise_conv1.weight' in sd and not any(k.startswith('head') for k in sd)
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    assert torch.equal(net.stages[1][0].ffn.w1[3], ck['backbone.stages.1.0.pointwise_conv1.weight'])


def test_unsupported_options_raise():
    from sm3det_amd.convnext_moe import ConvNeXt_moe
    with pytest.raises(NotImplementedError):
        ConvNeXt_moe(arch='tiny', gate='softmax', MoE_Block_inds=[[], [0], [], []])  # the reference knows cosine | linear
    with pytest.raises(NotImplementedError):
        ConvNeXt_moe(arch='tiny', use_grn=True)
    # gate='linear' constructs with the reference's parameter: w_gate (C, E) zeros instead of the CosineTopKGate module
    net = ConvNeXt_moe(arch='tiny', gate='linear', MoE_Block_inds=[[], [0], [], []], num_experts=4)
    sd = net.state_dict()
    assert sd['stages.1.0.ffn.w_gate'].shape == (192, 4) and 'stages.1.0.ffn.w_gate.temperature' not in sd


def test_collect_by_source_matches_reference_gather_logic():
    """host half of TriSourceDetector.gather_dict_values (trisource_H1stage_R2stage_detector.py:190-196)"""
    import torch
    from sm3det_amd.h2d import collect_by_source
    data = [dict(sar=torch.zeros(2), rgb=None), dict(rgb=torch.ones(3)), dict(sar=torch.ones(2), ifr='meta')]
    got = collect_by_source(data, ['sar', 'rgb', 'ifr'])
    assert [t.tolist() for t in got['sar']] == [[0.0, 0.0], [1.0, 1.0]]
    assert len(got['rgb']) == 1 and got['ifr'] == ['meta']


def test_backward_schedule_switches_host_logic():
    """host side of the concurrent-backward-partner schedule (backbone_ops): the `pairing` context sets and restores the
    level, `_compute` carries a node's level into its backward, nothing is deferred outside a backward pass, and the per-level
    stream helper of the heads degrades to the plain loop off the GPU"""
    from sm3det_amd import backbone_ops as B, level_streams
    from sm3det_amd import _lib_backbone as LB
    base = B.PAIR_DGRAD
    with B.pairing(0):
        assert B.PAIR_DGRAD == 0
        with B.pairing(3):
            assert B.PAIR_DGRAD == 3
        assert B.PAIR_DGRAD == 0
    assert B.PAIR_DGRAD == base
    mode = LB.COMPUTE
    with B._compute(mode, pair=2):  # a node whose forward ran at level 2 runs its backward at level 2 ...
        assert B.PAIR_DGRAD == 2
    with B._compute(mode):          # ... and one that recorded nothing leaves the switch alone
        assert B.PAIR_DGRAD == base
    assert B.PAIR_DGRAD == base and LB.COMPUTE == mode
    assert not B._callback_armed(torch.device('cpu'))  # no backward pass running: a join can never be deferred
    # below the requested level / off the GPU a pair is just the two calls (weight gradient first, as before round 6)
    order = []
    with B.pairing(0):
        r = B._paired(torch.device('cpu'), lambda: order.append('main') or 'm', lambda: order.append('side') or 's', 1)
    assert r == ('m', 's') and order == ['side', 'main']
    outs = level_streams.map_levels(lambda x, s: x * s, [torch.ones(2), torch.ones(3)], [2.0, 3.0])
    assert [float(o.sum()) for o in outs] == [4.0, 9.0]
