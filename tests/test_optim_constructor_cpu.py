"""CPU tier: `sm3det_amd.optim.reference_param_options` / `param_groups_from_cfg` -- the per-parameter `lr` /
`weight_decay` of the reference's optimizer -- against the REFERENCE'S OWN `DefaultOptimizerConstructor`
(mmcv/mmcv/runner/optimizer/default_constructor.py, imported unmodified by oracle/ref_optim.py) run on the REFERENCE'S
OWN backbone module (oracle/ref_moe.py), for the `paramwise_cfg` of the SM3Det configs and for ones that exercise every
rule of `add_params`: custom keys (longest first), bias lr, norm / depth-wise / bias decay."""
import pytest
import torch

from oracle import ref_moe, ref_optim

pytestmark = pytest.mark.skipif(not (ref_optim.available() and ref_moe.available()), reason='/root/reference not present')

KW = dict(arch='tiny', MoE_Block_inds=[[], [0], [0, 2], [0]], num_experts=4, top_k=2, drop_path_rate=0.1)
SM3DET = dict(custom_keys={k: dict(lr_mult=1.0) for k in ('backbone', 'neck', 'sar_bbox_head', 'rgb_rpn_head', 'rgb_roi_head',
                                                          'ifr_rpn_head', 'ifr_roi_head')})
CASES = [
    SM3DET,
    None,
    dict(norm_decay_mult=0.0, bias_lr_mult=2.0, bias_decay_mult=0.1, dwconv_decay_mult=0.5),
    dict(norm_decay_mult=0.0, bias_lr_mult=2.0, dwconv_decay_mult=0.5,
         custom_keys={'backbone.stages.2': dict(lr_mult=0.1), 'stages.2.1.norm': dict(lr_mult=3.0, decay_mult=0.0),
                      'w_gate': dict(decay_mult=0.0), 'dataset_stems': dict(lr_mult=0.5)}),
]


class _Detector(torch.nn.Module):
    def __init__(self, backbone):
        super().__init__()
        self.backbone = backbone


@pytest.mark.parametrize('pw', CASES)
def test_param_options_equal_the_references_optimizer_constructor(pw):
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    from sm3det_amd.optim import param_groups_from_cfg, reference_param_options
    base = dict(type='AdamW', lr=1e-4, betas=(0.9, 0.999), weight_decay=0.05)
    ref_net = ref_moe.build_reference_backbone(**KW)
    Ctor = ref_optim.load()
    opt = Ctor(dict(base), pw)(_Detector(ref_net))
    names = [n for n, _ in _Detector(ref_net).named_parameters()]
    if pw:
        assert len(opt.param_groups) == len(names)  # one group per parameter, in named_parameters() order
        ref = {n: (g['lr'], g['weight_decay']) for n, g in zip(names, opt.param_groups)}
    else:
        assert len(opt.param_groups) == 1
        ref = {n: (opt.param_groups[0]['lr'], opt.param_groups[0]['weight_decay']) for n in names}

    net = ConvNeXt_moe_MultiInput(**KW)
    cfg = dict(base, paramwise_cfg=pw) if pw else dict(base)
    mine = reference_param_options(net, cfg, 'backbone')
    assert ['backbone.' + k for k in mine] == names  # same parameters, same order
    for k, o in mine.items():
        assert (o['lr'], o['weight_decay']) == pytest.approx(ref['backbone.' + k], rel=1e-12), k
    if pw and len({v for v in ref.values()}) > 3:
        assert len({(round(o['lr'], 12), o['weight_decay']) for o in mine.values()}) > 3  # the rules were exercised

    # carried onto this package's storage (fused expert tensors, tap-major depthwise weights): one group per tensor
    groups = param_groups_from_cfg({'backbone': net}, cfg)
    assert len(groups) == sum(1 for p in net.parameters() if p.requires_grad)
    by_name = {g['name']: g for g in groups}
    g = by_name['backbone.stages.2.0.ffn.w1'] if 'backbone.stages.2.0.ffn.w1' in by_name else None
    if g is not None:
        r = ref['backbone.stages.2.0.ffn.experts.0.pointwise_conv1.weight']
        assert (g.get('lr', base['lr']), g.get('weight_decay', base['weight_decay'])) == pytest.approx(r)


def test_custom_key_that_splits_a_fused_tensor_is_refused():
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    from sm3det_amd.optim import param_groups_from_cfg
    net = ConvNeXt_moe_MultiInput(**KW)
    cfg = dict(type='AdamW', lr=1e-4, weight_decay=0.05, paramwise_cfg=dict(custom_keys={'experts.1.': dict(lr_mult=0.1)}))
    with pytest.raises(NotImplementedError):
        param_groups_from_cfg({'backbone': net}, cfg)
