"""CPU tier, SURVEY 8(f) row 4: loading an ImageNet ConvNeXt checkpoint (mmcls key schema) into the MoE backbone.
``remap_pretrained_state_dict`` + ``load_state_dict`` of the MI355X module must leave it with exactly the parameters the
REFERENCE module ends up with after its own ``init_weights()`` (mmrotate/models/backbones/convnext_moe.py:824-899) on
the same checkpoint; ``get_layer_depth`` (:616-658, drives the layer-wise lr decay) must agree name by name.
Needs /root/reference (skipped on the GPU box)."""
import pytest
import torch

ARCH = dict(depths=[1, 1, 3, 1], channels=[32, 64, 96, 128])
MOE = dict(MoE_Block_inds=[[], [0], [0, 2], [0]], num_experts=4, top_k=2)


def _dense_checkpoint():
    """a synthetic 'ImageNet ConvNeXt' checkpoint in the mmcls schema the reference expects (backbone.* keys, dense
    FFNs, stem = downsample_layers.0.{0,1}), plus a classifier head that must be dropped"""
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    torch.manual_seed(11)
    dense = ConvNeXt_moe_MultiInput(arch=ARCH)
    ck = {}
    for k, v in dense.state_dict().items():
        v = torch.randn_like(v) if v.is_floating_point() else v
        k = k.replace('ffn.pointwise_conv', 'pointwise_conv')
        if k.startswith('downsample_layers.0.0.'):
            k = k.replace('downsample_layers.0.0.', 'downsample_layers.0.1.')
        elif k.startswith('dataset_stems.single.'):
            k = k.replace('dataset_stems.single.', 'downsample_layers.0.0.')
        ck['backbone.' + k] = v
    ck['head.fc.weight'] = torch.randn(10, 128)
    return ck


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def test_pretrained_load_equals_reference_init_weights():
    from oracle import ref_moe
    if not ref_moe.available():
        pytest.skip('/root/reference not present (GPU box)')
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    mod = ref_moe.load_reference_module()
    ck = _dense_checkpoint()

    class Loader:
        @staticmethod
        def load_checkpoint(path, logger=None, map_location='cpu'):
            return dict(state_dict=ck)
    saved_loader, saved_logger = mod.CheckpointLoader, mod.MMLogger
    mod.CheckpointLoader = Loader
    mod.MMLogger = type('L', (), {'get_current_instance': staticmethod(lambda: type(
        'Lg', (), {'warn': staticmethod(lambda *a: None), 'info': staticmethod(lambda *a: None)})())})
    try:
        torch.manual_seed(1)
        ref = mod.ConvNeXt_moe_MultiInput(arch=ARCH, init_cfg=_Cfg(type='Pretrained', checkpoint='fake.pth'), **MOE)
        ref.init_weights()
    finally:
        mod.CheckpointLoader, mod.MMLogger = saved_loader, saved_logger
    torch.manual_seed(2)  # different random init on purpose: everything the checkpoint covers must be overwritten
    net = ConvNeXt_moe_MultiInput(arch=ARCH, **MOE)
    res = net.load_state_dict(net.remap_pretrained_state_dict(dict(state_dict=ck)['state_dict']), strict=False)
    assert not res.unexpected_keys
    rsd, sd = ref.state_dict(), net.state_dict()
    assert list(rsd) == list(sd)
    covered = 0
    for k in rsd:
        if k in res.missing_keys:  # gate parameters etc.: not in an ImageNet checkpoint, stay at their random init
            assert any(t in k for t in ('w_gate', 'w_noise', '.mean', '.std')), k
            continue
        assert torch.equal(rsd[k], sd[k]), k
        covered += 1
    assert covered > 40
    # every expert of an MoE block received the dense FFN of that block
    assert torch.equal(net.stages[2][2].ffn.w1[3], ck['backbone.stages.2.2.pointwise_conv1.weight'])
    # layer-wise lr-decay depth of every parameter name, with and without the 'backbone.' prefix
    for n, _ in ref.named_parameters():
        assert ref.get_layer_depth(n) == net.get_layer_depth(n), n
        assert ref.get_layer_depth('backbone.' + n, 'backbone.') == net.get_layer_depth('backbone.' + n, 'backbone.')
    assert ref.get_layer_depth('neck.x', 'backbone.') == net.get_layer_depth('neck.x', 'backbone.')


def test_reference_optimizer_state_is_translated_through_the_state_dict_hooks():
    """`resume_from` of a reference checkpoint: the reference's AdamW keeps one state entry per reference parameter
    (per-expert Linear tensors, depthwise (C,1,7,7)); MultiTensorAdamW.load_reference_state must land them in the fused /
    tap-major storage exactly where load_state_dict lands the weights."""
    import torch
    from sm3det_amd.convnext_moe import ConvNeXt_moe
    from sm3det_amd.optim import MultiTensorAdamW
    torch.manual_seed(0)
    net = ConvNeXt_moe(arch=dict(depths=[1, 1, 1, 1], channels=[32, 32, 32, 32]), MoE_Block_inds=[[], [0], [], []],
                       num_experts=4, top_k=2)
    buf = {n for n, _ in net.named_buffers()}
    ref = {k: v for k, v in net.state_dict().items() if k not in buf}   # reference keys / shapes / order
    g = torch.Generator().manual_seed(1)
    state = {i: dict(step=torch.tensor(7.0), exp_avg=torch.randn(v.shape, generator=g),
                     exp_avg_sq=torch.rand(v.shape, generator=g)) for i, v in enumerate(ref.values())}
    ref_sd = dict(state=state, param_groups=[dict(params=[i], lr=1e-4) for i in range(len(ref))])
    opt = MultiTensorAdamW([dict(params=[p]) for p in net.parameters()], lr=1e-4)
    opt.load_reference_state(net, ref_sd)
    keys = list(ref.keys())
    moe = net.stages[1][0].ffn
    i = keys.index('stages.1.0.ffn.experts.2.pointwise_conv1.weight')
    assert torch.equal(opt.state[moe.w1]['exp_avg'][2], state[i]['exp_avg'])
    assert torch.equal(opt.state[moe.w1]['exp_avg_sq'][2], state[i]['exp_avg_sq'])
    j = keys.index('stages.0.0.depthwise_conv.weight')
    dw = net.stages[0][0].depthwise_conv
    assert opt.state[dw.weight]['exp_avg'].shape == (49, 32)
    assert torch.equal(opt.state[dw.weight]['exp_avg'], state[j]['exp_avg'].reshape(32, 49).t())
    assert float(opt._step) == 7.0 and all('exp_avg' in opt.state[p] for p in net.parameters())
    # a state that misses some experts of a fused tensor cannot be fused: rejected, not silently zero-filled
    del state[i]
    import pytest
    with pytest.raises(ValueError):
        opt.load_reference_state(net, ref_sd)
