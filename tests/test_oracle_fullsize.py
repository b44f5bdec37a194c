"""CPU tier: pins oracle/moe_oracle.py at the HEADLINE size -- config #2 (ConvNeXt-T, 8 experts top-2) on
1x3x1024x1024, train-mode forward + backward -- against the fixture the REFERENCE module produced
(tests/golden/make_golden_fullsize.py).  Same element-wise metric as the GPU test; both sides are fp32 CPU
computations, so the tolerances are 10x tighter (forward 1e-5, gradients 1e-4).  ~10 s on 8 cores."""
import torch

from oracle import moe_oracle as MO
from tests import fullsize_common as FC


def test_oracle_replays_full_size_reference_fixture():
    case = 'full_e8t2_b1'
    fx = FC.load(case)
    cfg, seed = fx['cfg'], fx['seed']
    # key/shape template of the reference schema: the MI355X mirror exposes it without touching the GPU
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    template = ConvNeXt_moe_MultiInput(**cfg).state_dict()
    sd = FC.seeded_state_dict(template, seed)
    p = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('.mean', '.std'))) for k, v in sd.items()}
    x, noise, drop = FC.make_inputs(case, noise_seed=fx['noise_seed'])
    outs, gl = MO.backbone_forward(x, p, arch=cfg['arch'], moe_block_inds=cfg['MoE_Block_inds'],
                                   num_experts=cfg['num_experts'], top_k=cfg['top_k'], train=True, noise=noise,
                                   drop_scale=drop)
    for i, (o, ref) in enumerate(zip(outs, fx['outs'])):
        for name, e in FC.compare_output(i, o, ref).items():
            assert float(e.max()) < 1e-5, (i, name, float(e.max()))
    assert abs(float(gl) - fx['gate_loss']) / abs(fx['gate_loss']) < 1e-5
    FC.loss_of(outs, gl, seed).backward()
    table = fx['grads']['table']
    worst = (0.0, None)
    for key in table:
        assert p[key].grad is not None, key
        e, l2 = FC.compare_grad(key, p[key].grad, fx['grads'])
        if max(e, l2) > worst[0]:
            worst = (max(e, l2), key)
    assert worst[0] < 1e-4, worst
    assert len(table) > 400
