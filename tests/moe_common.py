"""helpers shared by the MoE oracle (CPU) and backbone (GPU) tests"""
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_fixture(name):
    return torch.load(os.path.join(GOLDEN, name + '.pt'), map_location='cpu', weights_only=False)


def loss_of(outs, gl):
    """L = sum_i <out_i, R_i> + 10 * gate_loss, R_i from manual_seed(100+i) (same as make_golden_moe.py)."""
    L = 10.0 * gl
    for i, o in enumerate(outs):
        R = torch.randn(o.shape, generator=torch.Generator().manual_seed(100 + i)).to(o.device, o.dtype)
        L = L + (o * R).sum()
    return L


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def oracle_kwargs(cfg):
    return dict(arch=cfg['arch'], moe_block_inds=cfg['MoE_Block_inds'], num_experts=cfg['num_experts'],
                top_k=cfg['top_k'])
