"""Generate tests/golden/moe_*.pt by running the REFERENCE backbone module itself
(/root/reference/mmrotate/models/backbones/convnext_moe.py, imported unmodified through oracle/ref_moe.py).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden_moe.py

Each fixture holds: the constructor kwargs, a randomised state_dict (reference key schema), the input batch, the
injected train-mode randomness (gate noise per MoE block = what torch.randn_like would draw at :203, drop-path
factors per block = timm DropPath masks / keep_prob), and the reference's results: eval outputs + gate loss, train
outputs + gate loss, and the gradients of  L = sum_i <out_i, R_i> + 10 * gate_loss  w.r.t. every parameter
(R_i = torch.randn(out_i.shape, generator=manual_seed(100+i))).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_moe  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    'moe_e4k2': dict(arch=dict(depths=[1, 1, 2, 1], channels=[32, 32, 64, 64]),
                     MoE_Block_inds=[[], [0], [0], [0]], num_experts=4, top_k=2, drop_path_rate=0.3,
                     input=(2, 3, 64, 64)),
    'moe_lin_e4k2': dict(arch=dict(depths=[1, 1, 2, 1], channels=[32, 32, 64, 64]), gate='linear',
                         MoE_Block_inds=[[], [0], [1], [0]], num_experts=4, top_k=2, drop_path_rate=0.2,
                         input=(2, 3, 64, 64)),
    'moe_e8k3': dict(arch=dict(depths=[1, 2, 1, 1], channels=[32, 32, 32, 64]),
                     MoE_Block_inds=[[], [1], [0], []], num_experts=8, top_k=3, drop_path_rate=0.2,
                     input=(1, 3, 96, 64)),
}


class _FixedDrop(torch.nn.Module):
    def __init__(self, rs):
        super().__init__()
        self.rs = rs

    def forward(self, x):
        return x * self.rs.view(-1, 1, 1, 1)


def randomise(net, g):
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith('gamma'):
                p.copy_(torch.empty_like(p).uniform_(0.5, 1.5, generator=g))
            elif 'w_noise' in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
            elif n.endswith('w_gate'):  # gate='linear': (C, E) parameter, zeros by default
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
            elif 'sim_matrix' in n:
                p.copy_(torch.randn(p.shape, generator=g))
            elif 'temperature' in n:
                p.copy_(torch.tensor([1.2]))
            elif ('norm' in n or 'downsample_layers' in n) and p.dim() == 1 and n.endswith('weight'):
                p.copy_(torch.empty_like(p).uniform_(0.5, 1.5, generator=g))
            elif n.endswith('bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)


def loss_of(outs, gl):
    L = 10.0 * gl
    for i, o in enumerate(outs):
        R = torch.randn(o.shape, generator=torch.Generator().manual_seed(100 + i))
        L = L + (o * R).sum()
    return L


def make(name, cfg):
    cfg = dict(cfg)
    shape = cfg.pop('input')
    g = torch.Generator().manual_seed(1234)
    torch.manual_seed(0)
    net = ref_moe.build_reference_backbone(**cfg)
    randomise(net, g)
    x = torch.randn(shape, generator=g)
    B = shape[0]
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    # ---- eval
    net.eval()
    with torch.no_grad():
        outs, gl = net(x, ['single'])
    ev = dict(outs=[o.clone() for o in outs], gate_loss=gl.clone())
    # ---- train with injected randomness
    net.train()
    blocks = [b for st in net.stages for b in st]
    keep = [1.0 - float(getattr(b.drop_path, 'drop_prob', 0.0)) for b in blocks]
    drop_scale = []
    for b, kp in zip(blocks, keep):
        if kp < 1.0:
            m = torch.empty(B).bernoulli_(kp, generator=g)
            if B > 1 and m.sum() == 0:
                m[0] = 1.0
            rs = m / kp
        else:
            rs = torch.ones(B)
        drop_scale.append(rs)
        b.drop_path = _FixedDrop(rs)
    E = cfg['num_experts']
    noise = []
    # token counts per MoE block, in forward order
    H, W = shape[2] // 4, shape[3] // 4
    for i, st in enumerate(net.stages):
        if i > 0:
            H, W = H // 2, W // 2
        for b in st:
            if b.MoE_cfg is not None:
                noise.append(torch.randn(B * H * W, E, generator=g))
    it = iter(noise)
    real = torch.randn_like
    torch.randn_like = lambda t, *a, **k: next(it).to(t.dtype)
    try:
        outs, gl = net(x, ['single'])
        L = loss_of(outs, gl)
        L.backward()
    finally:
        torch.randn_like = real
    tr = dict(outs=[o.detach().clone() for o in outs], gate_loss=gl.detach().clone(),
              grads={k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    fx = dict(cfg=cfg, state_dict=sd, x=x, noise=noise, drop_scale=drop_scale, eval=ev, train=tr,
              torch_version=torch.__version__)
    path = os.path.join(HERE, name + '.pt')
    torch.save(fx, path)
    print(name, 'params', sum(v.numel() for v in sd.values()), 'file MB', os.path.getsize(path) / 1e6,
          'train gate_loss', float(tr['gate_loss']), 'eval gate_loss', float(ev['gate_loss']))


if __name__ == '__main__':
    assert ref_moe.available(), 'needs /root/reference'
    for n, c in CASES.items():
        if len(sys.argv) > 1 and n not in sys.argv[1:]:
            continue
        make(n, c)
