"""two-segment backward (bench.py N>1 path) vs one backward: parameter gradients must agree (same RNG seed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

net = bench.build_model().cuda().train()
x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(1)).cuda()
named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
LATE = ('stages.2.', 'stages.3.', 'downsample_layers.2.', 'downsample_layers.3.', 'norm2.', 'norm3.')
late = [p for n, p in reversed(named) if n.startswith(LATE)]
early = [p for n, p in reversed(named) if not n.startswith(LATE)]
proj = None

def one():
    global proj
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    for _, p in named: p.grad = None
    outs, gl = net(x, ['single'])
    if proj is None:
        proj = [torch.randn_like(o) for o in outs]
    l = bench.loss_fn(outs, gl, proj)
    l.backward()
    return float(l), {n: p.grad.clone() for n, p in named}

def two():
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    for _, p in named: p.grad = None
    net.stage_boundary = 1
    outs, _ = net(x, ['single'])
    net.stage_boundary = None
    terms, hb = net._gate_loss_terms, net._boundary_tokens
    l_late = sum(g for i, g in terms if i >= 2) / len(terms)
    l_early = sum(g for i, g in terms if i < 2) / len(terms)
    for i, (o, r) in enumerate(zip(outs, proj)):
        t = (o * r).sum() * 1e-4
        if i >= 2: l_late = l_late + t
        else: l_early = l_early + t
    grads = torch.autograd.grad(l_late, [hb] + late, allow_unused=True)
    for p, g in zip(late, grads[1:]): p.grad = g
    torch.autograd.backward([l_early, hb], grad_tensors=[torch.ones_like(l_early), grads[0]], inputs=early)
    return float(l_late + l_early), {n: (p.grad.clone() if p.grad is not None else None) for n, p in named}

l1, g1 = one()
l2, g2 = two()
print('loss', l1, l2)
worst = (0.0, None)
for n in g1:
    if g2[n] is None:
        print('MISSING', n); continue
    e = float((g1[n] - g2[n]).abs().max() / g1[n].abs().max().clamp_min(1e-30))
    if e > worst[0]: worst = (e, n)
print('worst rel diff', worst)
