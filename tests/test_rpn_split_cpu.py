"""CPU tier: `rpn_head._SplitClsReg` -- the node that presents the RPN tower's fused, padded head output (B, H, W, NP) as the
objectness (B, A, H, W) and delta (B, 6A, H, W) maps -- equals plain slicing in forward and backward (generic path: two
independent incoming gradients, or only one of them), and its shortcut (both gradients are the matching views of ONE
zero-initialised buffer of the fused layout, as `det_losses._RPNLoss.backward` produces them) returns that buffer itself."""
import torch

from sm3det_amd.rpn_head import _SplitClsReg


def _slices(o, A):
    return o[..., :A].permute(0, 3, 1, 2), o[..., A:7 * A].permute(0, 3, 1, 2)


def test_split_node_equals_slicing_forward_and_backward():
    torch.manual_seed(0)
    B, H, W, A, NP = 2, 5, 7, 3, 32
    o1 = torch.randn(B, H, W, NP, requires_grad=True)
    o2 = o1.detach().clone().requires_grad_(True)
    c1, r1 = _SplitClsReg.apply(o1, A)
    c2, r2 = _slices(o2, A)
    assert torch.equal(c1, c2) and torch.equal(r1, r2) and c1.shape == (B, A, H, W) and r1.shape == (B, 6 * A, H, W)
    wc, wr = torch.randn_like(c1), torch.randn_like(r1)
    ((c1 * wc).sum() + (r1 * wr).sum()).backward()
    ((c2 * wc).sum() + (r2 * wr).sum()).backward()
    assert torch.equal(o1.grad, o2.grad)
    assert float(o1.grad[..., 7 * A:].abs().max()) == 0.0  # the padding channels get no gradient
    # only one output used
    o3 = o1.detach().clone().requires_grad_(True)
    c3, _ = _SplitClsReg.apply(o3, A)
    (c3 * wc).sum().backward()
    exp = torch.zeros_like(o3)
    exp[..., :A] = wc.permute(0, 2, 3, 1)
    assert torch.equal(o3.grad, exp)


def test_split_node_hands_on_a_shared_gradient_buffer():
    B, H, W, A, NP = 2, 4, 4, 3, 32
    o = torch.randn(B, H, W, NP, requires_grad=True)
    c, r = _SplitClsReg.apply(o, A)
    buf = torch.zeros(B, H, W, NP)
    buf[..., :7 * A] = torch.randn(B, H, W, 7 * A)
    buf._sm3_rest_is_zero = True
    hits = _SplitClsReg.fast_hits
    torch.autograd.backward([c, r], [buf[..., :A].permute(0, 3, 1, 2), buf[..., A:7 * A].permute(0, 3, 1, 2)])
    assert _SplitClsReg.fast_hits == hits + 1
    assert torch.equal(o.grad, buf)
    # views of two different buffers, or a buffer without the zero-rest guarantee: the generic path, same values
    o2 = o.detach().clone().requires_grad_(True)
    c2, r2 = _SplitClsReg.apply(o2, A)
    other = buf.clone()
    torch.autograd.backward([c2, r2], [buf[..., :A].permute(0, 3, 1, 2), other[..., A:7 * A].permute(0, 3, 1, 2)])
    assert _SplitClsReg.fast_hits == hits + 1 and torch.equal(o2.grad, buf)
