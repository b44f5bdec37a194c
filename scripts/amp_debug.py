"""debug aid: AMP forward with fp16 storage vs fp32 storage (operands rounded in the loader) must agree bit for bit in the
forward pass -- find the first block where they do not, and unit-check each fp16-storage GEMM form against torch."""
import os
import sys

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sm3det_amd import _lib_backbone as LB, amp, backbone_ops as ops  # noqa: E402
from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput  # noqa: E402
from tests.moe_common import load_fixture  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def gemm_units():
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).cuda()  # noqa: E731
    for (M, N, K) in [(256, 128, 192), (1000, 384, 96), (300, 96, 384)]:
        A, W, b = (r(M, K) * 0.5), r(N, K) * 0.1, r(N)
        ref = A.half().double() @ W.half().double().t() + b.double()
        with amp.autocast():
            C0 = torch.zeros(M, N, device='cuda')
            LB.gemm(LB.NT, A, W, C0, M, N, K, epilogue=LB.EPI_BIAS, bias=b)
            C1 = torch.zeros(M, N, device='cuda')
            LB.gemm(LB.NT, A.half(), W, C1, M, N, K, epilogue=LB.EPI_BIAS, bias=b)
            act = torch.zeros(M, N, device='cuda', dtype=torch.half)
            dact = torch.zeros(M, N, device='cuda', dtype=torch.half)
            LB.gemm(LB.NT, A.half(), W, act, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=b, aux_out=dact)
            act0, dact0 = torch.zeros(M, N, device='cuda'), torch.zeros(M, N, device='cuda')
            LB.gemm(LB.NT, A, W, act0, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=b, aux_out=dact0)
        print(f'NT {M}x{N}x{K}: fp32-storage vs ref {rel(C0, ref):.2e}; A16 vs ref {rel(C1, ref):.2e}; '
              f'A16 == fp32-storage: {torch.equal(C0, C1)}; GELU f16 vs fp32-storage.half(): '
              f'{torch.equal(act, act0.half())} / {torch.equal(dact, dact0.half())} '
              f'({rel(act, act0):.2e}, {rel(dact, dact0):.2e})')
        # TN
        Kt = 3000
        dy, x = r(Kt, M) * 0.3, r(Kt, N) * 0.5
        with amp.autocast():
            D0 = torch.zeros(M, N, device='cuda')
            LB.gemm(LB.TN, dy, x, D0, M, N, Kt)
            D1 = torch.zeros(M, N, device='cuda')
            LB.gemm(LB.TN, dy, x.half(), D1, M, N, Kt)
            D2 = torch.zeros(M, N, device='cuda')
            LB.gemm(LB.TN, dy.half(), x.half(), D2, M, N, Kt)
        refd = dy.half().double().t() @ x.half().double()
        print(f'TN {M}x{N}x{Kt}: fp32-storage {rel(D0, refd):.2e}  B16 {rel(D1, refd):.2e}  A16|B16 {rel(D2, refd):.2e}')
        # NN
        dh, W2 = r(M, N) * 0.3, r(N, K) * 0.1
        with amp.autocast():
            X0 = torch.zeros(M, K, device='cuda')
            LB.gemm(LB.NN, dh, W2, X0, M, K, N)
            X1 = torch.zeros(M, K, device='cuda')
            LB.gemm(LB.NN, dh.half(), W2, X1, M, K, N)
        print(f'NN {M}x{K}x{N}: fp32-storage {rel(X0, dh.half().double() @ W2.half().double()):.2e}  A16 '
              f'{rel(X1, dh.half().double() @ W2.half().double()):.2e}  equal {torch.equal(X0, X1)}')


def blocks():
    fx = load_fixture('moe_e4k2')
    net = ConvNeXt_moe_MultiInput(**fx['cfg'])
    net.load_state_dict(fx['state_dict'])
    net = amp.wrap_fp16_model(net.cuda()).train()
    outs = {}
    for mode in (True, False):
        ops.AMP_HALF_STORAGE = mode
        rec = []
        hooks = []
        for i, st in enumerate(net.stages):
            for j, blk in enumerate(st):
                orig = blk.forward_tokens

                def wrapped(x, B, H, W, noise=None, drop_scale=None, _o=orig, _i=i, _j=j):
                    r_ = _o(x, B, H, W, noise=noise, drop_scale=drop_scale)
                    rec.append((f'stage{_i}.block{_j}', r_[0].detach().clone()))
                    return r_
                blk.forward_tokens = wrapped
                hooks.append((blk, orig))
        o, gl = net(fx['x'].cuda(), ['single'], noise=[n.cuda() for n in fx['noise']],
                    drop_scale=[d.cuda() for d in fx['drop_scale']])
        for blk, orig in hooks:
            blk.forward_tokens = orig
        outs[mode] = (rec, [t.detach().clone() for t in o], [b.ffn.last_top_idx.clone() for st in net.stages for b in st
                                                              if b.MoE_cfg is not None])
    for (n1, a), (n2, b) in zip(outs[True][0], outs[False][0]):
        print(n1, 'equal' if torch.equal(a, b) else f'DIFF {rel(a, b):.3e}')
    for a, b in zip(outs[True][2], outs[False][2]):
        print('routing equal:', torch.equal(a, b), int((a != b).any(1).sum()), 'of', a.shape[0])


if __name__ == '__main__':
    gemm_units()
    blocks()
