"""GPU tier: the PERSISTENT form of the GEMM family (gemm_f32_kernel.h "work items"; gemm_f32_p.hip / gemm_h16_p.hip) --
an opt-in measurement form (bit 17 of sm3_gemm_desc.tuning; slower than one workgroup per tile on this chip, see
gemm_f32.hip) that sm3_gemm_f32 uses when a GEMM has more output tiles than the chip holds workgroups at once.  Every case
here is sized past that threshold (> 1024 tiles), so the kernels under test walk a device-side item queue, prefetch the
next item's first k-tile inside the current epilogue and leave the queue zeroed for the next launch.

Reference: the fp64 product (tolerance 1e-4 of the result's scale for fp32 operands, fp16 cases against the product of the
fp16-rounded operands at 2e-5 + output rounding).  Each case runs TWICE back to back (the second launch finds the queue the
first one left) and must give bit-identical results; a split-K launch in between uses the ticket slots below the queue."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


PERSIST_BIT = 1 << 17


@pytest.fixture(autouse=True)
def _persistent_form():
    from sm3det_amd import _lib_backbone as LB
    old, LB.TUNING = LB.TUNING, PERSIST_BIT
    yield
    LB.TUNING = old


def _LB():
    from sm3det_amd import _lib_backbone as LB
    return LB


def _rand(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).cuda()


def _close(a, ref64, tol=1e-4):
    err = (a.double() - ref64).abs().max().item()
    scale = ref64.abs().max().item() + 1e-12
    assert err <= tol * scale, (err, scale)


def _queue_is_zero(LB, dev):
    c = LB.gemm_counters(dev)
    assert int(c[-16:].abs().sum()) == 0, c[-16:].tolist()
    assert int(c.abs().sum()) == 0


def _splitk_between(LB):
    """a small split-K launch with the in-kernel fix-up (tickets in the low slots of the same counter array)"""
    A, B = _rand(100, 3072, seed=90), _rand(96, 3072, seed=91)
    C = torch.empty(100, 96, device='cuda')
    LB.gemm(LB.NT, A, B, C, 100, 96, 3072, splits=4)
    _close(C, A.double() @ B.double().t())


@pytest.mark.parametrize('M,N,K', [(140000, 384, 96), (50000, 1536, 384), (131072 + 77, 96, 384), (70000, 384, 1536)])
def test_nt_bias_and_plain(M, N, K):
    LB = _LB()
    A, B, bias = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1), _rand(N, seed=3)
    ref = A.double() @ B.double().t()
    C = torch.full((M, N), float('nan'), device='cuda')
    LB.gemm(LB.NT, A, B, C, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
    _close(C, ref + bias.double())
    _splitk_between(LB)
    C2 = torch.full((M, N), float('nan'), device='cuda')
    LB.gemm(LB.NT, A, B, C2, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
    assert torch.equal(C, C2)
    C3 = torch.full((M, N), float('nan'), device='cuda')
    LB.gemm(LB.NT, A, B, C3, M, N, K)
    _close(C3, ref)
    torch.cuda.synchronize()
    _queue_is_zero(LB, C.device)


def test_nt_gelu_and_scale_residual_and_nn_dgrads():
    LB = _LB()
    M, N, K = 150000, 384, 96
    A, B, bias = _rand(M, K, seed=8), _rand(N, K, seed=9, scale=0.2), _rand(N, seed=10)
    hpre, act = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    LB.gemm(LB.NT, A, B, act, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=bias, aux_out=hpre)
    h64 = A.double() @ B.double().t() + bias.double()
    cdf = 0.5 * (1 + torch.erf(h64 / 2 ** 0.5))
    _close(act, h64 * cdf, tol=2e-4)
    _close(hpre, cdf + h64 * torch.exp(-0.5 * h64 * h64) / (2 * np.pi) ** 0.5, tol=2e-4)
    # FC2 + layer scale + stochastic depth + residual
    M2, N2, K2 = 150016, 96, 384
    A2, B2, b2 = _rand(M2, K2, seed=11), _rand(N2, K2, seed=12, scale=0.1), _rand(N2, seed=13)
    res, gamma = _rand(M2, N2, seed=14), _rand(N2, seed=15)
    rps = M2 // 4
    rs = torch.tensor([0.0, 1.0 / 0.9, 1.0, 1.0 / 0.9], device='cuda')
    y, out = torch.empty(M2, N2, device='cuda'), torch.empty(M2, N2, device='cuda')
    LB.gemm(LB.NT, A2, B2, out, M2, N2, K2, epilogue=LB.EPI_BIAS_SCALE_RES, bias=b2, aux_in=res, aux_out=y, gamma=gamma,
            rowscale=rs, rows_per_scale=rps)
    y64 = A2.double() @ B2.double().t() + b2.double()
    _close(y, y64)
    _close(out, res.double() + rs.double().repeat_interleave(rps)[:, None] * gamma.double()[None] * y64)
    # dgrad through GELU' with the bias gradient as a by-product, then the plain dgrad
    dY, W = _rand(M, K, seed=16), _rand(K, N, seed=17, scale=0.2)
    dH, db = torch.empty(M, N, device='cuda'), torch.empty(N, device='cuda')
    LB.gemm(LB.NN, dY, W, dH, M, N, K, epilogue=LB.EPI_GELU_BWD, aux_in=hpre, colsum_out=db)
    ref = (dY.double() @ W.double()) * hpre.double()
    _close(dH, ref, tol=2e-4)
    _close(db, ref.sum(0), tol=2e-4)
    W1 = _rand(N, K, seed=18, scale=0.1)
    dX = torch.empty(M, K, device='cuda')
    LB.gemm(LB.NN, dH, W1, dX, M, K, N)
    _close(dX, dH.double() @ W1.double())
    torch.cuda.synchronize()
    _queue_is_zero(LB, dX.device)


@pytest.mark.parametrize('E,counts', [(8, [30000, 0, 12900, 1, 128, 50000, 6400, 45007]),
                                      (16, [9000] * 15 + [3]), (4, [0, 0, 0, 150000])])
def test_grouped_ragged_experts(E, counts):
    """ragged expert segments: empty groups, one-row groups, surplus items at the end of the queue (the launch allots
    ceil(S / BM) + E row tiles)"""
    LB = _LB()
    C_, Hd = 96, 384
    S = sum(counts)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device='cuda')
    X, W1, b1 = _rand(S, C_, seed=20), _rand(E, Hd, C_, seed=21, scale=0.1), _rand(E, Hd, seed=22)
    hpre, act = torch.zeros(S, Hd, device='cuda'), torch.zeros(S, Hd, device='cuda')
    LB.gemm(LB.NT, X, W1, act, S, Hd, C_, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=hpre, offsets=offs, num_groups=E)
    o = offs.tolist()
    ref = torch.cat([X[o[e]:o[e + 1]].double() @ W1[e].double().t() + b1[e].double() for e in range(E)])
    _close(act, torch.nn.functional.gelu(ref), tol=2e-4)
    W2, b2 = _rand(E, C_, Hd, seed=25, scale=0.1), _rand(E, C_, seed=26)
    Y = torch.full((S, C_), float('nan'), device='cuda')
    LB.gemm(LB.NT, act, W2, Y, S, C_, Hd, epilogue=LB.EPI_BIAS, bias=b2, offsets=offs, num_groups=E)
    ref2 = torch.cat([act[o[e]:o[e + 1]].double() @ W2[e].double().t() + b2[e].double() for e in range(E)])
    _close(Y, ref2)
    dY = _rand(S, C_, seed=24)
    dH, db = torch.zeros(S, Hd, device='cuda'), torch.full((E, Hd), float('nan'), device='cuda')
    LB.gemm(LB.NN, dY, W2, dH, S, Hd, C_, epilogue=LB.EPI_GELU_BWD, aux_in=hpre, offsets=offs, num_groups=E, colsum_out=db)
    refh = torch.cat([dY[o[e]:o[e + 1]].double() @ W2[e].double() for e in range(E)]) * hpre.double()
    _close(dH, refh, tol=2e-4)
    refb = torch.stack([refh[o[e]:o[e + 1]].sum(0) for e in range(E)])
    _close(db, refb, tol=2e-4)
    dX = torch.full((S, C_), float('nan'), device='cuda')
    LB.gemm(LB.NN, dH, W1, dX, S, C_, Hd, offsets=offs, num_groups=E)
    refx = torch.cat([dH[o[e]:o[e + 1]].double() @ W1[e].double() for e in range(E)])
    _close(dX, refx)
    torch.cuda.synchronize()
    _queue_is_zero(LB, dX.device)


def test_amp_data_path_persistent_forms():
    """the four NT / NN combinations of the fp16 data path (gemm_h16_p.hip) past the persistence threshold"""
    LB = _LB()
    from sm3det_amd import amp
    h16 = torch.float16
    d = lambda t: t.double()  # noqa: E731
    S, C_, Hd, E = 150000, 192, 768, 8
    counts = [S // E + (17 if e % 2 else -17) for e in range(E)]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device='cuda')
    o = offs.tolist()
    with amp.autocast():
        X = _rand(S, C_, seed=30).to(h16)
        W1, b1 = _rand(E, Hd, C_, seed=31, scale=0.1), _rand(E, Hd, seed=32)
        act, hpre = torch.empty(S, Hd, device='cuda', dtype=h16), torch.empty(S, Hd, device='cuda', dtype=h16)
        LB.gemm(LB.NT, X, W1, act, S, Hd, C_, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=hpre, offsets=offs, num_groups=E)
        h64 = torch.cat([d(X[o[e]:o[e + 1]]) @ d(W1[e].half()).t() + d(b1[e]) for e in range(E)])
        cdf = 0.5 * (1 + torch.erf(h64 / 2 ** 0.5))
        _close(act, h64 * cdf, tol=1.5e-3)   # + the fp16 rounding of the stored output (2^-11)
        _close(hpre, cdf + h64 * torch.exp(-0.5 * h64 * h64) / (2 * np.pi) ** 0.5, tol=1.5e-3)
        W2, b2 = _rand(E, C_, Hd, seed=33, scale=0.1), _rand(E, C_, seed=34)
        Y = torch.full((S, C_), float('nan'), device='cuda')
        LB.gemm(LB.NT, act, W2, Y, S, C_, Hd, epilogue=LB.EPI_BIAS, bias=b2, offsets=offs, num_groups=E)
        _close(Y, torch.cat([d(act[o[e]:o[e + 1]]) @ d(W2[e].half()).t() + d(b2[e]) for e in range(E)]), tol=5e-5)
        dY = _rand(S, C_, seed=35)
        dH = torch.empty(S, Hd, device='cuda', dtype=h16)
        db = torch.empty(E, Hd, device='cuda')
        LB.gemm(LB.NN, dY, W2, dH, S, Hd, C_, epilogue=LB.EPI_GELU_BWD, aux_in=hpre, offsets=offs, num_groups=E, colsum_out=db)
        refh = torch.cat([d(dY[o[e]:o[e + 1]].half()) @ d(W2[e].half()) for e in range(E)]) * d(hpre)
        _close(dH, refh, tol=1.5e-3)
        dX = torch.full((S, C_), float('nan'), device='cuda')
        LB.gemm(LB.NN, dH, W1, dX, S, C_, Hd, offsets=offs, num_groups=E)
        _close(dX, torch.cat([d(dH[o[e]:o[e + 1]]) @ d(W1[e].half()) for e in range(E)]), tol=5e-5)
        # dense FC2 with layer scale + residual (A16)
        M2 = 140000
        A2 = _rand(M2, Hd, seed=36).to(h16)
        Wd, bd, res, gamma = _rand(C_, Hd, seed=37, scale=0.05), _rand(C_, seed=38), _rand(M2, C_, seed=39), _rand(C_, seed=40)
        y, out = torch.empty(M2, C_, device='cuda'), torch.empty(M2, C_, device='cuda')
        LB.gemm(LB.NT, A2, Wd, out, M2, C_, Hd, epilogue=LB.EPI_BIAS_SCALE_RES, bias=bd, aux_in=res, aux_out=y, gamma=gamma)
        y64 = d(A2) @ d(Wd.half()).t() + d(bd)
        _close(y, y64, tol=5e-5)
        _close(out, d(res) + d(gamma)[None] * y64, tol=5e-5)
    torch.cuda.synchronize()
    _queue_is_zero(LB, dX.device)


def test_persistent_launches_replay_from_a_hipgraph():
    """the queue words are zero on entry and on exit, so a captured launch replays"""
    LB = _LB()
    M, N, K = 140000, 384, 96
    A, B, bias = _rand(M, K, seed=50), _rand(N, K, seed=51, scale=0.1), _rand(N, seed=52)
    C = torch.empty(M, N, device='cuda')
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        LB.gemm(LB.NT, A, B, C, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)  # eager warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        LB.gemm(LB.NT, A, B, C, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
    ref = A.double() @ B.double().t() + bias.double()
    for _ in range(3):
        C.fill_(float('nan'))
        g.replay()
        torch.cuda.synchronize()
        _close(C, ref)
