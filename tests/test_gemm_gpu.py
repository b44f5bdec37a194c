"""GPU tier: fp32 MFMA GEMM family vs a plain PyTorch fp32/fp64 reference of the same contraction.
Tolerance: |d| <= 2e-5 * sum|a||b| scale (f32 FMA chain, K up to 3072) -- asserted as rtol 1e-4 on fp64 reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods():
    from sm3det_amd import _lib_backbone as LB
    return LB


@pytest.fixture(autouse=True, params=['bf16x3', 'f32'])
def _arith(request):
    """every test of this module runs in both arithmetics of the fp32 GEMM family: the bf16x3 form (three exact bf16 pieces
    per operand element, six v_mfma_f32_32x32x16_bf16 products, the default) and the native v_mfma_f32_32x32x2_f32 form"""
    LB = _mods()
    old, LB.ARITH32 = LB.ARITH32, {'bf16x3': 2, 'f32': 0}[request.param]
    yield request.param
    LB.ARITH32 = old


def _rand(*shape, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.randn(*shape, generator=g).cuda()


def _close(a, ref64, tol=1e-4):
    err = (a.double() - ref64).abs().max().item()
    scale = ref64.abs().max().item() + 1e-12
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize('M,N,K', [(1000, 384, 96), (4096, 96, 384), (513, 768, 192), (128, 128, 64), (77, 3072, 768)])
def test_nt_bias(M, N, K):
    LB = _mods()
    A, B, bias = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
    C = torch.empty(M, N, device='cuda')
    LB.gemm(LB.NT, A, B, C, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
    _close(C, A.double() @ B.double().t() + bias.double())
    C2 = torch.full((M, N), float('nan'), device='cuda')
    LB.gemm(LB.NT, A, B, C2, M, N, K)
    _close(C2, A.double() @ B.double().t())


def test_nt_asymmetric_identity():
    """transpose-detecting check: A = I, asymmetric B."""
    LB = _mods()
    K = 128
    A = torch.eye(K, device='cuda')
    B = (torch.arange(256 * K, device='cuda', dtype=torch.float32).reshape(256, K) % 97) * 0.25
    C = torch.empty(K, 256, device='cuda')
    LB.gemm(LB.NT, A, B, C, K, 256, K)
    assert torch.equal(C, B.t().contiguous())


@pytest.mark.parametrize('M,N,K', [(1000, 96, 384), (2048, 384, 96), (300, 192, 768)])
def test_nn(M, N, K):
    LB = _mods()
    A, B = _rand(M, K, seed=4), _rand(K, N, seed=5)
    C = torch.empty(M, N, device='cuda')
    LB.gemm(LB.NN, A, B, C, M, N, K)
    _close(C, A.double() @ B.double())


@pytest.mark.parametrize('Kt,M,N,splits', [(5000, 96, 384, 8), (4096, 384, 96, 4), (1000, 768, 192, 1), (70, 128, 128, 3)])
def test_tn_splitk(Kt, M, N, splits):
    LB = _mods()
    A, B = _rand(Kt, M, seed=6), _rand(Kt, N, seed=7)
    C = torch.empty(M, N, device='cuda')
    LB.gemm(LB.TN, A, B, C, M, N, Kt, splits=splits)
    _close(C, A.double().t() @ B.double())


def test_epilogues():
    LB = _mods()
    M, N, K = 700, 384, 96
    A, B, bias = _rand(M, K, seed=8), _rand(N, K, seed=9) * 0.2, _rand(N, seed=10)
    hpre = torch.empty(M, N, device='cuda')
    act = torch.empty(M, N, device='cuda')
    LB.gemm(LB.NT, A, B, act, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=bias, aux_out=hpre)
    h64 = (A.double() @ B.double().t() + bias.double()).requires_grad_(True)
    a64 = torch.nn.functional.gelu(h64)
    a64.sum().backward()
    _close(act, a64.detach(), tol=2e-4)
    _close(hpre, h64.grad, tol=2e-4)  # aux_out = gelu'(h)
    # scale + residual (+ per-image stochastic-depth scale)
    M2, N2, K2 = 512, 96, 384
    A2, B2, b2 = _rand(M2, K2, seed=11), _rand(N2, K2, seed=12) * 0.1, _rand(N2, seed=13)
    res, gamma = _rand(M2, N2, seed=14), _rand(N2, seed=15)
    rs = torch.tensor([0.0, 1.0 / 0.9], device='cuda')
    y = torch.empty(M2, N2, device='cuda')
    out = torch.empty(M2, N2, device='cuda')
    LB.gemm(LB.NT, A2, B2, out, M2, N2, K2, epilogue=LB.EPI_BIAS_SCALE_RES, bias=b2, aux_in=res, aux_out=y,
            gamma=gamma, rowscale=rs, rows_per_scale=256)
    y64 = A2.double() @ B2.double().t() + b2.double()
    _close(y, y64)
    sc = rs.double().repeat_interleave(256)[:, None] * gamma.double()[None]
    _close(out, res.double() + sc * y64)
    # dgrad through GELU
    dY, W = _rand(M, K, seed=16), _rand(K, N, seed=17) * 0.2
    dH = torch.empty(M, N, device='cuda')
    db = torch.empty(N, device='cuda')
    LB.gemm(LB.NN, dY, W, dH, M, N, K, epilogue=LB.EPI_GELU_BWD, aux_in=hpre, colsum_out=db)
    ref = (dY.double() @ W.double()) * h64.grad
    _close(dH, ref, tol=2e-4)
    _close(db, ref.sum(0), tol=2e-4)


@pytest.mark.parametrize('E,counts', [(8, [300, 0, 129, 1, 128, 500, 64, 7]), (16, [40] * 16), (4, [0, 0, 0, 1000])])
def test_grouped(E, counts):
    LB = _mods()
    C_, Hd = 96, 384
    S = sum(counts)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device='cuda')
    X = _rand(S, C_, seed=20)
    W1 = _rand(E, Hd, C_, seed=21) * 0.1
    b1 = _rand(E, Hd, seed=22)
    hpre = torch.zeros(S, Hd, device='cuda')
    act = torch.zeros(S, Hd, device='cuda')
    LB.gemm(LB.NT, X, W1, act, S, Hd, C_, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=hpre, offsets=offs,
            num_groups=E)
    ref = torch.cat([X[offs[e]:offs[e + 1]].double() @ W1[e].double().t() + b1[e].double() for e in range(E)])
    _close(act, torch.nn.functional.gelu(ref), tol=2e-4)
    # grouped dgrad through GELU with fused per-expert bias gradient
    dY = _rand(S, C_, seed=24)
    W2 = _rand(E, C_, Hd, seed=25) * 0.1
    dHg, dbg = torch.zeros(S, Hd, device='cuda'), torch.full((E, Hd), float('nan'), device='cuda')
    LB.gemm(LB.NN, dY, W2, dHg, S, Hd, C_, epilogue=LB.EPI_GELU_BWD, aux_in=hpre, offsets=offs, num_groups=E,
            colsum_out=dbg)
    refg = torch.cat([dY[offs[e]:offs[e + 1]].double() @ W2[e].double() for e in range(E)]) * hpre.double()
    _close(dHg, refg)
    _close(dbg, torch.stack([refg[offs[e]:offs[e + 1]].sum(0) for e in range(E)]))
    # grouped dgrad NN: dX = dH @ W1  (W1[e] is (Hd,C) = K x N)
    dH = _rand(S, Hd, seed=23)
    dX = torch.zeros(S, C_, device='cuda')
    LB.gemm(LB.NN, dH, W1, dX, S, C_, Hd, offsets=offs, num_groups=E)
    ref = torch.cat([dH[offs[e]:offs[e + 1]].double() @ W1[e].double() for e in range(E)])
    _close(dX, ref)
    # grouped wgrad TN: dW1[e] = dH_e^T @ X_e  (every expert gets a, possibly zero, gradient)
    dW = torch.full((E, Hd, C_), float('nan'), device='cuda')
    LB.gemm(LB.TN, dH, X, dW, Hd, C_, S, offsets=offs, num_groups=E, splits=4)
    ref = torch.stack([dH[offs[e]:offs[e + 1]].double().t() @ X[offs[e]:offs[e + 1]].double() for e in range(E)])
    _close(dW, ref)
    # grouped bias grad
    db = torch.empty(E, Hd, device='cuda')
    LB.colsum(dH, S, Hd, db, offsets=offs, num_groups=E)
    ref = torch.stack([dH[offs[e]:offs[e + 1]].double().sum(0) for e in range(E)])
    _close(db, ref)


def test_gemm_throughput_report():
    """Not a pass/fail perf gate; prints achieved fp32 MFMA TFLOP/s for the stage-2 expert shape."""
    LB = _mods()
    M, N, K = 16384, 1536, 384
    A, B = _rand(M, K, seed=30), _rand(N, K, seed=31)
    C = torch.empty(M, N, device='cuda')
    for _ in range(3):
        LB.gemm(LB.NT, A, B, C, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        LB.gemm(LB.NT, A, B, C, M, N, K)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'\ngemm NT {M}x{N}x{K}: {ms * 1e3:.1f} us, {2 * M * N * K / ms / 1e9:.1f} TFLOP/s fp32 (peak 157.3)')
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        torch.matmul(A, B.t(), out=C)
    t1.record()
    torch.cuda.synchronize()
    ms2 = t0.elapsed_time(t1) / 20
    print(f'torch.matmul (rocBLAS/hipBLASLt) same shape: {ms2 * 1e3:.1f} us, {2 * M * N * K / ms2 / 1e9:.1f} TFLOP/s')


# ------------------------------------------------------------------------------------------ tile shapes / split-K fix-up
def _tune(tile=None, bk=None, splits=0, fixup=False):
    """sm3_gemm_desc.tuning word (include/sm3det_hip.h)"""
    return ((tile + 1) if tile is not None else 0) | ({None: 0, 16: 1, 32: 2}[bk] << 4) | (splits << 8) | (int(fixup) << 16)


class _tuned:
    def __init__(self, **kw):
        self.word = _tune(**kw)

    def __enter__(self):
        LB = _mods()
        self.old, LB.TUNING = LB.TUNING, self.word

    def __exit__(self, *exc):
        _mods().TUNING = self.old


@pytest.mark.parametrize('mode,tile,bk', [
    ('nt', 0, 16), ('nt', 0, 32), ('nt', 1, 16), ('nt', 1, 32), ('nt', 3, 16), ('nt', 5, 16), ('nt', 5, 32),
    ('nn', 0, 16), ('nn', 1, 16), ('nn', 1, 32), ('nn', 3, 16), ('nn', 5, 32),
    ('tn', 0, 16), ('tn', 0, 32), ('tn', 1, 16), ('tn', 2, 16), ('tn', 2, 32), ('tn', 3, 16), ('tn', 4, 16)])
@pytest.mark.parametrize('splits', [1, 3])
def test_every_tile_shape_and_k_step_with_and_without_split_k_fixup(mode, tile, bk, splits):
    """every instantiated (tile, k-step) of each mode on a shape with ragged edges in M and N, without split-K and with
    the in-kernel last-arriver fix-up (3 slices), against fp64."""
    LB = _mods()
    M, N, K = 333, 292, 448  # M, N not multiples of any tile edge; K = 14 k-tiles of 32
    with _tuned(tile=tile, bk=bk, splits=splits):
        if mode == 'nt':
            A, B, bias = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
            C = torch.full((M, N), float('nan'), device='cuda')
            LB.gemm(LB.NT, A, B, C, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
            _close(C, A.double() @ B.double().t() + bias.double())
        elif mode == 'nn':
            A, B = _rand(M, K, seed=4), _rand(K, N, seed=5)
            C = torch.full((M, N), float('nan'), device='cuda')
            LB.gemm(LB.NN, A, B, C, M, N, K)
            _close(C, A.double() @ B.double())
        else:
            Kt, Mo, No = 1111, 292, 332
            A, B = _rand(Kt, Mo, seed=6), _rand(Kt, No, seed=7)
            for fix in (False, True):
                C = torch.full((Mo, No), float('nan'), device='cuda')
                LB.TUNING = _tune(tile=tile, bk=bk, splits=splits, fixup=fix)
                LB.gemm(LB.TN, A, B, C, Mo, No, Kt)
                _close(C, A.double().t() @ B.double())
    # the ticket counters are left zeroed
    assert int(LB.gemm_counters(torch.device('cuda', torch.cuda.current_device())).abs().sum()) == 0


def test_default_tile_choice_for_convnext_widths():
    """N = 96 / 192 / 288 (ConvNeXt-T stage widths, gate projection) take the exact-width tiles by default; results
    still equal fp64 and the GELU / scale-residual / GELU-backward epilogues work on them."""
    LB = _mods()
    for M, N, K in [(1000, 96, 384), (777, 192, 768), (515, 288, 768), (300, 96, 3072)]:
        A, B, bias = _rand(M, K, seed=N), _rand(N, K, seed=N + 1) * 0.1, _rand(N, seed=N + 2)
        res, gamma = _rand(M, N, seed=N + 3), _rand(N, seed=N + 4)
        y, out = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
        LB.gemm(LB.NT, A, B, out, M, N, K, epilogue=LB.EPI_BIAS_SCALE_RES, bias=bias, aux_in=res, aux_out=y,
                gamma=gamma)
        y64 = A.double() @ B.double().t() + bias.double()
        _close(y, y64)
        _close(out, res.double() + gamma.double()[None] * y64)
        # NN with GELU-backward epilogue + fused column sums on the same widths
        dY, W, hp = _rand(M, K, seed=N + 5), _rand(K, N, seed=N + 6) * 0.1, _rand(M, N, seed=N + 7)
        dH, db = torch.empty(M, N, device='cuda'), torch.empty(N, device='cuda')
        LB.gemm(LB.NN, dY, W, dH, M, N, K, epilogue=LB.EPI_GELU_BWD, aux_in=hp, colsum_out=db)
        ref = (dY.double() @ W.double()) * hp.double()
        _close(dH, ref)
        _close(db, ref.sum(0), tol=2e-4)
    for Kt, M, N in [(5000, 96, 384), (5000, 384, 96), (3000, 192, 768), (3000, 768, 192), (4000, 224, 384)]:
        A, B = _rand(Kt, M, seed=M), _rand(Kt, N, seed=N)
        C = torch.empty(M, N, device='cuda')
        LB.gemm(LB.TN, A, B, C, M, N, Kt)
        _close(C, A.double().t() @ B.double())


@pytest.mark.parametrize('splits', [2, 5, 8])
def test_split_k_fixup_with_every_epilogue_and_ragged_groups(splits):
    """NT / NN launches sliced along K with the last-arriver fix-up: the epilogue (bias+GELU with its second output,
    GELU-backward with fused per-expert column sums) runs once, in the last block, on the summed accumulators; ragged
    expert segments incl. empty ones; repeated launches reuse the self-resetting counters."""
    LB = _mods()
    E, counts = 8, [300, 0, 129, 1, 128, 500, 64, 7]
    C_, Hd = 384, 1536
    S = sum(counts)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device='cuda')
    X, W1, b1 = _rand(S, C_, seed=20), _rand(E, Hd, C_, seed=21) * 0.05, _rand(E, Hd, seed=22)
    ref = torch.cat([X[offs[e]:offs[e + 1]].double() @ W1[e].double().t() + b1[e].double() for e in range(E)])
    for rep in range(2):
        hpre, act = torch.zeros(S, Hd, device='cuda'), torch.zeros(S, Hd, device='cuda')
        with _tuned(splits=splits):
            LB.gemm(LB.NT, X, W1, act, S, Hd, C_, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=hpre, offsets=offs,
                    num_groups=E)
        _close(act, torch.nn.functional.gelu(ref), tol=2e-4)
    dY, W2 = _rand(S, C_, seed=24), _rand(E, C_, Hd, seed=25) * 0.05
    dHg, dbg = torch.zeros(S, Hd, device='cuda'), torch.full((E, Hd), float('nan'), device='cuda')
    with _tuned(splits=splits):
        LB.gemm(LB.NN, dY, W2, dHg, S, Hd, C_, epilogue=LB.EPI_GELU_BWD, aux_in=hpre, offsets=offs, num_groups=E,
                colsum_out=dbg)
    refg = torch.cat([dY[offs[e]:offs[e + 1]].double() @ W2[e].double() for e in range(E)]) * hpre.double()
    _close(dHg, refg)
    _close(dbg, torch.stack([refg[offs[e]:offs[e + 1]].sum(0) for e in range(E)]), tol=2e-4)
    # grouped TN: the second reduce pass (default) and the in-kernel fix-up give the same sums
    dH = _rand(S, Hd, seed=23)
    outs = []
    for fix in (False, True):
        dW = torch.full((E, Hd, C_), float('nan'), device='cuda')
        with _tuned(splits=splits, fixup=fix):
            LB.gemm(LB.TN, dH, X, dW, Hd, C_, S, offsets=offs, num_groups=E)
        outs.append(dW)
    refw = torch.stack([dH[offs[e]:offs[e + 1]].double().t() @ X[offs[e]:offs[e + 1]].double() for e in range(E)])
    _close(outs[0], refw)
    _close(outs[1], refw)
    assert int(LB.gemm_counters(torch.device('cuda', torch.cuda.current_device())).abs().sum()) == 0


def test_automatic_split_k_on_few_tile_long_k_shapes_is_deterministic():
    """the stage-3 dense FFN shapes (96 / 384 output tiles on 256 CUs) slice K automatically; two launches give
    bit-identical results (slices are summed in slice order, whichever block arrives last)."""
    LB = _mods()
    for M, N, K in [(2048, 768, 3072), (2048, 3072, 768), (8192, 384, 1536)]:
        A, B, bias = _rand(M, K, seed=1), _rand(N, K, seed=2) * 0.05, _rand(N, seed=3)
        C1, C2 = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
        LB.gemm(LB.NT, A, B, C1, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
        LB.gemm(LB.NT, A, B, C2, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
        assert torch.equal(C1, C2)
        _close(C1, A.double() @ B.double().t() + bias.double())


@pytest.mark.parametrize('splits', [9, 28, 341])
def test_tn_many_slices_separate_reduce(splits):
    LB = _mods()
    Kt, M, N = 131072 // 4, 96, 384
    A, B = _rand(Kt, M, seed=6), _rand(Kt, N, seed=7)
    C = torch.empty(M, N, device='cuda')
    LB.gemm(LB.TN, A, B, C, M, N, Kt, splits=splits)
    _close(C, A.double().t() @ B.double())


@pytest.mark.parametrize('tile,bk', [(0, 16), (0, 32), (1, 16), (1, 32), (3, 16), (5, 16), (5, 32)])
def test_every_epilogue_kind_on_every_tile(tile, bk):
    """the LDS-staged (row-contiguous, full-cache-line) epilogue against fp64 for every epilogue kind on every NT / NN
    tile shape, ragged M / N, grouped rows, with and without the split-K fix-up in front of it"""
    LB = _mods()
    E, counts = 4, [150, 0, 77, 130]
    S = sum(counts)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device='cuda')
    C_, Hd = 192, 292
    X, W1, b1 = _rand(S, C_, seed=20), _rand(E, Hd, C_, seed=21) * 0.1, _rand(E, Hd, seed=22)
    ref = torch.cat([X[offs[e]:offs[e + 1]].double() @ W1[e].double().t() + b1[e].double() for e in range(E)])
    for splits in (1, 2):
        hpre, act = torch.zeros(S, Hd, device='cuda'), torch.zeros(S, Hd, device='cuda')
        with _tuned(tile=tile, bk=bk, splits=splits):
            LB.gemm(LB.NT, X, W1, act, S, Hd, C_, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=hpre, offsets=offs,
                    num_groups=E)
        _close(act, torch.nn.functional.gelu(ref), tol=2e-4)
        dY, W2 = _rand(S, C_, seed=24), _rand(E, C_, Hd, seed=25) * 0.1
        dHg, dbg = torch.zeros(S, Hd, device='cuda'), torch.full((E, Hd), float('nan'), device='cuda')
        with _tuned(tile=tile, bk=bk, splits=splits):
            LB.gemm(LB.NN, dY, W2, dHg, S, Hd, C_, epilogue=LB.EPI_GELU_BWD, aux_in=hpre, offsets=offs, num_groups=E,
                    colsum_out=dbg)
        refg = torch.cat([dY[offs[e]:offs[e + 1]].double() @ W2[e].double() for e in range(E)]) * hpre.double()
        _close(dHg, refg)
        _close(dbg, torch.stack([refg[offs[e]:offs[e + 1]].sum(0) for e in range(E)]), tol=2e-4)
    M, N, K = 333, 292, 448
    A, B, bias = _rand(M, K, seed=1), _rand(N, K, seed=2) * 0.1, _rand(N, seed=3)
    res, gamma = _rand(M, N, seed=4), _rand(N, seed=5)
    rs = torch.tensor([0.0, 1.0 / 0.9, 1.0], device='cuda')
    y, out = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    with _tuned(tile=tile, bk=bk):
        LB.gemm(LB.NT, A, B, out, M, N, K, epilogue=LB.EPI_BIAS_SCALE_RES, bias=bias, aux_in=res, aux_out=y,
                gamma=gamma, rowscale=rs, rows_per_scale=128)
        y64 = A.double() @ B.double().t() + bias.double()
        _close(y, y64)
        sc = rs.double().repeat_interleave(128)[:M, None] * gamma.double()[None]
        _close(out, res.double() + sc * y64)
        for epi, f in ((LB.EPI_NONE, lambda t: t - bias.double()), (LB.EPI_BIAS, lambda t: t),
                       (LB.EPI_BIAS_RELU, lambda t: t.clamp_min(0))):
            C = torch.full((M, N), float('nan'), device='cuda')
            LB.gemm(LB.NT, A, B, C, M, N, K, epilogue=epi, bias=bias if epi != LB.EPI_NONE else None)
            _close(C, f(y64))


@pytest.mark.parametrize('splits', [0, 1, 3, 16])
@pytest.mark.parametrize('E,counts,M,N', [(1, [5000], 96, 384), (1, [777], 384, 96), (8, [300, 0, 129, 1, 128, 500, 64, 7], 384, 96),
                                          (4, [0, 0, 0, 1000], 224, 384), (1, [4096], 32, 256)])
def test_tn_column_sums_of_a_as_a_by_product(E, counts, M, N, splits):
    """TN with colsum_out: dW[g] = A_g^T B_g and db[g] = column sums of A_g from the same launch (bias gradient next to
    the weight gradient), any slice count, ragged / empty groups, tile edges (M = 224, 32)."""
    LB = _mods()
    S = sum(counts)
    A, B = _rand(S, M, seed=31), _rand(S, N, seed=32)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device='cuda') if E > 1 else None
    dW = torch.full((E, M, N), float('nan'), device='cuda')
    db = torch.full((E, M), float('nan'), device='cuda')
    kw = dict(offsets=offs, num_groups=E) if E > 1 else {}
    LB.gemm(LB.TN, A, B, dW, M, N, S, colsum_out=db, splits=splits, **kw)
    bounds = np.concatenate([[0], np.cumsum(counts)])
    refw = torch.stack([A[bounds[e]:bounds[e + 1]].double().t() @ B[bounds[e]:bounds[e + 1]].double() for e in range(E)])
    refb = torch.stack([A[bounds[e]:bounds[e + 1]].double().sum(0) for e in range(E)])
    _close(dW, refw)
    assert (db.double() - refb).abs().max().item() <= 1e-4 * (refb.abs().max().item() + 1e-12) + 1e-6


# ------------------------------------------------------------------------------------------ bf16x3 vs native fp32 MFMA
_STEP_SHAPES = [  # (mode, M, N, K): the contraction shapes of the training step (rows capped: the error does not depend on M)
    ('nt', 8192, 384, 96), ('nt', 8192, 96, 384), ('nn', 8192, 384, 96), ('nn', 8192, 96, 384),
    ('tn', 384, 96, 131072), ('tn', 96, 384, 131072),
    ('nt', 8192, 768, 192), ('nt', 8192, 192, 768), ('nn', 8192, 768, 192), ('nn', 8192, 192, 768),
    ('tn', 768, 192, 32768), ('tn', 192, 768, 32768),
    ('nt', 8192, 1536, 384), ('nt', 8192, 384, 1536), ('nn', 8192, 1536, 384), ('nn', 8192, 384, 1536),
    ('tn', 1536, 384, 8192), ('tn', 384, 1536, 8192),
    ('nt', 2048, 3072, 768), ('nt', 2048, 768, 3072), ('nn', 2048, 3072, 768), ('nn', 2048, 768, 3072),
    ('tn', 3072, 768, 2048), ('tn', 768, 3072, 2048),
    ('nt', 8192, 224, 384), ('nn', 8192, 384, 224), ('tn', 224, 384, 8192), ('nt', 8192, 128, 192), ('nt', 2048, 288, 768)]


@pytest.mark.parametrize('mode,M,N,K', _STEP_SHAPES)
def test_bf16x3_error_is_that_of_the_native_fp32_form(mode, M, N, K, _arith):
    """Guardrail of the bf16x3 form: on every contraction shape of the training step its error against the fp64 product of
    the SAME fp32 operands (max-norm and rms) is at most 1.5 x the native fp32 MFMA kernel's -- i.e. it is an fp32
    evaluation, not a reduced-precision one (24 operand bits are kept: three bf16 pieces sum to the fp32 value exactly)."""
    if _arith != 'bf16x3':
        pytest.skip('comparison test: runs once')
    LB = _mods()
    md = dict(nt=LB.NT, nn=LB.NN, tn=LB.TN)[mode]
    if mode == 'nt':
        A, B = _rand(M, K, seed=11), _rand(N, K, seed=12) * 0.05
        ref = A.double() @ B.double().t()
    elif mode == 'nn':
        A, B = _rand(M, K, seed=13), _rand(K, N, seed=14) * 0.05
        ref = A.double() @ B.double()
    else:
        A, B = _rand(K, M, seed=15), _rand(K, N, seed=16)
        ref = A.double().t() @ B.double()
    err = {}
    for name, ar in (('f32', 0), ('bf16x3', 2)):
        LB.ARITH32 = ar
        C = torch.full((M, N), float('nan'), device='cuda')
        LB.gemm(md, A, B, C, M, N, K)
        d = C.double() - ref
        err[name] = (d.abs().max().item(), d.pow(2).mean().sqrt().item())
    LB.ARITH32 = 2
    assert err['bf16x3'][0] <= 1.5 * err['f32'][0], err
    assert err['bf16x3'][1] <= 1.5 * err['f32'][1], err


def test_single_term_products_carry_all_24_operand_bits():
    """One non-zero k per row: C[i][j] = a_i * b_j.  Both arithmetics must return the fp32 product to within two ulp
    (|err| <= 2^-22 |ab|: the six-product form measured 1.03 ulp, the native form half an ulp); one bf16 piece would give 2^-8, two pieces 2^-16.  Operands with fully populated mantissas."""
    LB = _mods()
    old = LB.ARITH32
    try:
        K = 32
        A = torch.zeros(256, K, device='cuda')
        B = torch.zeros(128, K, device='cuda')
        A[:, 5] = _rand(256, seed=41) * 3.0
        B[:, 5] = _rand(128, seed=42) * 0.01
        ref = A[:, 5].double()[:, None] * B[:, 5].double()[None, :]
        for ar in (0, 2):
            LB.ARITH32 = ar
            C = torch.full((256, 128), float('nan'), device='cuda')
            LB.gemm(LB.NT, A, B, C, 256, 128, K)
            rel = ((C.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
            assert rel <= 2.0 ** -22, (ar, rel)
    finally:
        LB.ARITH32 = old


# ------------------------------------------------------------------------------------------ bf16x3: the edges of the split
def _b3_vs_native(A, B):
    """(max relative-to-scale error of the bf16x3 form, of the native form) for C = A . B^T against the fp64 product"""
    LB = _mods()
    ref = A.double() @ B.double().t()
    out = []
    old = LB.ARITH32
    try:
        for ar in (2, 0):
            LB.ARITH32 = ar
            C = torch.full((A.shape[0], B.shape[0]), float('nan'), device='cuda')
            LB.gemm(LB.NT, A, B, C, A.shape[0], B.shape[0], A.shape[1])
            out.append(((C.double() - ref).abs().max() / ref.abs().max()).item())
    finally:
        LB.ARITH32 = old
    return out


@pytest.mark.parametrize('ea,eb', [(100, -100), (-100, 100), (60, 60), (-60, -60), (-100, 60)])
def test_bf16x3_extreme_but_normal_magnitudes(ea, eb):
    """bf16 has fp32's exponent range, so the three-piece split needs no scaling: with operands scaled by 2^+-100 (every
    piece still a NORMAL bf16: the lowest piece of x sits ~2^-17 below x, i.e. above 2^-126 for |x| >= 2^-109) the error
    is that of the unscaled problem -- the same 1.5 x native guardrail as on the step's shapes."""
    A = _rand(512, 384, seed=51) * 2.0 ** ea
    B = _rand(256, 384, seed=52) * 2.0 ** eb
    b3, nat = _b3_vs_native(A, B)
    assert b3 <= 1.5 * nat, (b3, nat)


def test_bf16x3_subnormal_low_pieces_degrade_gracefully():
    """Below |x| ~ 2^-110 the second and third pieces of x (2^-8 and 2^-16 below it) are bf16 SUBNORMALS or underflow to
    zero, so the operand keeps fewer than 24 bits: the result is still correct to the leading piece and a half (measured
    and pinned here: relative error <= 2^-7; the native fp32 form keeps ~2^-22).  Documented in DESIGN.md section 4; no tensor
    of the training step is near this range (activations and weights are O(1e-3..1e2))."""
    A = _rand(512, 384, seed=53) * 2.0 ** -120
    B = _rand(256, 384, seed=54) * 2.0 ** 100
    b3, nat = _b3_vs_native(A, B)
    assert nat <= 1e-5
    assert b3 <= 2.0 ** -7, b3


def test_bf16x3_non_finite_and_overflowing_operands():
    """What the default arithmetic does with operands outside the split's domain -- pinned so that a change is noticed:
    (i) an inf or NaN operand element gives NaN in every output it touches (the split computes inf - inf; the native form
    would give +-inf for an inf operand) and leaves all other outputs exact; (ii) a FINITE operand above the largest bf16
    (3.3895e38, the top 0.4 % of the fp32 range) rounds its first piece to bf16-inf and gives a non-finite output, where the
    native form stays finite.  `SM3_GEMM_ARITH=f32` / `sm3_gemm_desc.compute = 0` selects the native form for such data."""
    LB = _mods()
    old = LB.ARITH32
    try:
        LB.ARITH32 = 2
        A, B = _rand(256, 96, seed=55), _rand(128, 96, seed=56)
        ref = (A.double() @ B.double().t())
        for bad in (float('inf'), float('-inf'), float('nan')):
            A2 = A.clone()
            A2[7, 13] = bad
            C = torch.empty(256, 128, device='cuda')
            LB.gemm(LB.NT, A2, B, C, 256, 128, 96)
            assert torch.isnan(C[7]).all(), bad
            rest = torch.cat([C[:7], C[8:]]).double()
            refr = torch.cat([ref[:7], ref[8:]])
            assert torch.isfinite(rest).all() and ((rest - refr).abs().max() / refr.abs().max()).item() < 1e-5
        A3 = torch.zeros(256, 96, device='cuda')
        B3 = torch.zeros(128, 96, device='cuda')
        A3[:, 0] = 3.4e38      # finite in fp32, above the largest finite bf16
        B3[:, 0] = 2.0 ** -10  # the exact product 3.3e35 is finite
        C = torch.empty(256, 128, device='cuda')
        LB.gemm(LB.NT, A3, B3, C, 256, 128, 96)
        assert not torch.isfinite(C).any()
        LB.ARITH32 = 0
        LB.gemm(LB.NT, A3, B3, C, 256, 128, 96)
        assert torch.isfinite(C).all() and abs(C[0, 0].item() / (3.4e38 * 2.0 ** -10) - 1) < 1e-6
    finally:
        LB.ARITH32 = old
