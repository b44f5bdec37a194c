"""GPU tier: GFL head conv towers (sm3det_amd/gfl_head.py: implicit-GEMM 3x3 convs + GroupNorm/ReLU kernels) against
the plain-torch restatement oracle/gfl_oracle.py (mmdet semantics -- PARITY UNPINNED, see that file) and GroupNorm
against torch.nn.functional.group_norm in fp64.  Tolerances: forward 1e-4, gradients 1e-3 (max-norm relative)."""
import pytest
import torch
import torch.nn.functional as F

from tests.moe_common import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,H,W,C,G,relu', [(2, 16, 24, 256, 32, True), (1, 8, 8, 256, 32, False), (3, 5, 7, 96, 8, True),
                                            (1, 64, 64, 256, 32, True), (2, 3, 3, 512, 32, True)])
def test_groupnorm_relu_fwd_bwd_vs_torch_fp64(B, H, W, C, G, relu):
    from sm3det_amd.gfl_head import group_norm_relu
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, H, W, C, generator=g) * 2 + 0.5).cuda().requires_grad_(True)
    w = (torch.rand(C, generator=g) + 0.5).cuda().requires_grad_(True)
    b = torch.randn(C, generator=g).cuda().requires_grad_(True)
    y = group_norm_relu(x, w, b, G, 1e-5, relu)
    go = torch.randn(B, H, W, C, generator=g).cuda()
    y.backward(go)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = F.group_norm(xr.permute(0, 3, 1, 2), G, wr, br, 1e-5)
    if relu:
        yr = F.relu(yr)
    yr = yr.permute(0, 2, 3, 1)
    yr.backward(go.double())
    assert rel_err(y, yr) < 1e-5
    assert rel_err(x.grad, xr.grad) < 1e-4 and rel_err(w.grad, wr.grad) < 1e-4 and rel_err(b.grad, br.grad) < 1e-4


def test_gfl_head_config_shapes_state_dict_and_values_vs_oracle():
    from oracle import gfl_oracle as GO
    from sm3det_amd.gfl_head import GFLHead
    torch.manual_seed(0)
    head = GFLHead(num_classes=26, in_channels=256, stacked_convs=4, feat_channels=256,
                   anchor_generator=dict(type='AnchorGenerator', ratios=[1.0], octave_base_scale=8,
                                         scales_per_octave=1, strides=[8, 16, 32, 64, 128]),
                   loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=1.0),
                   loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25), reg_max=16,
                   loss_bbox=dict(type='GIoULoss', loss_weight=2.0))  # local_configs/main_SM3Det.py:29-48
    sd = head.state_dict()
    assert sd['cls_convs.0.conv.weight'].shape == (256, 256, 3, 3) and 'cls_convs.0.conv.bias' not in sd
    assert sd['cls_convs.3.gn.weight'].shape == (256,) and sd['gfl_cls.weight'].shape == (26, 256, 3, 3)
    assert sd['gfl_reg.weight'].shape == (68, 256, 3, 3) and sd['scales.4.scale'].shape == ()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():  # O(1) weights so the comparison is not dominated by the 0.01-std init
        for n, p in head.named_parameters():
            if n.endswith('conv.weight') or n.endswith('gfl_cls.weight') or n.endswith('gfl_reg.weight'):
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (9 * 256)) ** 0.5)
            elif n.endswith('gn.weight'):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif n.endswith('bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif n.endswith('scale'):
                p.copy_(torch.rand((), generator=g) + 0.5)
    ref_p = {k: v.detach().double().requires_grad_(True) for k, v in head.state_dict().items()}
    head = head.cuda()
    feats = [torch.randn(2, 256, s, s, generator=g) for s in (16, 8, 4, 2, 1)]  # 5 levels of a 128^2 image
    fg = [f.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
    cls, reg = head(fg)
    fr = [f.double().requires_grad_(True) for f in feats]
    rc, rr = GO.forward(fr, ref_p)
    for a, b in zip(cls + reg, rc + rr):
        assert tuple(a.shape) == tuple(b.shape)
        assert rel_err(a, b) < 1e-4, rel_err(a, b)
    R = [torch.randn(t.shape, generator=g) for t in rc + rr]
    sum((a * r.cuda()).sum() for a, r in zip(cls + reg, R)).backward()
    sum((a * r.double()).sum() for a, r in zip(rc + rr, R)).backward()
    got = dict(head.state_dict(keep_vars=True))
    worst = (0.0, None)
    for k, v in ref_p.items():
        gp = dict(head.named_parameters())[k].grad
        if k.endswith('conv.weight') or k in ('gfl_cls.weight', 'gfl_reg.weight'):
            gp = gp.permute(0, 3, 1, 2)  # kernel layout (Cout,3,3,Cin) -> reference (Cout,Cin,3,3)
        e = rel_err(gp, v.grad)
        if e > worst[0]:
            worst = (e, k)
    assert worst[0] < 1e-3, worst
    for a, b in zip(fg, fr):
        assert rel_err(a.grad, b.grad) < 1e-3
    assert len(got) == len(ref_p)


# ------------------------------------------------------------------------------------------- loss-side kernels (round 6)
def _pyramid_anchors(sizes, strides, scale=8.0):
    out = []
    for (h, w), s in zip(sizes, strides):
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
        cx, cy = xs.reshape(-1).float() * s, ys.reshape(-1).float() * s
        half = scale * s / 2.0
        out.append(torch.stack((cx - half, cy - half, cx + half, cy + half), 1))
    return out


def _random_gts(k, extent, g, lo=16.0, hi=160.0):
    wh = torch.rand(k, 2, generator=g) * (hi - lo) + lo
    c = torch.rand(k, 2, generator=g) * (extent - hi) + hi / 2
    return torch.cat((c - wh / 2, c + wh / 2), 1)


@pytest.mark.parametrize('seed,k,masked', [(0, 8, False), (1, 1, False), (2, 23, True), (3, 5, True), (4, 0, False)])
def test_atss_kernel_equals_the_masked_torch_form(seed, k, masked):
    """sm3_atss_assign (+ decode) against `gfl_losses.atss_assign` -- the restatement tests/test_gfl_loss_cpu.py pins on the
    oracle's indexing form -- on the SAR pyramid of a 512^2 image (5 levels, 5 456 anchors), with anchors outside the padded
    image masked: gt_inds integer-equal (random boxes: no exact distance ties; those have their own test below)."""
    from sm3det_amd import gfl_losses as GL
    from sm3det_amd.assign import ATSSAssigner
    g = torch.Generator().manual_seed(seed)
    strides = [8, 16, 32, 64, 128]
    sizes = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
    lv = _pyramid_anchors(sizes, strides)
    num = [a.shape[0] for a in lv]
    anchors = torch.cat(lv).cuda()
    gts = _random_gts(k, 512.0, g).cuda()
    labels = torch.randint(0, 26, (k,), generator=g).cuda()
    valid = None
    if masked:  # positions beyond a 400 x 448 padded image
        valid = torch.cat([((torch.arange(h)[:, None] * s < 400) & (torch.arange(w)[None, :] * s < 448)).reshape(-1)
                           for (h, w), s in zip(sizes, strides)]).cuda()
    want_inds, _, want_lab = GL.atss_assign(anchors, num, gts, labels, topk=9, valid=valid)
    res = ATSSAssigner(topk=9).assign(anchors, num, gts, gt_labels=labels, valid=valid)
    assert torch.equal(res.gt_inds, want_inds), int((res.gt_inds != want_inds).sum())
    if k:
        assert torch.equal(res.labels, want_lab)
        pos = want_inds > 0
        assert int(pos.sum()) > 0
        ov = GL.bbox_overlaps(anchors, gts)
        assert torch.equal(res.max_overlaps[pos], ov[pos, want_inds[pos] - 1])  # the winning IoU, bit for bit


def test_atss_kernel_distance_ties_go_to_the_lower_anchor_index():
    """a gt centred exactly between grid positions has equidistant anchors at the top-k cut; the kernel's rule (documented
    in csrc/gfl.hip: lower index first) against a stable-sort evaluation of the same algorithm"""
    import numpy as np
    from sm3det_amd import gfl_losses as GL
    from sm3det_amd.assign import ATSSAssigner
    strides, sizes = [8, 16], [(16, 16), (8, 8)]
    lv = _pyramid_anchors(sizes, strides)
    num = [a.shape[0] for a in lv]
    anchors = torch.cat(lv)
    gts = torch.tensor([[28.0, 28.0, 100.0, 100.0], [4.0, 36.0, 60.0, 92.0]])  # centres (64, 64) and (32, 64): on / between nodes
    topk = 9
    a, gt = anchors.numpy().astype(np.float32), gts.numpy().astype(np.float32)
    ov = GL.bbox_overlaps(anchors, gts).numpy()
    cx, cy = (a[:, 0] + a[:, 2]) / 2, (a[:, 1] + a[:, 3]) / 2
    best = np.full(a.shape[0], -1.0, np.float32)
    want = np.zeros(a.shape[0], np.int64)
    for gi in range(gt.shape[0]):
        gx, gy = (gt[gi, 0] + gt[gi, 2]) / 2, (gt[gi, 1] + gt[gi, 3]) / 2
        d = np.sqrt((cx - gx) ** 2 + (cy - gy) ** 2).astype(np.float32)
        cand, s = [], 0
        for n in num:
            cand += list(s + np.argsort(d[s:s + n], kind='stable')[:topk])
            s += n
        c = np.array(cand)
        thr = np.float32(ov[c, gi].astype(np.float64).mean()) + np.float32(ov[c, gi].astype(np.float64).std(ddof=1))
        for i in c:
            side = min(cx[i] - gt[gi, 0], cy[i] - gt[gi, 1], gt[gi, 2] - cx[i], gt[gi, 3] - cy[i])
            if ov[i, gi] >= thr and side > 0.01 and ov[i, gi] > best[i]:
                best[i], want[i] = ov[i, gi], gi + 1
    res = ATSSAssigner(topk=topk).assign(anchors.cuda(), num, gts.cuda())
    assert np.array_equal(res.gt_inds.cpu().numpy(), want), (np.nonzero(res.gt_inds.cpu().numpy() != want), want.sum())
    assert want.sum() > 0


@pytest.mark.parametrize('seed,ks,masked', [(0, (8, 8), False), (1, (3, 0), True), (2, (12, 5), True)])
def test_gfl_loss_kernels_equal_the_masked_torch_form_values_and_gradients(seed, ks, masked):
    """GFLHead.loss (one ATSS launch per image + sm3_gfl_loss_fwd / _bwd) against GFLHead.loss_torch (round 5's plain-PyTorch
    evaluation of the same formulas, itself pinned on oracle/gfl_oracle.py by tests/test_gfl_loss_cpu.py): every per-level
    loss value <= 1e-5 relative, gradients w.r.t. every head output <= 1e-4 (max-norm relative)."""
    from sm3det_amd.gfl_head import GFLHead
    torch.manual_seed(seed)
    head = GFLHead(num_classes=26, in_channels=256, stacked_convs=4, feat_channels=256,
                   anchor_generator=dict(type='AnchorGenerator', ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                                         strides=[8, 16, 32, 64, 128]),
                   loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=1.0),
                   loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25), reg_max=16,
                   loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
                   train_cfg=dict(assigner=dict(type='ATSSAssigner', topk=9), allowed_border=-1, pos_weight=-1)).cuda()
    g = torch.Generator().manual_seed(100 + seed)
    B, sizes = len(ks), [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
    mk = lambda c: [(torch.randn(B, c, h, w, generator=g) * 1.5).cuda().requires_grad_(True) for h, w in sizes]  # noqa: E731
    cls_a, box_a = mk(26), mk(68)
    cls_b = [t.detach().clone().requires_grad_(True) for t in cls_a]
    box_b = [t.detach().clone().requires_grad_(True) for t in box_a]
    gts = [_random_gts(k, 512.0, g).cuda() for k in ks]
    labels = [torch.randint(0, 26, (k,), generator=g).cuda() for k in ks]
    metas = [dict(pad_shape=((400, 448, 3) if masked else (512, 512, 3)), img_shape=(512, 512, 3)) for _ in ks]
    got = head.loss(cls_a, box_a, gts, labels, metas)
    want = head.loss_torch(cls_b, box_b, gts, labels, metas)
    wsum = torch.randn(3, 5, generator=g).cuda()
    tot_g = sum((torch.stack(got[k]) * wsum[i]).sum() for i, k in enumerate(('loss_cls', 'loss_bbox', 'loss_dfl')))
    tot_w = sum((torch.stack(want[k]) * wsum[i]).sum() for i, k in enumerate(('loss_cls', 'loss_bbox', 'loss_dfl')))
    for k in ('loss_cls', 'loss_bbox', 'loss_dfl'):
        a, b = torch.stack(got[k]).double(), torch.stack(want[k]).double()
        assert float((a - b).abs().max() / b.abs().max().clamp(min=1e-12)) < 1e-5, (k, a, b)
    tot_g.backward()
    tot_w.backward()
    for x, y in zip(cls_a + box_a, cls_b + box_b):
        assert rel_err(x.grad, y.grad) < 1e-4, rel_err(x.grad, y.grad)


# ------------------------------------------------------------------------------------------- pyramid levels on parallel streams
def _same_levels(ref, got):
    """outputs and input gradients bit for bit; the shared-weight gradients are sums over the levels, which autograd forms in
    the order the per-level gradients ARRIVE (stream-dependent): equal to fp32 rounding of a 5-term sum"""
    for kind in (0, 1):
        for i, (a, b) in enumerate(zip(ref[kind], got[kind])):
            assert torch.equal(a, b), (kind, i, float((a - b).abs().max()))
    for i, (a, b) in enumerate(zip(ref[2], got[2])):
        assert float((a - b).abs().max()) <= 4e-7 * float(a.abs().max()), (i, float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize('mode', [1, 2])
def test_levels_on_parallel_streams_give_the_serial_loops_bits(mode, monkeypatch):
    """sm3det_amd/level_streams.py (opt-in): the five levels of the shared-weight heads on their own streams -- outputs,
    input gradients must be the serial loop's bit for bit and the (level-accumulated) parameter gradients equal to summation
    order, eagerly and replayed from a hipGraph (fork / join captured as graph edges)."""
    from sm3det_amd import level_streams
    from sm3det_amd.rpn_head import OrientedRPNHead
    torch.manual_seed(3)
    head = OrientedRPNHead(in_channels=128, feat_channels=128).cuda()
    g = torch.Generator().manual_seed(4)
    feats = [torch.randn(2, 128, s, s, generator=g).cuda().contiguous(memory_format=torch.channels_last) for s in (32, 16, 8, 4, 2)]
    R = None

    def run():
        nonlocal R
        for p in head.parameters():
            p.grad = None
        fs = [f.detach().clone().requires_grad_(True) for f in feats]
        cls, reg = head(fs)
        if R is None:
            R = [torch.randn(t.shape, generator=g).cuda() for t in cls + reg]
        sum((a * r).sum() for a, r in zip(cls + reg, R)).backward()
        return ([t.detach().clone() for t in cls + reg], [f.grad.clone() for f in fs],
                [p.grad.clone() for p in head.parameters()])

    monkeypatch.setattr(level_streams, 'MODE', 0)
    monkeypatch.setattr(level_streams, 'ENABLED', False)
    ref = run()
    monkeypatch.setattr(level_streams, 'MODE', mode)
    monkeypatch.setattr(level_streams, 'ENABLED', True)
    got = run()
    torch.cuda.synchronize()
    _same_levels(ref, got)
    # under capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        cap = run()
    gr.replay()
    torch.cuda.synchronize()
    _same_levels(ref, cap)
