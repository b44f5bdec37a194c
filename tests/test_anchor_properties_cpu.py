"""CPU tier: the anchor grid, `valid_flags` and `anchor_inside_flags` restatements of mmdet's ``AnchorGenerator`` (absent from
/root/reference: restated from its published formula and labelled *parity unpinned* in DESIGN.md) narrowed by PROPERTY tests --
the invariants the published algorithm implies and that the reference's own consumers rely on
(mmrotate/models/dense_heads/rotated_rpn_head.py:117-143 `get_anchors` -> `anchor_inside_flags`, oriented_rpn_head.py:44-134;
local_configs/main_SM3Det.py:35-40,54-58: scales [8], ratios [0.5, 1, 2] for the RPN, octave_base_scale 8 x ratio 1 for GFL).
hypothesis drives the geometry; the structural properties are exact (integer / bit equality)."""
import math

import pytest
import torch
from hypothesis import given, settings, strategies as st

from sm3det_amd.rpn_head import grid_anchors

SIZES = st.lists(st.tuples(st.integers(1, 24), st.integers(1, 24)), min_size=1, max_size=5)
RATIOS = st.lists(st.sampled_from([0.25, 0.5, 1.0, 2.0, 4.0]), min_size=1, max_size=3, unique=True)
SCALES = st.lists(st.sampled_from([2.0, 4.0, 8.0, 16.0]), min_size=1, max_size=2, unique=True)


@settings(max_examples=60, deadline=None)
@given(SIZES, RATIOS, SCALES, st.sampled_from([4, 8, 16, 32, 64, 128]))
def test_grid_structure(sizes, ratios, scales, stride0):
    strides = [stride0 * 2 ** i for i in range(len(sizes))]
    lv = grid_anchors(sizes, strides, scales, ratios, device='cpu')
    nb = len(ratios) * len(scales)
    assert len(lv) == len(sizes)
    for (H, W), s, a in zip(sizes, strides, lv):
        assert a.shape == (H * W * nb, 4) and a.dtype == torch.float32
        g = a.view(H, W, nb, 4)
        # translation structure: anchor (y, x, b) = anchor (0, 0, b) + (x s, y s, x s, y s), exactly (integers times a power of 2)
        sx = torch.arange(W, dtype=torch.float32) * s
        sy = torch.arange(H, dtype=torch.float32) * s
        shift = torch.stack(torch.broadcast_tensors(sx[None, :], sy[:, None], sx[None, :], sy[:, None]), -1)
        assert torch.equal(g, g[0, 0][None, None] + shift[:, :, None, :])
        # centre offset 0: every base anchor is centred on its grid node
        ctr = (g[..., :2] + g[..., 2:]) / 2
        # (to fp32 rounding of node +- half-size: irrational half-sizes such as 8 s / sqrt(2) are not symmetric to the last bit)
        assert torch.allclose(ctr, shift[:, :, None, :2].expand_as(ctr), rtol=0, atol=1e-3)
        # base anchors: ratio-major, scale-minor; h / w = ratio, area = (scale * stride)^2
        base = g[0, 0]
        w, h = base[:, 2] - base[:, 0], base[:, 3] - base[:, 1]
        k = 0
        for r in ratios:
            for sc in scales:
                assert math.isclose(float(h[k] / w[k]), r, rel_tol=1e-6)
                assert math.isclose(float(w[k] * h[k]), (sc * s) ** 2, rel_tol=1e-5)
                k += 1
        assert bool((w > 0).all() and (h > 0).all())


@settings(max_examples=40, deadline=None)
@given(SIZES, st.sampled_from([4, 8, 16]))
def test_levels_are_nested_coarsenings(sizes, stride0):
    """a node of level l + 1 with even coordinates on level l's grid is the same image point: x_{l+1} s_{l+1} = (2 x_{l+1}) s_l"""
    strides = [stride0 * 2 ** i for i in range(len(sizes))]
    lv = grid_anchors(sizes, strides, [8.0], [1.0], device='cpu')
    for l in range(len(sizes) - 1):
        (H0, W0), (H1, W1) = sizes[l], sizes[l + 1]
        c0 = ((lv[l][:, :2] + lv[l][:, 2:]) / 2).view(H0, W0, 2)
        c1 = ((lv[l + 1][:, :2] + lv[l + 1][:, 2:]) / 2).view(H1, W1, 2)
        h, w = min(H1, (H0 + 1) // 2), min(W1, (W0 + 1) // 2)
        assert torch.allclose(c1[:h, :w], c0[::2, ::2][:h, :w], rtol=0, atol=1e-3)
        # and the anchors are twice as large
        s0 = lv[l][0, 2] - lv[l][0, 0]
        s1 = lv[l + 1][0, 2] - lv[l + 1][0, 0]
        assert float(s1) == 2 * float(s0)


@settings(max_examples=60, deadline=None)
@given(SIZES, st.integers(1, 400), st.integers(1, 400), st.sampled_from([8, 16]))
def test_valid_flags_are_the_positions_inside_the_padded_image(sizes, ph, pw, stride0):
    """AnchorGenerator.valid_flags (anchor_generator.py:397-450): position (y, x) of a level is valid iff y < min(ceil(h / s), H)
    and x < min(ceil(w / s), W) -- i.e. iff its node (x s, y s) lies inside the padded image, clipped to the map; with
    allowed_border = -1 (every SM3Det config) anchor_inside_flags IS valid_flags"""
    from sm3det_amd.gfl_head import GFLHead
    strides = [stride0 * 2 ** i for i in range(len(sizes))]
    head = GFLHead.__new__(GFLHead)
    head.strides = strides
    flags = GFLHead._valid_flags(head, sizes, (ph, pw, 3), 'cpu')
    lv = grid_anchors(sizes, strides, [8.0], [1.0], device='cpu')
    for (H, W), s, f, a in zip(sizes, strides, flags, lv):
        assert f.shape == (H * W,) and f.dtype == torch.bool
        ctr = (a[:, :2] + a[:, 2:]) / 2
        inside = (ctr[:, 0].round() < pw) & (ctr[:, 1].round() < ph)  # node coordinates are multiples of s: x s < w  <=>  x < ceil(w / s)
        assert torch.equal(f, inside)
        assert int(f.sum()) == min(math.ceil(ph / s), H) * min(math.ceil(pw / s), W)
        # monotone: a valid position has every position above / left of it valid
        g = f.view(H, W)
        assert torch.equal(g, g.cummin(0)[0].bool() & g.cummin(1)[0].bool() if g.numel() else g)


def test_rpn_anchor_count_of_the_baseline_geometry():
    """the 1024^2 RPN pyramid of local_configs/main_SM3Det.py:54-58: 3 base anchors on 256^2 .. 16^2 = 261 888 anchors; the GFL
    pyramid (strides 8..128, one base anchor): 21 824"""
    rpn = grid_anchors([(256, 256), (128, 128), (64, 64), (32, 32), (16, 16)], [4, 8, 16, 32, 64], [8], [0.5, 1.0, 2.0], 'cpu')
    assert sum(a.shape[0] for a in rpn) == 261888
    gfl = grid_anchors([(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], [8, 16, 32, 64, 128], [8], [1.0], 'cpu')
    assert sum(a.shape[0] for a in gfl) == 21824
    # ratio 0.5 is the WIDE anchor (h / w = 0.5): first base anchor of every position
    w, h = rpn[0][0, 2] - rpn[0][0, 0], rpn[0][0, 3] - rpn[0][0, 1]
    assert float(w) > float(h) and math.isclose(float(h / w), 0.5, rel_tol=1e-6)
