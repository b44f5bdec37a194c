"""CPU tier: the ALGORITHM of `nms_sweep_lds_kernel` (sm3det_amd/csrc/ops_rotated.hip) restated step for step in numpy --
64-box blocks, the transposed diagonal tile, the parallel fixed point that resolves a block, survivors' rows applied one
step late by the helper waves while wave 0 adds the first off-diagonal word itself -- against the sequential greedy
sweep it replaces, on random suppression matrices from empty to dense and on adversarial chains.  It pins the reasoning
the kernel's comments give (what `remv[c]` holds when block c is decided; that an unchanged round of the fixed point is
the sequential answer; the bound of 64 rounds), independently of the GPU tests that pin the kernel itself."""
import numpy as np
import pytest


def greedy(M, n):
    """reference: box i survives unless an earlier survivor suppresses it; M[i, j] (j > i): i suppresses j"""
    removed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if not removed[i]:
            keep.append(i)
            removed |= M[i]
    return keep


def kernel_schedule(M, n):
    """the kernel's schedule: returns (keep list, max fixed-point rounds used)"""
    nblk = (n + 63) // 64
    pad = nblk * 64
    Mp = np.zeros((pad, pad), bool)
    Mp[:n, :n] = M
    remv = np.zeros((nblk + 4, 64), bool)          # removal vector, one 64-bit word per block
    kept_prev = np.zeros(64, bool)                  # s_kept of the previous step
    keep, max_rounds = [], 0
    for blk in range(nblk + 2):                     # whole triples of steps in the kernel: the extra ones are empty
        # helper waves: rows of the PREVIOUS block's survivors into remv[blk ..] (idempotent ORs)
        if blk >= 1 and blk - 1 < nblk:
            rows = Mp[(blk - 1) * 64:(blk) * 64]
            for c in range(blk, nblk):
                remv[c] |= rows[kept_prev][:, c * 64:(c + 1) * 64].any(0) if kept_prev.any() else False
        if blk >= nblk:
            kept_prev = np.zeros(64, bool)
            continue
        # wave 0: the block's boxes still alive, then the parallel fixed point over the transposed diagonal tile
        valid = np.arange(64) + blk * 64 < n
        S = valid & ~remv[blk]
        diag = Mp[blk * 64:(blk + 1) * 64, blk * 64:(blk + 1) * 64]
        diag_t = diag.T                              # diag_t[b, b'] : b' suppresses b (b' < b)
        kept = S.copy()
        rounds = 0
        for rounds in range(1, 67):
            nk = S & ~(diag_t & kept[None, :]).any(1)
            if (nk == kept).all():
                break
            kept = nk
        else:
            raise AssertionError('fixed point did not settle in 66 rounds')
        max_rounds = max(max_rounds, rounds)
        keep += [blk * 64 + int(b) for b in np.nonzero(kept)[0]]
        # wave 0 itself: first word right of the diagonal of every survivor -> the next block is ready without the helpers
        if blk + 1 < nblk and kept.any():
            remv[blk + 1] |= Mp[blk * 64:(blk + 1) * 64][kept][:, (blk + 1) * 64:(blk + 2) * 64].any(0)
        kept_prev = kept
    return keep, max_rounds


def _random_upper(n, density, rng):
    M = np.triu(rng.rand(n, n) < density, 1)
    return M


@pytest.mark.parametrize('n,density', [(1, 0.5), (64, 0.0), (64, 0.05), (65, 0.3), (200, 0.002), (500, 0.02), (700, 0.2),
                                       (1000, 0.9), (333, 0.5)])
def test_block_fixed_point_schedule_equals_sequential_greedy(n, density):
    rng = np.random.RandomState(n * 7 + int(density * 1000))
    for _ in range(3):
        M = _random_upper(n, density, rng)
        got, rounds = kernel_schedule(M, n)
        assert got == greedy(M, n)
        assert rounds <= 65


@pytest.mark.parametrize('n,width', [(64, 1), (300, 1), (256, 3), (200, 70)])
def test_longest_chains_need_as_many_rounds_as_the_chain_inside_a_block(n, width):
    """box i suppresses boxes i+1 .. i+width: the survivors are every (width+1)-th box and the dependency chain runs
    through every box of a block -- the fixed point needs about 64 / (width + 1) + 1 rounds there, never more than 65"""
    M = np.zeros((n, n), bool)
    for i in range(n):
        M[i, i + 1:i + 1 + width] = True
    got, rounds = kernel_schedule(M, n)
    assert got == greedy(M, n) == list(range(0, n, width + 1))
    assert rounds <= 65
    if width == 1 and n >= 64:
        assert rounds >= 32  # a 64-long alternating chain really is the slow case
