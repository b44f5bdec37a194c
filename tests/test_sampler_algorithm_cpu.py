"""CPU tier: the ALGORITHM of the device-side RandomSampler (`sampler.hip`: count -> list the candidates whose key lies
under a threshold -> sort the short list by (key, index) -> slots; a short or overflowing list is re-listed with a doubled /
bisected threshold) restated in numpy, against the rule it must reproduce -- `RandomSampler.sample_fixed_host`
(sm3det_amd/assign.py: two stable sorts of the masked keys), which tests/test_oracle_heads_live.py compares with the
reference's own sampler class.  The kernels themselves are pinned on the same rule by tests/test_assign_gpu.py."""
import numpy as np
import pytest
import torch

CAP = 4096


def _threshold(m, count):
    if m <= 0:
        return np.float32(-1.0)
    want = np.float32(m) + np.float32(6.0) * np.sqrt(np.float32(m)) + np.float32(24.0)
    return np.float32(2.0) if want >= np.float32(count) else np.float32(want / np.float32(count))


def _select(keys, cand, m):
    """the m smallest keys among the candidate indices `cand`, in (key, index) order, the way the emit kernel finds them;
    also returns how many re-listing rounds the slow path took"""
    count = len(cand)
    tau, lo, hi = _threshold(m, count), np.float32(-1.0), np.float32(3.0)
    listed = cand[keys[cand] < tau]
    ok = m <= len(listed) <= CAP
    if not ok:
        if len(listed) < m:
            lo = tau
        else:
            hi = tau
    rounds = 0
    while not ok and rounds < 64:
        if hi > 2.5:
            tau = np.float32(2.0) if tau >= 0.5 else max(np.float32(tau * 2), np.float32(1e-6))
        else:
            tau = np.float32(0.5) * (max(lo, np.float32(0.0)) + hi)
        if rounds == 63:
            tau = np.float32(2.0) if hi > 2.5 else hi
        listed = cand[keys[cand] < tau]
        rounds += 1
        if len(listed) < m:
            lo = tau
        elif len(listed) > CAP:
            hi = tau
        else:
            ok = True
    listed = listed[:CAP]
    order = np.lexsort((listed, keys[listed]))  # by key, ties by index
    return listed[order][:m], rounds


def sample_kernel_model(gt_inds, keys, num, pos_fraction, neg_pos_ub):
    exp_pos = int(num * pos_fraction)
    pos, neg = np.flatnonzero(gt_inds > 0), np.flatnonzero(gt_inds == 0)
    m_pos = min(len(pos), exp_pos)
    m_neg = min(len(neg), num - m_pos)
    if neg_pos_ub >= 0:
        m_neg = min(m_neg, int(np.float32(neg_pos_ub) * np.float32(max(m_pos, 1))))
    sp, rp = _select(keys, pos, m_pos)
    sn, rn = _select(keys, neg, m_neg)
    return np.concatenate([sp, sn]), m_pos, m_neg, rp + rn


def _host(gt_inds, keys, num, frac, ub):
    from sm3det_amd.assign import RandomSampler
    s = RandomSampler(num=num, pos_fraction=frac, neg_pos_ub=ub, add_gt_as_proposals=False)
    idx, is_pos, valid, n_pos, n_neg = s.sample_fixed_host(torch.from_numpy(gt_inds), key=torch.from_numpy(keys))
    return idx.numpy()[valid.numpy()], int(n_pos), int(n_neg)


@pytest.mark.parametrize('n,n_pos,n_ign,num,frac,ub', [
    (261888, 150, 9000, 256, 0.5, -1), (2008, 700, 30, 512, 0.25, -1), (2008, 3, 0, 512, 0.25, -1), (300, 20, 100, 512, 0.25, -1),
    (5000, 64, 10, 256, 0.5, 3), (7, 0, 0, 256, 0.5, -1), (64, 64, 0, 256, 0.5, -1), (100000, 60000, 100, 2048, 0.5, -1)])
def test_threshold_list_selection_equals_the_two_sort_rule(n, n_pos, n_ign, num, frac, ub):
    rng = np.random.RandomState(n + n_pos)
    gt = np.zeros(n, np.int64)
    perm = rng.permutation(n)
    gt[perm[:n_pos]] = rng.randint(1, 9, n_pos)
    gt[perm[n_pos:n_pos + n_ign]] = -1
    for rep in range(3):
        keys = rng.rand(n).astype(np.float32)
        if rep == 2 and n > 16:
            keys[: n // 2] = keys[0]  # heavy ties
        got, mp, mn, rounds = sample_kernel_model(gt, keys, num, frac, ub)
        exp, ep, en = _host(gt, keys, num, frac, ub)
        assert (mp, mn) == (ep, en)
        assert np.array_equal(got, exp)
        if rep < 2:
            assert rounds == 0  # uniform keys: the first list holds the wanted count (margin 6 sigma)


def test_adversarial_keys_take_the_slow_path_and_stay_exact():
    n = 50000
    gt = np.zeros(n, np.int64)
    gt[:20000] = 1
    rng = np.random.RandomState(5)
    for keys in (rng.rand(n).astype(np.float32) * 0.5 + 0.5,        # none below 0.5: empty list, then overflow -> bisection
                 rng.rand(n).astype(np.float32) * np.float32(1e-4),  # all under the first threshold: overflow
                 np.full(n, 0.25, np.float32)):                       # one value everywhere: the bisection cannot separate
        got, mp, mn, rounds = sample_kernel_model(gt, keys, 256, 0.5, -1)
        assert rounds > 0 and (mp, mn) == (128, 128)
        if len(np.unique(keys)) > 1:
            exp, _, _ = _host(gt, keys, 256, 0.5, -1)
            assert np.array_equal(got, exp)
        else:  # (ties beyond the list capacity: still a valid sample -- right classes, no duplicates)
            assert (gt[got[:128]] > 0).all() and (gt[got[128:]] == 0).all() and len(set(got.tolist())) == 256


def test_first_list_falls_short_with_probability_below_1e_9():
    """the listing threshold's margin (the doc-comment's claim): the list length is Binomial(count, tau) <= Poisson(want)
    in the lower tail; P(length < m) < 1e-9 for every wanted count, and the list stays far below its capacity"""
    from scipy.stats import poisson
    for m in (1, 2, 8, 64, 128, 256, 512, 1024, 2048):
        want = m + 6.0 * np.sqrt(m) + 24.0
        assert poisson.cdf(m - 1, want) < 1e-9
        assert want + 8 * np.sqrt(want) < CAP
