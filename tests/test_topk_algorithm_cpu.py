"""CPU tier: the ALGORITHM of the merge-tree top-k (`topk_desc`, sm3det_amd/csrc/ops_rotated.hip) restated in numpy --
64-bit keys `(~orderable(score)) << 32 | index`, 4096-key chunks sorted with the direction bit of a bitonic sort's
k = 4096 stage (even chunks ascending, odd ones descending), then a tree of bitonic merges of two 2048-runs that keeps the
smaller half -- against the prefix of the stable descending sort it must equal."""
import numpy as np
import pytest

CHUNK, RUN = 4096, 2048


def _keys(scores):
    u = scores.astype(np.float32).view(np.uint32).astype(np.uint64)
    orderable = np.where(u & 0x80000000, ~u & 0xffffffff, u | 0x80000000)
    return ((~orderable & np.uint64(0xffffffff)) << np.uint64(32)) | np.arange(len(scores), dtype=np.uint64)


def _bitonic_merge_smaller_half(a_asc, b_desc):
    s = np.concatenate([a_asc, b_desc])  # ascending + descending = bitonic
    j = RUN
    while j > 0:
        t = np.arange(RUN)
        i = ((t & ~(j - 1)) << 1) | (t & (j - 1))
        q = i | j
        swap = s[i] > s[q]
        s[i[swap]], s[q[swap]] = s[q[swap]], s[i[swap]]
        j >>= 1
    return s[:RUN]


def topk_tree(scores, k):
    n = len(scores)
    npad = (n + CHUNK - 1) // CHUNK * CHUNK
    keys = np.full(npad, np.uint64(0xffffffffffffffff))
    keys[:n] = _keys(scores)
    chunks = keys.reshape(-1, CHUNK)
    for c in range(chunks.shape[0]):  # sort_lds_kernel, stages k = 2 .. 4096: direction = bit 12 of the element index
        chunks[c] = np.sort(chunks[c])[::-1] if c & 1 else np.sort(chunks[c])
    nruns = chunks.shape[0]
    if nruns == 1:
        cur = chunks[0][:RUN].copy()[None]
    else:
        cur, first = chunks, True
        while nruns > 1:
            out = []
            for p in range((nruns + 1) // 2):
                a = cur[2 * p][:RUN]
                if 2 * p + 1 < nruns:
                    b = cur[2 * p + 1][RUN:] if first else cur[2 * p + 1][::-1]  # odd chunk: its smaller half, descending
                else:
                    b = np.full(RUN, np.uint64(0xffffffffffffffff))
                out.append(_bitonic_merge_smaller_half(a.copy(), b.copy()))
            cur, nruns, first = np.stack(out), len(out), False
    return (cur[0][:min(k, n)] & np.uint64(0xffffffff)).astype(np.int64)


@pytest.mark.parametrize('n,k,ties', [(1, 1, True), (100, 17, True), (4096, 2000, True), (4097, 2048, False),
                                      (12295, 2000, True), (49152, 2000, False), (3 * 4096, 2048, True), (20000, 1, True)])
def test_merge_tree_topk_equals_prefix_of_stable_descending_sort(n, k, ties):
    rs = np.random.RandomState(n + k)
    v = rs.randint(0, 300, size=n).astype(np.float32) if ties else rs.permutation(n).astype(np.float32)
    v[::7] *= -1  # negative scores too: the orderable transform
    got = topk_tree(v, k)
    exp = np.argsort(-v.astype(np.float64), kind='stable')[:k]
    assert np.array_equal(got, exp)
