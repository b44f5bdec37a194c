"""Seeded synthetic inputs shared by the CPU and GPU parity tests (SURVEY.md 8(d) shapes)."""
import numpy as np


def rotated_boxes(n, seed, extent=1024.0, wh=(8.0, 128.0), cluster=False):
    """(n,5) float32 (cx,cy,w,h,theta) with theta in U(-pi/2, pi/2) ('le90')."""
    rng = np.random.RandomState(seed)
    if cluster:  # dense overlaps: few centres, jittered
        k = max(1, n // 16)
        ctr = rng.uniform(0.1 * extent, 0.9 * extent, size=(k, 2))
        c = ctr[rng.randint(0, k, size=n)] + rng.normal(0, wh[1] * 0.15, size=(n, 2))
    else:
        c = rng.uniform(0, extent, size=(n, 2))
    w = rng.uniform(wh[0], wh[1], size=(n, 1))
    h = rng.uniform(wh[0], wh[1], size=(n, 1))
    t = rng.uniform(-np.pi / 2, np.pi / 2, size=(n, 1))
    return np.concatenate([c, w, h, t], 1).astype(np.float32)


def degenerate_rotated_pairs():
    """Pairs that hit the tie bands of the hull sort: identical, axis-aligned, shared edges, zero area."""
    b1 = np.array([
        [10, 10, 4, 6, 0.0], [10, 10, 4, 6, 0.3], [0, 0, 2, 2, 0.0], [0, 0, 2, 2, 0.0],
        [5, 5, 4, 4, np.pi / 4], [5, 5, 4, 4, 0.0], [1, 1, 1e-8, 1e-8, 0.1], [3, 3, 2, 6, np.pi / 2],
        [100, 100, 50, 20, 0.5], [100, 100, 50, 20, 0.5], [0, 0, 10, 10, 0.0], [7, 7, 3, 3, 1e-7],
    ], dtype=np.float32)
    b2 = np.array([
        [10, 10, 4, 6, 0.0], [10, 10, 4, 6, 0.3], [2, 0, 2, 2, 0.0], [1, 1, 2, 2, 0.0],
        [5, 5, 4, 4, 0.0], [5, 5, 4, 4, np.pi / 2], [1, 1, 3, 3, 0.2], [3, 3, 6, 2, 0.0],
        [100, 100, 50, 20, 0.5 + np.pi], [100, 100, 20, 50, 0.5 + np.pi / 2], [0, 0, 5, 5, 0.0],
        [7, 7, 3, 3, 0.0],
    ], dtype=np.float32)
    return b1, b2


def hboxes(n, seed, extent=1024.0, wh=(16.0, 128.0), cluster=False):
    rng = np.random.RandomState(seed)
    if cluster:
        k = max(1, n // 16)
        ctr = rng.uniform(0.1 * extent, 0.9 * extent, size=(k, 2))
        c = ctr[rng.randint(0, k, size=n)] + rng.normal(0, wh[1] * 0.2, size=(n, 2))
    else:
        c = rng.uniform(0, extent, size=(n, 2))
    w = rng.uniform(wh[0], wh[1], size=(n, 1))
    h = rng.uniform(wh[0], wh[1], size=(n, 1))
    return np.concatenate([c - np.c_[w, h] / 2, c + np.c_[w, h] / 2], 1).astype(np.float32)


def unique_scores(n, seed):
    """tie-free scores in (0,1)."""
    rng = np.random.RandomState(seed)
    s = rng.permutation(n).astype(np.float64) + rng.uniform(0.1, 0.9, size=n)
    return (s / (n + 1)).astype(np.float32)


def rois_for_level(n, seed, batch, extent, wh=(8.0, 256.0)):
    """(n,6) float32 [batch_idx, cx, cy, w, h, theta] in image coordinates."""
    b = rotated_boxes(n, seed, extent=extent, wh=wh)
    rng = np.random.RandomState(seed + 7)
    idx = rng.randint(0, batch, size=(n, 1)).astype(np.float32)
    return np.concatenate([idx, b], 1).astype(np.float32)
