"""Generate tests/golden/rpn_decode.pt with the REFERENCE ``MidpointOffsetCoder.decode`` / ``obb2xyxy``
(/root/reference/mmrotate/core/bbox/..., imported through oracle/ref_rpn.py) on seeded anchors and deltas -- the
main_SM3Det.py coder settings (le90, stds (1,1,1,1,0.5,0.5)).  Includes saturating deltas (clamps), tiny and
degenerate anchors.  Run here; the .pt is committed.

    python tests/golden/make_golden_rpn.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
MEANS, STDS = (0., 0., 0., 0., 0., 0.), (1., 1., 1., 1., 0.5, 0.5)


def seeded_case(n=4096):
    g = torch.Generator().manual_seed(2024)
    a = torch.rand(n, 4, generator=g) * 900
    a[:, 2:] = a[:, :2] + torch.rand(n, 2, generator=g) * 300 + 0.5
    d = torch.randn(n, 6, generator=g) * 0.6
    d[: n // 16] *= 8.0          # saturate the dw/dh/da/db clamps
    a[n // 16: n // 8, 2:] = a[n // 16: n // 8, :2] + 1e-3   # tiny anchors
    return a, d


def main():
    from oracle import ref_rpn
    T, C = ref_rpn.load()
    a, d = seeded_case()
    coder = C.MidpointOffsetCoder(target_means=MEANS, target_stds=STDS, angle_range='le90')
    obb = coder.decode(a, d)
    fx = dict(means=MEANS, stds=STDS, n=a.shape[0], proposals=obb, hboxes=T.obb2xyxy(obb, 'le90'))
    path = os.path.join(ROOT, 'tests', 'golden', 'rpn_decode.pt')
    torch.save(fx, path)
    print('rpn_decode', tuple(obb.shape), os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
