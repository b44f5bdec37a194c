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


X_MEANS, X_STDS = (0., 0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2, 0.1)   # main_SM3Det.py RoI-head coder


def seeded_boxes(n=2048):
    """proposal boxes (x1y1x2y2), gt / roi oriented boxes (le90) and (n,5) deltas for the coder tests"""
    import math
    g = torch.Generator().manual_seed(77)
    props = torch.rand(n, 4, generator=g) * 500
    props[:, 2:] = props[:, :2] + torch.rand(n, 2, generator=g) * 200 + 1
    gt = torch.rand(n, 5, generator=g)
    gt[:, :2] *= 600
    gt[:, 2:4] = gt[:, 2:4] * 200 + 2
    gt[:, 4] = (gt[:, 4] - 0.5) * math.pi
    rois = gt.clone()
    rois[:, :2] += torch.randn(n, 2, generator=g) * 10
    rois[:, 2:4] *= 0.5 + torch.rand(n, 2, generator=g)
    rois[:, 4] = (torch.rand(n, generator=g) - 0.5) * math.pi
    deltas = torch.randn(n, 5, generator=g) * 0.5
    return props, gt, rois, deltas


def main():
    from oracle import ref_rpn
    T, C, X = ref_rpn.load()
    a, d = seeded_case()
    coder = C.MidpointOffsetCoder(target_means=MEANS, target_stds=STDS, angle_range='le90')
    obb = coder.decode(a, d)
    fx = dict(means=MEANS, stds=STDS, n=a.shape[0], proposals=obb, hboxes=T.obb2xyxy(obb, 'le90'))
    props, gt, rois, deltas = seeded_boxes()
    fx['midpoint_encode'] = C.bbox2delta(props, gt, MEANS, STDS, 'le90')
    fx['xywha'] = {}
    for es, pj in ((True, True), (False, False)):
        fx['xywha'][(es, pj)] = dict(
            encode=X.bbox2delta(rois, gt, X_MEANS, X_STDS, 'le90', None, es, pj),
            decode=X.delta2bbox(rois, deltas, X_MEANS, X_STDS, None, 16 / 1000, False, 32, 'le90', None, es, pj),
            decode_clamped=X.delta2bbox(rois, deltas, X_MEANS, X_STDS, (512, 640), 16 / 1000, False, 32, 'le90', None,
                                        es, pj))
    path = os.path.join(ROOT, 'tests', 'golden', 'rpn_decode.pt')
    torch.save(fx, path)
    print('rpn_decode', tuple(obb.shape), os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
