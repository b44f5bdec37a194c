"""debug aid: which part of the detector slice breaks hipGraph capture (each stage in its own process)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for st in (6, 7, 8, 9):
        r = subprocess.run([sys.executable, '-X', 'faulthandler', __file__, str(st)], capture_output=True, text=True)
        print('stage', st, 'rc', r.returncode, (r.stdout + r.stderr)[-300:].replace('\n', ' | '))
    sys.exit(0)
stage = int(sys.argv[1])
variant = stage
stage = min(stage, 6)
import torch, numpy as np
import bench
from tests import synth
from sm3det_amd.fpn import MultitaskFPN
from sm3det_amd.rpn_head import OrientedRPNHead, grid_anchors
from sm3det_amd.roi_head import RotatedShared2FCBBoxHead, RotatedSingleRoIExtractor
from sm3det_amd.assign import MaxIoUAssigner, RandomSampler
RES = int(os.environ.get('SLICE_RES', '512'))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bb = bench.build_model().cuda().train()
fpn = MultitaskFPN(in_channels=[96, 192, 384, 768], out_channels=256, extra_level=1, add_extra_convs='on_output', num_outs=5).cuda()
rpn = OrientedRPNHead(in_channels=256, feat_channels=256, version='le90', bbox_coder=dict(type='MidpointOffsetCoder', angle_range='le90', target_means=[0.0] * 6, target_stds=[1.0, 1.0, 1.0, 1.0, 0.5, 0.5])).cuda()
rpn.init_weights()
head = RotatedShared2FCBBoxHead(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=26, reg_class_agnostic=True).cuda()
ext = RotatedSingleRoIExtractor(dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True), 256, [4, 8, 16, 32])
asg = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, iou_calculator=dict(type='RBboxOverlaps2D'))
sampler = RandomSampler(num=512, pos_fraction=0.25, add_gt_as_proposals=False)
img = torch.randn(2, 3, RES, RES, device='cuda')
gts = [dev(synth.rotated_boxes(8, 40 + i, extent=float(RES))) for i in range(2)]
anchors = grid_anchors([(RES // s, RES // s) for s in (4, 8, 16, 32, 64)], [4, 8, 16, 32, 64], [8], [0.5, 1.0, 2.0], device='cuda')
cfg = dict(nms_pre=2000, max_per_img=2000, nms=dict(type='nms', iou_threshold=0.8), min_bbox_size=0)
bidx = torch.arange(2, device='cuda', dtype=torch.float32).view(2, 1, 1).expand(2, 512, 1)
def step():
    for m in (bb, fpn, rpn, head):
        for q in m.parameters(): q.grad = None
    feats, gl = bb(img, ['single'])
    pyr = fpn(feats)
    cls, reg = rpn(pyr)
    loss = gl + sum((c * c).mean() for c in cls) + sum((r * r).mean() for r in reg)
    if stage >= 2:
        with torch.no_grad():
            props, cnt = rpn.get_bboxes_fixed(cls, reg, (RES, RES, 3), cfg, mlvl_anchors=anchors)
            sel = []
            for i in range(2):
                p5 = props[i, :, :5].contiguous()
                if stage >= 3:
                    ar = asg.assign(p5, gts[i])
                    gi = ar.gt_inds
                else:
                    gi = torch.zeros(2000, dtype=torch.long, device='cuda')
                if stage >= 4:
                    idx = sampler.sample_fixed(gi)[0]
                else:
                    idx = torch.arange(512, device='cuda')
                sel.append(p5[idx])
            rois = torch.cat([bidx, torch.stack(sel)], -1).view(-1, 6)
        if stage >= 5:
            f = ext(pyr[:4], rois)
            loss = loss + (f * f).mean()
            if stage >= 6:
                a, b = head(f)
                loss = loss + (a * a).mean() + (b * b).mean()
    loss.backward()
    return loss
if variant in (7, 9):  # eager runs on the legacy default stream first (what bench.py's timeit does)
    for _ in range(2): step()
    torch.cuda.synchronize()
if variant in (8, 9):  # another captured graph that used the wgrad side stream is alive
    def bstep():
        for q in bb.parameters(): q.grad = None
        o, gl = bb(img, ['single'])
        (sum((t * t).mean() for t in o) + gl).backward()
    s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0): bstep()
    torch.cuda.current_stream().wait_stream(s0); torch.cuda.synchronize()
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0): bstep()
    g0.replay(); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    l = step()
g.replay(); torch.cuda.synchronize()
print('ok', float(l))
