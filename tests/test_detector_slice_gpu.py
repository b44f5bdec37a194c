"""GPU tier: the MI355X pieces chained the way TriSourceDetector chains them for the 2-stage (RGB) branch
(trisource_H1stage_R2stage_detector.py:142-170 extract_feat; oriented_standard_roi_head.py:60-95): backbone -> neck ->
Oriented-RPN tower -> proposals -> RoI extractor -> Shared2FC head, forward and backward, on NHWC memory end to end
(no layout copies between the pieces).  Checks composition, shapes, layouts and that every parameter of every piece
receives a finite gradient; numerical parity of each piece is covered by its own test file."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_backbone_neck_rpn_roi_chain_forward_backward():
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    from sm3det_amd.fpn import MultitaskFPN
    from sm3det_amd.roi_head import RotatedShared2FCBBoxHead, RotatedSingleRoIExtractor
    from sm3det_amd.rpn_head import OrientedRPNHead
    torch.manual_seed(0)
    B, RES = 2, 256
    backbone = ConvNeXt_moe_MultiInput(arch='tiny', MoE_Block_inds=[[], [0, 2], [0, 2, 4, 6, 8], [0, 2]],
                                       num_experts=8, top_k=2, drop_path_rate=0.1).cuda().train()
    neck = MultitaskFPN(in_channels=[96, 192, 384, 768], out_channels=256, extra_level=1,
                        add_extra_convs='on_output', num_outs=5).cuda()
    neck.init_weights()
    rpn = OrientedRPNHead(in_channels=256, feat_channels=256, version='le90',
                          bbox_coder=dict(type='MidpointOffsetCoder', angle_range='le90', target_means=[0.0] * 6,
                                          target_stds=[1.0, 1.0, 1.0, 1.0, 0.5, 0.5]),
                          test_cfg=dict(nms_pre=500, max_per_img=200, nms=dict(type='nms', iou_threshold=0.8),
                                        min_bbox_size=0)).cuda()
    rpn.init_weights()
    ext = RotatedSingleRoIExtractor(dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True), 256,
                                    [4, 8, 16, 32])
    head = RotatedShared2FCBBoxHead(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=26,
                                    reg_class_agnostic=True).cuda()
    head.init_weights()
    x = torch.randn(B, 3, RES, RES, device='cuda')
    feats, gate_loss = backbone(x, ['single'])
    assert all(f.is_contiguous(memory_format=torch.channels_last) for f in feats)
    pyr = neck(feats)  # start_level 0: the 2-stage branch
    assert [tuple(p.shape) for p in pyr] == [(B, 256, RES >> (2 + i), RES >> (2 + i)) for i in range(5)]
    assert all(p.permute(0, 2, 3, 1).is_contiguous() for p in pyr)  # NHWC memory handed on without a copy
    cls, reg = rpn(pyr)
    assert tuple(cls[0].shape) == (B, 3, RES // 4, RES // 4) and tuple(reg[0].shape) == (B, 18, RES // 4, RES // 4)
    props = rpn.get_bboxes(cls, reg)
    assert len(props) == B and all(p.shape[1] == 6 and 0 < p.shape[0] <= 200 for p in props)
    rois = torch.cat([torch.cat([torch.full((p.shape[0], 1), float(i), device='cuda'), p[:, :5]], 1)
                      for i, p in enumerate(props)])  # rbbox2roi: (batch_idx, cx, cy, w, h, a)
    roi_feats = ext(pyr[:4], rois)
    assert tuple(roi_feats.shape) == (rois.shape[0], 256, 7, 7)
    cls_score, bbox_pred = head(roi_feats)
    assert tuple(cls_score.shape) == (rois.shape[0], 27) and tuple(bbox_pred.shape) == (rois.shape[0], 5)
    loss = (cls_score.square().mean() + bbox_pred.square().mean() + gate_loss
            + sum(c.square().mean() for c in cls) + sum(r.square().mean() for r in reg))
    loss.backward()
    for name, mod in (('backbone', backbone), ('neck', neck), ('rpn', rpn), ('head', head)):
        for n, p in mod.named_parameters():
            if name == 'neck' and n.startswith('fpn_convs.5.'):
                assert p.grad is None  # the second extra level is only used by the start_level=1 (1-stage) call
                continue
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), f'{name}.{n}'
