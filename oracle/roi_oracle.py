"""CPU oracle of the RoI path -- TEST INFRASTRUCTURE ONLY.

* ``extract``            -- RotatedSingleRoIExtractor.forward, /root/reference/mmrotate/models/roi_heads/roi_extractors/
  rotate_single_level_roi_extractor.py:103-140, with ``map_roi_levels`` (:66-84) in torch and the per-level
  RoIAlignRotated of the plain-C oracle (oracle/ops_oracle.c, itself bit-exact vs the compiled reference op);
* ``extract_backward``   -- its gradient w.r.t. every level (the reference gets it from autograd: index_put backward ->
  RoIAlignRotated backward per level);
* ``shared2fc_forward``  -- RotatedConvFCBBoxHead.forward for the Shared2FC form, convfc_rbbox_head.py:162-201.
Pinned: tests/test_oracle_heads_live.py runs the reference's own ``RotatedSingleRoIExtractor`` (over a stand-in for its
mmdet base class's constructor bookkeeping, with the reference's own ``RoIAlignRotated`` wrapper and the reference's CPU
C++ compiled by oracle/build_ref.py underneath) and ``RotatedShared2FCBBoxHead`` live: forward bit-identical, the
extractor's backward (through the reference's autograd Function) within 1e-5."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ops_oracle as OO


def map_roi_levels(rois, num_levels, finest_scale=56):
    scale = torch.sqrt(rois[:, 3] * rois[:, 4])
    target_lvls = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return target_lvls.clamp(min=0, max=num_levels - 1).long()


def extract(feats, rois, strides, out_size=7, sampling_ratio=2, aligned=True, clockwise=True, finest_scale=56):
    """feats: list of NCHW tensors; rois (k,6) -> (k, C, out, out), levels (k)"""
    lv = map_roi_levels(rois, len(feats), finest_scale)
    out = feats[0].new_zeros(rois.size(0), feats[0].size(1), out_size, out_size)
    for i, f in enumerate(feats):
        inds = (lv == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel() > 0:
            o = OO.roi_align_rotated_forward(f.numpy(), rois[inds].numpy(), out_size, out_size, 1.0 / strides[i],
                                             sampling_ratio, aligned, clockwise)
            out[inds] = torch.from_numpy(np.asarray(o))
    return out, lv


def extract_backward(gout, feats_shapes, rois, strides, out_size=7, sampling_ratio=2, aligned=True, clockwise=True,
                     finest_scale=56):
    lv = map_roi_levels(rois, len(feats_shapes), finest_scale)
    grads = []
    for i, shp in enumerate(feats_shapes):
        inds = (lv == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel() > 0:
            g = OO.roi_align_rotated_backward(gout[inds].contiguous().numpy(), rois[inds].numpy(), tuple(shp),
                                              out_size, out_size, 1.0 / strides[i], sampling_ratio, aligned,
                                              clockwise)
            grads.append(torch.from_numpy(np.asarray(g)))
        else:
            grads.append(torch.zeros(shp))
    return grads


def shared2fc_forward(x, p):
    x = x.flatten(1)
    for i in range(2):
        x = F.relu(F.linear(x, p[f'shared_fcs.{i}.weight'], p[f'shared_fcs.{i}.bias']))
    return F.linear(x, p['fc_cls.weight'], p['fc_cls.bias']), F.linear(x, p['fc_reg.weight'], p['fc_reg.bias'])
