"""oracle/gfl_oracle.py -- plain-torch restatement of the GFL head's conv towers (mmdet 2.x
``mmdet/models/dense_heads/gfl_head.py`` ``_init_layers`` / ``forward_single``: 4 x [Conv2d 3x3 no bias, GroupNorm(32),
ReLU] per tower, ``gfl_cls`` / ``gfl_reg`` 3x3 convs, per-level ``Scale``) over an mmdet-schema state_dict.

TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED**: mmdet is not vendored under /root/reference (requirements/runtime.txt:4
`mmdet>=2.25.1,<3.0.0`; call site: the type string at local_configs/main_SM3Det.py:30), so there is neither source nor a
reference test to check this restatement against; it follows the published mmdet 2.25 implementation from memory."""
import torch.nn.functional as F


def tower(x, p, prefix, n):
    for i in range(n):
        x = F.conv2d(x, p[f'{prefix}.{i}.conv.weight'], None, padding=1)
        x = F.relu(F.group_norm(x, 32, p[f'{prefix}.{i}.gn.weight'], p[f'{prefix}.{i}.gn.bias'], 1e-5))
    return x


def forward_single(x, p, level, stacked_convs=4):
    cls_feat = tower(x, p, 'cls_convs', stacked_convs)
    reg_feat = tower(x, p, 'reg_convs', stacked_convs)
    cls_score = F.conv2d(cls_feat, p['gfl_cls.weight'], p['gfl_cls.bias'], padding=1)
    bbox_pred = F.conv2d(reg_feat, p['gfl_reg.weight'], p['gfl_reg.bias'], padding=1) * p[f'scales.{level}.scale']
    return cls_score, bbox_pred.float()


def forward(feats, p, stacked_convs=4):
    outs = [forward_single(f, p, l, stacked_convs) for l, f in enumerate(feats)]
    return [o[0] for o in outs], [o[1] for o in outs]
