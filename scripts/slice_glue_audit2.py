"""Forward-side glue of the detector slice by source line: a TorchDispatchMode logs every aten op that launches work on
the GPU (any CUDA tensor among inputs / outputs) with the innermost sm3det_amd frame.  (The backward runs on autograd's own
thread, which the mode does not see: its glue is gradient accumulation `add_`, layout copies and fills.)"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
from torch.utils._pytree import tree_flatten  # noqa: E402

import bench  # noqa: E402
from sm3det_amd.config import build_detector_pieces  # noqa: E402
from tests import synth  # noqa: E402

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
torch.manual_seed(1)
pcs = build_detector_pieces(bench.load_config(bench.DEFAULT_CONFIG)['model'])
fpn, rpn, roi = pcs['neck'].cuda(), pcs['rgb_rpn_head'].cuda(), pcs['rgb_roi_head'].cuda()
rpn.init_weights()
roi.init_weights()
bb = bench.build_model().cuda().train()
B, RES = bench.BATCH, bench.RES
img = torch.randn(B, 3, RES, RES, device='cuda')
gts = [dev(synth.rotated_boxes(8, 40 + i)) for i in range(B)]
gls = [torch.randint(0, 26, (8,), generator=torch.Generator().manual_seed(50 + i)).cuda() for i in range(B)]
metas = [dict(img_shape=(RES, RES, 3), pad_shape=(RES, RES, 3)) for _ in range(B)]
prop_cfg = bench.load_config(bench.DEFAULT_CONFIG)['model']['rgb_train_cfg']['rpn_proposal']
VIEW = ('view', 'reshape', 'permute', 'transpose', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'as_strided', 'detach',
        'alias', 't.default', 'unbind', 'split', 'chunk', '_unsafe_view', 'empty', 'sym_', 'is_', 'size', 'stride', 'numel')


class Audit(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEW):
            return out
        flat, _ = tree_flatten((args, kwargs, out))
        if not any(isinstance(t, torch.Tensor) and t.is_cuda for t in flat):
            return out
        frame = '?'
        for fs in reversed(traceback.extract_stack()):
            if '/sm3det_amd/' in fs.filename:
                frame = f'{os.path.basename(fs.filename)}:{fs.lineno} {fs.line[:70] if fs.line else ""}'
                break
        self.cnt[(name.replace('aten.', ''), frame)] += 1
        return out


def fwd():
    feats, gl = bb(img, ['single'])
    pyr = fpn(feats)
    rl, props = rpn.forward_train(pyr, metas, gts, proposal_cfg=prop_cfg)
    ol = roi.forward_train(pyr, metas, props, gts, gls)
    return gl + sum(rl['loss_rpn_cls']) + sum(rl['loss_rpn_bbox']) + ol['loss_cls'] + ol['loss_bbox']


fwd().backward()
a = Audit()
with a:
    loss = fwd()
print('aten ops touching CUDA tensors in the forward (views excluded):', sum(a.cnt.values()))
byline = collections.Counter()
for (n, f), c in a.cnt.items():
    byline[f] += c
for f, c in byline.most_common(60):
    ops = ', '.join(f'{n}x{k}' for (n, ff), k in a.cnt.items() if ff == f)
    print(f'{c:4d}  {f:100s} {ops[:120]}')
