"""Which Python lines of the detector slice still launch torch-native kernels?  One eager training step of the two-stage
(RGB) branch -- backbone + MultitaskFPN + OrientedRPNHead.forward_train + OrientedStandardRoIHead.forward_train + backward +
optimizer, the workload of bench.py's `full_slice_imgs_per_sec` -- under torch.profiler with stacks: aten op -> kernel
launches -> innermost repo frame, forward and backward separately (backward ops are attributed to the autograd node)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from sm3det_amd.config import build_detector_pieces  # noqa: E402
from sm3det_amd.optim import MultiTensorAdamW  # noqa: E402
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
params = [q for m in (bb, fpn, rpn, roi) for q in m.parameters() if q.requires_grad]
opt = MultiTensorAdamW([dict(params=[q]) for q in params], lr=1e-4, weight_decay=0.05, max_grad_norm=35.0)


def step():
    for q in params:
        q.grad = None
    feats, gl = bb(img, ['single'])
    pyr = fpn(feats)
    rl, props = rpn.forward_train(pyr, metas, gts, proposal_cfg=prop_cfg)
    ol = roi.forward_train(pyr, metas, props, gts, gls)
    loss = gl + sum(rl['loss_rpn_cls']) + sum(rl['loss_rpn_bbox']) + ol['loss_cls'] + ol['loss_bbox']
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
tim = collections.Counter()
total = 0
for ev in prof.events():
    if ev.device_type.name != 'CPU' or not ev.name.startswith('aten::'):
        continue
    ks = getattr(ev, 'kernels', None)
    if not ks:
        continue
    frame = next((s for s in ev.stack if '/sm3det_amd/' in s or 'slice_glue_audit' in s), ev.stack[0] if ev.stack else '')
    shapes = str([tuple(x) for x in (ev.input_shapes or []) if x])[:70]  # (stacks are not recorded on this build: shapes)
    key = (ev.name, (frame.strip()[-60:] + ' ' + shapes).strip())
    cnt[key] += len(ks)
    tim[key] += sum(k.duration for k in ks)
    total += len(ks)
print(f'torch-native kernel launches in one slice step: {total}, {sum(tim.values()) / 1e3:.2f} ms')
for key, n in cnt.most_common(70):
    print(f'{n:5d} {tim[key]:8.1f} us  {key[0]:32s} {key[1]}')
