"""Per-parameter gradient errors of one full-size case against the reference fixture under a chosen mode.

    python scripts/amp_grad_debug.py <case> <fp32|amp> <natural|forced> [top]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import torch  # noqa: E402

from tests import fullsize_common as FC  # noqa: E402
from tests.test_backbone_gpu import _ref_key_grads  # noqa: E402


def main():
    case, mode, routing = sys.argv[1:4]
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    fx = FC.load(case)
    cfg, seed = fx['cfg'], fx['seed']
    torch.manual_seed(0)
    net = ConvNeXt_moe_MultiInput(**cfg)
    net.load_state_dict(FC.seeded_state_dict(net.state_dict(), seed), strict=True)
    net = net.cuda().train()
    if mode == 'amp':
        from sm3det_amd import amp as _amp
        _amp.wrap_fp16_model(net)
    x, noise, drop = FC.make_inputs(case, noise_seed=fx['noise_seed'])
    fr = [r['topk'].to(torch.int32).cuda() for r in fx['routing']] if routing == 'forced' else None
    outs, gl = net(x.cuda(), ['single'], noise=[n.cuda() for n in noise], drop_scale=[d.cuda() for d in drop],
                   forced_routing=fr)
    scale = float(os.environ.get('SM3_DEBUG_LOSS_SCALE', '1'))
    L = FC.loss_of(outs, gl, seed) * scale
    L.backward()
    torch.cuda.synchronize()
    grads = _ref_key_grads(net)
    rows = []
    for key in fx['grads']['table']:
        g = grads[key] / scale
        e, l2 = FC.compare_grad(key, g, fx['grads'])
        mx, proj = FC.compare_grad_maxnorm(key, g, fx['grads'])
        rows.append((e, l2, key, bool(torch.isfinite(g).all()), mx, proj))
    rows.sort(reverse=True)
    print(f'== {case} {mode} {routing} storage={os.environ.get("SM3_AMP_STORAGE", "half")} loss_scale={scale}: '
          f'{sum(r[0] > 2e-2 for r in rows)} of {len(rows)} gradients above 2e-2')
    for e, l2, key, fin, mx, proj in rows[:top]:
        print(f'   {e:10.3e} l2 {l2:9.3e} maxnorm {mx:9.3e} proj {proj:9.3e} finite={fin} {key}')
    rows.sort(key=lambda r: -r[4])
    print('  by max-norm:')
    for e, l2, key, fin, mx, proj in rows[:top]:
        print(f'   {e:10.3e} l2 {l2:9.3e} maxnorm {mx:9.3e} proj {proj:9.3e} finite={fin} {key}')
    print(json.dumps(dict(gate_loss=float(gl), ref_gate_loss=fx['gate_loss'])))


if __name__ == '__main__':
    main()
