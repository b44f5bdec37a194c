"""Generate tests/golden/fpn_*.pt by running the REFERENCE MultitaskFPN (imported from /root/reference through
oracle/ref_fpn.py) on seeded inputs: outputs for the two call patterns of the detector (start_level 0 and 1,
trisource_H1stage_R2stage_detector.py:158-167) and the parameter / input gradients of a seeded scalar loss.
Run here (the reference does not exist on the GPU box); the .pt files are committed.

To keep the fixtures small the parameters are NOT stored: they are drawn per state_dict key from a generator seeded
with crc32(key) (``seeded_state_dict``), and gradients of tensors above 64 Ki elements are stored as every 16th
element of the flattened tensor (``sample``).

    python tests/golden/make_golden_fpn.py
"""
import os
import sys
import zlib

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = {
    # name: (ctor kwargs, input spatial size of level 0, batch, start levels to run)
    'fpn_main': (dict(in_channels=[32, 64, 128, 256], out_channels=128, extra_level=1, add_extra_convs='on_output',
                      num_outs=5), 32, 2, (0, 1)),
    'fpn_on_input': (dict(in_channels=[32, 64, 128, 256], out_channels=128, extra_level=1,
                          add_extra_convs='on_input', num_outs=5), 16, 1, (0, 1)),
    'fpn_maxpool': (dict(in_channels=[32, 64, 128, 256], out_channels=128, num_outs=5), 16, 2, (0, 1)),
}
SAMPLE_ABOVE, SAMPLE_STRIDE = 65536, 16


def seeded_state_dict(shapes):
    """{key: shape} -> {key: tensor}; O(0.05)-magnitude weights, O(0.1) biases, independent of iteration order."""
    out = {}
    for k, shape in shapes.items():
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
        out[k] = torch.randn(tuple(shape), generator=g) * (0.05 if len(shape) > 1 else 0.1)
    return out


def sample(t):
    return t.flatten()[::SAMPLE_STRIDE].clone() if t.numel() > SAMPLE_ABOVE else t.clone()


def seeded_inputs(kw, s0, B):
    g = torch.Generator().manual_seed(11)
    return [torch.randn(B, c, s0 >> i, s0 >> i, generator=g) for i, c in enumerate(kw['in_channels'])]


def seeded_proj(outs, sl):
    gp = torch.Generator().manual_seed(100 + sl)
    return [torch.randn(o.shape, generator=gp) for o in outs]


def main():
    from oracle import ref_fpn
    mod = ref_fpn.load_reference_module()
    for name, (kw, s0, B, levels) in CASES.items():
        net = mod.MultitaskFPN(**kw)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict(seeded_state_dict(shapes))
        xs = seeded_inputs(kw, s0, B)
        fx = dict(cfg=kw, s0=s0, batch=B, shapes=shapes, runs={})
        for sl in levels:
            for p in net.parameters():
                p.grad = None
            xin = [x.clone().requires_grad_(True) for x in xs]
            extra = kw.get('add_extra_convs', False)
            outs = net(xin, start_level=sl) if not extra else net(xin, start_level=sl, add_extra_convs=extra)
            sum((o * q).sum() for o, q in zip(outs, seeded_proj(outs, sl))).backward()
            fx['runs'][sl] = dict(outs=[o.detach().clone() for o in outs],
                                  grads={k: sample(v.grad) for k, v in net.named_parameters() if v.grad is not None},
                                  dinputs=[None if x.grad is None else x.grad.clone() for x in xin])
        path = os.path.join(ROOT, 'tests', 'golden', name + '.pt')
        torch.save(fx, path)
        print(name, 'outs', [tuple(o.shape) for o in fx['runs'][levels[0]]['outs']], os.path.getsize(path) // 1024,
              'KiB')


if __name__ == '__main__':
    main()
