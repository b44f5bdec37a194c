"""Full-size (1024^2) value-parity fixtures of the backbone: shared by the generator
(tests/golden/make_golden_fullsize.py -- runs the REFERENCE module in the build container) and by the tests that
replay the same case on the MI355X (tests/test_fullsize_gpu.py) or on the CPU oracle (tests/test_oracle_fullsize.py).

Nothing large is stored.  Weights, input, gate noise, drop-path masks and the loss projections are regenerated from
seeds with torch's CPU generator (identical on the build container and the GPU box: same image, same torch); the
fixture holds only what the reference produced, compressed to statistics that still see every token:

* per output: 8192 sampled elements, the per-(image, channel) plane sums, and the channel sums of every 2x2 token
  block (any wrong token row moves one of those);
* the gate loss, and [importance | load] of every MoE block;
* the reference's routing (top-k expert sets per token, uint8) and its top-k margins, so that a token whose k-th and
  (k+1)-th logits are closer than fp32 noise can be recognised as a legitimate routing flip instead of a kernel bug;
* per parameter gradient (reference key schema): 512 sampled elements (everything if the tensor has <= 768), the sum
  and the L2 norm; all values packed into one flat tensor.
"""
import os
import zlib

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

TINY_E8 = dict(arch='tiny', MoE_Block_inds=[[], [0, 2], [0, 2, 4, 6, 8], [0, 2]], num_experts=8, top_k=2,
               drop_path_rate=0.1)   # local_configs/main_SM3Det.py:13-21 (BASELINE config #2)
TINY_E16 = dict(arch='tiny', MoE_Block_inds=[[], [], [0, 2, 4, 6, 8], [0, 2]], num_experts=16, top_k=2,
                drop_path_rate=0.1)  # ablation_moe_et_*e16t2_last2blocks.py (BASELINE config #4)

BASE_E8 = dict(arch='base', MoE_Block_inds=[[], [0, 2], [i * 2 for i in range(14)], [0, 2]], num_experts=8, top_k=2,
               drop_path_rate=0.1)   # local_configs/SM3Det_convnext_b.py:13-20 (BASELINE config #5: 18 MoE + 18 dense blocks)

CASES = {
    'full_e8t2_b1': dict(cfg=TINY_E8, batch=1, res=1024, seed=11),
    'full_e8t2_b2': dict(cfg=TINY_E8, batch=2, res=1024, seed=12),   # the headline configuration
    'full_e16t2_b1': dict(cfg=TINY_E16, batch=1, res=1024, seed=13),
    'full_e16t2_b2': dict(cfg=TINY_E16, batch=2, res=1024, seed=14),  # config #4 at its per-GPU batch
    'full_base_b1': dict(cfg=BASE_E8, batch=1, res=1024, seed=15),    # config #5 (ConvNeXt-B, C = 128..1024)
    'full_base_b2': dict(cfg=BASE_E8, batch=2, res=1024, seed=17),    # config #5 at the per-GPU batch BASELINE.json names
    # the gate-noise seed of the cases above is the best of 24 candidates by widest top-k margin (routing flips are rare by
    # construction); this one takes its FIRST candidate unselected, so near-ties occur at their natural rate
    'full_e8t2_b1_plainseed': dict(cfg=TINY_E8, batch=1, res=1024, seed=16, select_noise=False),
}
ARCHS = {'tiny': dict(depths=[3, 3, 9, 3], channels=[96, 192, 384, 768]),
         'base': dict(depths=[3, 3, 27, 3], channels=[128, 256, 512, 1024])}
ARCH_TINY = ARCHS['tiny']
N_OUT_SAMPLES, N_GRAD_SAMPLES, SMALL_TENSOR = 8192, 512, 768
FRAGILE_REL_GAP = 3e-2  # tokens whose top-k margin / logit scale is below this are listed in the fixture (fp32 replays may flip
# only margins < 2e-4, the fp16 data path of the AMP configs only margins < 2e-2 = its tolerance: tests/test_fullsize_gpu.py)


def _gen(tag, seed):
    return torch.Generator().manual_seed((zlib.crc32(tag.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def seeded_state_dict(template, seed):
    """One tensor per key of `template` (a reference-schema state_dict: only names and shapes are used), drawn from a
    generator seeded by (key, seed) -- independent of key order, so the reference module and the MI355X module get the
    same values from their own state_dict() templates.  Scales keep activations O(1) through 18 residual blocks."""
    out = {}
    for k, v in template.items():
        g = _gen(k, seed)
        shape = tuple(v.shape)
        if not v.is_floating_point() or k.endswith(('.mean', '.std')):
            out[k] = v.detach().clone()
        elif k.endswith('gamma'):
            out[k] = torch.empty(shape).uniform_(0.5, 1.5, generator=g)
        elif k.endswith('temperature'):
            out[k] = torch.tensor([1.2])
        elif k.endswith('sim_matrix'):
            out[k] = torch.randn(shape, generator=g)
        elif k.endswith('w_noise'):
            out[k] = torch.randn(shape, generator=g) * (0.5 / shape[0] ** 0.5)
        elif k.endswith('bias'):
            out[k] = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 1:  # LayerNorm weights
            out[k] = torch.empty(shape).uniform_(0.5, 1.5, generator=g)
        else:  # conv / linear weights: unit-gain fan-in scaling
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[k] = torch.randn(shape, generator=g) / fan_in ** 0.5
    return out


def moe_token_counts(cfg, batch, res):
    """tokens seen by each MoE block, in forward order"""
    out, H = [], res // 4
    for i, inds in enumerate(cfg['MoE_Block_inds']):
        if i > 0:
            H //= 2
        out += [batch * H * H] * len([q for q in inds if q < ARCHS[cfg['arch']]['depths'][i]])
    return out


def make_inputs(case, noise_seed=None):
    """x (B,3,R,R), noise [(T,E)] per MoE block, drop_scale [(B,)] per block -- all from seeds."""
    c = CASES[case]
    cfg, B, res, seed = c['cfg'], c['batch'], c['res'], c['seed']
    x = torch.randn(B, 3, res, res, generator=_gen('x', seed))
    ns = seed if noise_seed is None else noise_seed
    noise = [torch.randn(t, cfg['num_experts'], generator=_gen(f'noise{j}', ns))
             for j, t in enumerate(moe_token_counts(cfg, B, res))]
    nblocks = sum(ARCHS[cfg['arch']]['depths'])
    dpr = torch.linspace(0, cfg['drop_path_rate'], nblocks).tolist()
    drop = []
    for j, r in enumerate(dpr):
        keep = 1.0 - r
        m = torch.empty(B).bernoulli_(keep, generator=_gen(f'drop{j}', seed))
        if j % 6 == 5:  # make sure dropped samples occur at full size (rate 0.1 alone rarely drops any of B <= 2)
            m[-1] = 0.0
        drop.append(m / keep)
    return x, noise, drop


def loss_of(outs, gl, seed):
    """L = sum_i <out_i, R_i> / sqrt(numel_i) + 10 * gate_loss.  Returns (L, [R_i])."""
    L = 10.0 * gl
    for i, o in enumerate(outs):
        R = torch.randn(o.shape, generator=_gen(f'R{i}', seed)).to(o.device, o.dtype)
        L = L + (o * R).sum() / o.numel() ** 0.5
    return L


def sample_index(tag, numel, n, seed=0):
    return torch.randint(numel, (n,), generator=_gen('idx:' + tag, seed))


def summarise_output(i, o):
    """o: (B,C,H,W) on the CPU (any strides)."""
    o = o.detach().float().cpu()
    B, C, H, W = o.shape
    flat = o.contiguous().view(-1)
    idx = sample_index(f'out{i}', flat.numel(), N_OUT_SAMPLES)
    return dict(shape=(B, C, H, W), samples=flat[idx].clone(), plane_sum=o.double().sum((2, 3)).float(),
                block_sum=o.double().view(B, C, H // 2, 2, W // 2, 2).sum((1, 3, 5)).float(),
                max_abs=float(o.abs().max()), rms=float(o.double().pow(2).mean().sqrt()))


def summarise_grad(key, g):
    """-> (stats dict, values): values = the whole tensor if small, else N_GRAD_SAMPLES sampled elements"""
    g = g.detach().float().cpu().contiguous().view(-1)
    d = dict(numel=g.numel(), sum=float(g.double().sum()), l2=float(g.double().norm()), max_abs=float(g.abs().max()))
    vals = g.clone() if g.numel() <= SMALL_TENSOR else g[sample_index(key, g.numel(), N_GRAD_SAMPLES)].clone()
    return d, vals


def pack_grads(named_grads):
    """{key: grad} -> dict(table={key: stats + (off, n)}, values=flat tensor)"""
    table, vals, off = {}, [], 0
    for k, g in named_grads.items():
        d, v = summarise_grad(k, g)
        d['off'], d['n'] = off, v.numel()
        off += v.numel()
        table[k] = d
        vals.append(v)
    return dict(table=table, values=torch.cat(vals))


def load(case):
    return torch.load(os.path.join(GOLDEN, case + '.pt'), map_location='cpu', weights_only=False)


# ---------------------------------------------------------------------------------------------------------------
# comparison helpers (used identically by the GPU and the CPU-oracle tests)
def elementwise_err(a, ref, scale=None):
    """max over elements of |a - ref| / max(|ref|, 1e-2 * scale): the relative error of each element, with the
    denominator floored at 1 % of the tensor's largest magnitude (below that floor fp32 cancellation noise of a
    different summation order is not a relative quantity any more)."""
    a, ref = a.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    s = float(ref.abs().max()) if scale is None else scale
    return (a - ref).abs() / torch.clamp(ref.abs(), min=1e-2 * s + 1e-30)


def compare_output(i, o, ref):
    """returns dict(samples=elementwise rel errs (8192,), plane=..., block=...)"""
    got = summarise_output(i, o)
    sc = ref['max_abs']
    n_plane = ref['shape'][2] * ref['shape'][3]
    return dict(
        samples=elementwise_err(got['samples'], ref['samples'], sc),
        # sums of n elements of magnitude ~rms: scale the floor accordingly
        plane=elementwise_err(got['plane_sum'], ref['plane_sum'], ref['rms'] * n_plane ** 0.5 * 100),
        block=elementwise_err(got['block_sum'], ref['block_sum'], ref['rms'] * (4 * ref['shape'][1]) ** 0.5 * 100))


def compare_grad(key, g, packed):
    """-> (worst element-wise relative error over the stored elements, relative error of the L2 norm)"""
    ref = packed['table'][key]
    got, vals = summarise_grad(key, g)
    rv = packed['values'][ref['off']:ref['off'] + ref['n']]
    e = elementwise_err(vals, rv, ref['max_abs'])
    l2 = abs(got['l2'] - ref['l2']) / (ref['l2'] + 1e-30)
    return float(e.max()), l2


def compare_grad_maxnorm(key, g, packed):
    """-> (max |a - ref| over the stored elements / the tensor's largest |ref|, relative error of the projection of `a`
    on the reference direction over the stored elements): the max-norm form the AMP tests use (tests/test_amp_gpu.py);
    the projection catches a wrong overall scale, which random rounding noise does not produce"""
    ref = packed['table'][key]
    _, vals = summarise_grad(key, g)
    rv = packed['values'][ref['off']:ref['off'] + ref['n']].double()
    vals = vals.detach().double().cpu().reshape(-1)
    mx = float((vals - rv).abs().max() / (ref['max_abs'] + 1e-30))
    den = float((rv * rv).sum())  # 0 for the parameters of a block whose only sample was dropped (drop path)
    proj = float(abs((vals * rv).sum() / den - 1.0)) if den > 0 else 0.0
    return mx, proj
