"""MI355X-native mirror of the reference backbone module
``mmrotate/models/backbones/convnext_moe.py`` (same class names, constructor kwargs, call signature, train/eval
semantics and ``state_dict`` key schema -- SURVEY.md 8(b).2), executing on hand-written gfx950 kernels.

Differences by design (documented in DESIGN.md):
  * activations stay token-major (T, C) = NHWC between blocks; the four outputs are returned as logically-NCHW
    tensors backed by channels_last memory (values identical to the reference's ``.contiguous()`` NCHW outputs);
    pass ``nchw_outputs=True`` to get NCHW-contiguous copies;
  * expert weights are stored fused, ``(E, 4C, C)`` etc., and translated to/from the reference's per-expert
    ``ffn.experts.{e}.pointwise_conv{1,2}.{weight,bias}`` keys in ``state_dict()`` / ``load_state_dict()``;
  * the dispatch never synchronises with the host (reference: ``.cpu()`` per MoE block, :259);
  * both gates of the reference (``gate='cosine'`` -- every SM3Det config -- and ``gate='linear'``) run on the
    kernels; ``linear_pw_conv=False`` and ``use_grn=True`` raise ``NotImplementedError`` at construction.
There is no CPU path: calling the module with CPU tensors raises.
"""
import math
from collections import OrderedDict
from itertools import chain
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import backbone_ops as ops
from .registry import ROTATED_BACKBONES


class LayerNorm2d(nn.LayerNorm):
    """Channel LayerNorm of a feature map (reference :30-47).  Parameter holder + token-level apply."""

    def __init__(self, num_channels, **kwargs):
        super().__init__(num_channels, **kwargs)
        self.num_channels = self.normalized_shape[0]

    def forward_tokens(self, x_tok, patch_major=False, H=0, W=0):
        return ops.layer_norm(x_tok, self.weight, self.bias, self.eps, patch_major, H, W)

    def forward(self, x, data_format='channel_first'):
        assert x.dim() == 4, f'LayerNorm2d only supports (N, C, H, W) inputs, got {tuple(x.shape)}'
        if data_format == 'channel_last':
            B, H, W, C = x.shape
            return self.forward_tokens(x.reshape(-1, C)).view(B, H, W, C)
        B, C, H, W = x.shape
        y = self.forward_tokens(x.permute(0, 2, 3, 1).reshape(-1, C))
        return y.view(B, H, W, C).permute(0, 3, 1, 2)


def build_LayerNorm2d_layer(cfg, num_features):
    """reference :48-67 (default eps 1e-5 unless the cfg says otherwise; configs pass 1e-6)."""
    if not isinstance(cfg, dict):
        raise TypeError('cfg must be a dict')
    if 'type' not in cfg:
        raise KeyError('the cfg dict must contain the key "type"')
    cfg_ = cfg.copy()
    cfg_.pop('type')
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    layer = LayerNorm2d(num_features, **cfg_)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return layer


class CosineTopKGate(nn.Module):
    """Parameters of the cosine router (reference :88-106): temperature, cosine_projector, sim_matrix."""

    def __init__(self, model_dim, num_global_experts, init_t=0.5):
        super().__init__()
        proj_dim = min(model_dim // 2, 256)
        self.temperature = nn.Parameter(torch.log(torch.full([1], 1.0 / init_t)), requires_grad=True)
        self.cosine_projector = nn.Linear(model_dim, proj_dim)
        self.sim_matrix = nn.Parameter(torch.randn(size=(proj_dim, num_global_experts)), requires_grad=True)
        self.clamp_max = math.log(1. / 0.01)
        nn.init.normal_(self.sim_matrix, 0, 0.01)
        self.proj_dim = proj_dim

    def normalized_sim_and_scale(self):
        snorm = F.normalize(self.sim_matrix, dim=0)
        scale = torch.clamp(self.temperature, max=self.clamp_max).exp()
        return snorm.contiguous(), scale


class FFN(nn.Module):
    """Linear(C,4C) -> GELU -> Linear(4C,C) parameter holder (reference :381-405)."""

    def __init__(self, in_channels, mid_channels, act_cfg=dict(type='GELU'), use_grn=False):
        super().__init__()
        if use_grn:
            raise NotImplementedError('use_grn=True is outside the SM3Det hot path')
        if act_cfg.get('type', 'GELU') != 'GELU':
            raise NotImplementedError('only GELU (erf) is implemented')
        self.pointwise_conv1 = nn.Linear(in_channels, mid_channels)
        self.act = nn.GELU()
        self.pointwise_conv2 = nn.Linear(mid_channels, in_channels)
        self.grn = None


class MoE_layer(nn.Module):
    """Sparse MoE FFN (reference :108-248): cosine top-k gate with noisy gating, E experts, aux load loss.

    Fused parameters: ``w1 (E,4C,C)``, ``b1 (E,4C)``, ``w2 (E,C,4C)``, ``b2 (E,C)``; exposed in ``state_dict`` under
    the reference keys ``experts.{e}.pointwise_conv{1,2}.{weight,bias}``."""

    def __init__(self, moe_cfg):
        super().__init__()
        self.noisy_gating = moe_cfg['noisy_gating']
        self.num_experts = moe_cfg['num_experts']
        self.input_size = moe_cfg['in_channels']
        self.k = moe_cfg['top_k']
        self.gating = moe_cfg['gating']
        if self.gating not in ('cosine', 'linear'):
            raise NotImplementedError(f"gate={self.gating!r}: the reference knows 'cosine' and 'linear' (:127-130)")
        if moe_cfg.get('use_grn', False):
            raise NotImplementedError('use_grn=True is outside the SM3Det hot path')
        assert self.k <= self.num_experts
        E, C, Hd = self.num_experts, self.input_size, moe_cfg['mid_channels']
        self.mid_channels = Hd
        w1, b1, w2, b2 = [], [], [], []
        for _ in range(E):  # same init distribution (and RNG consumption order) as E separate FFNs
            f = FFN(C, Hd)
            w1.append(f.pointwise_conv1.weight.data)
            b1.append(f.pointwise_conv1.bias.data)
            w2.append(f.pointwise_conv2.weight.data)
            b2.append(f.pointwise_conv2.bias.data)
        self.w1 = nn.Parameter(torch.stack(w1))
        self.b1 = nn.Parameter(torch.stack(b1))
        self.w2 = nn.Parameter(torch.stack(w2))
        self.b2 = nn.Parameter(torch.stack(b2))
        if self.gating == 'linear':  # reference :127-128: clean_logits = x @ w_gate
            self.w_gate = nn.Parameter(torch.zeros(C, E), requires_grad=True)
        else:
            self.w_gate = CosineTopKGate(C, E)
        self.w_noise = nn.Parameter(torch.zeros(C, E), requires_grad=True)
        self.register_buffer('mean', torch.tensor([0.0]))
        self.register_buffer('std', torch.tensor([1.0]))
        self.loss_coef = 1e-2  # reference :226 (forward(x, loss_coef=1e-2))
        self.last_expert_offsets = None  # device int32 (E+1): expert loads of the last forward (no sync)
        self.last_importance_load = None  # device fp32 (2E): [importance | load] of the last forward
        self.last_top_idx = None  # device int32 (T, min(k+1,E)): the routing of the last forward, best first

    # ---- reference key schema <-> fused storage ---------------------------------------------------------
    _FUSED = {'w1': 'pointwise_conv1.weight', 'b1': 'pointwise_conv1.bias',
              'w2': 'pointwise_conv2.weight', 'b2': 'pointwise_conv2.bias'}

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        fused = {ref: destination.pop(prefix + name) for name, ref in self._FUSED.items()}
        for e in range(self.num_experts):  # the reference's key order: expert-major, conv1 w/b then conv2 w/b
            for ref, t in fused.items():
                destination[f'{prefix}experts.{e}.{ref}'] = t[e]

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        for fused, ref in self._FUSED.items():
            keys = [f'{prefix}experts.{e}.{ref}' for e in range(self.num_experts)]
            if prefix + fused not in state_dict and all(k in state_dict for k in keys):
                state_dict[prefix + fused] = torch.stack([state_dict.pop(k) for k in keys])
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def cv_squared(self, x):
        """reference :140-147"""
        eps = 1e-10
        if x.shape[0] == 1:
            return x.new_zeros(())
        return x.float().var() / (x.float().mean() ** 2 + eps)


class DepthwiseConv7x7(nn.Module):
    """Parameter holder of ``nn.Conv2d(C, C, kernel_size=7, padding=3, groups=C)`` (reference :311-312).

    The weight lives TAP-MAJOR, ``(49, C)`` -- the layout the gfx950 kernels read (one 16-byte load = one tap of four
    channels) -- so no per-step transpose / flip launches are needed; ``state_dict`` shows and accepts it under the
    reference key and shape ``weight (C, 1, 7, 7)``."""

    def __init__(self, channels, kernel_size=7, padding=3):
        super().__init__()
        if kernel_size != 7 or padding != 3:
            raise NotImplementedError('only the 7x7 / padding 3 depthwise conv is implemented')
        ref = nn.Conv2d(channels, channels, kernel_size=7, padding=3, groups=channels)  # same init / RNG consumption
        self.in_channels = self.out_channels = self.groups = channels
        self.kernel_size, self.padding, self.stride = (7, 7), (3, 3), (1, 1)
        self.weight = nn.Parameter(ref.weight.data.view(channels, 49).t().contiguous())
        self.bias = nn.Parameter(ref.bias.data.clone())

    @property
    def weight_oihw(self):
        """the reference layout (C, 1, 7, 7) (a copy)"""
        return self.weight.t().reshape(self.in_channels, 1, 7, 7)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        w = destination[prefix + 'weight']
        destination[prefix + 'weight'] = w.t().reshape(self.in_channels, 1, 7, 7)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        k = prefix + 'weight'
        if k in state_dict and state_dict[k].dim() == 4:
            state_dict[k] = state_dict[k].reshape(self.in_channels, 49).t().contiguous()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def forward(self, x):
        raise RuntimeError('DepthwiseConv7x7 is executed inside ConvNeXtBlock.forward_tokens (fused dwconv + LN)')


class ConvNeXtBlock(nn.Module):
    """ConvNeXt block (reference :295-379): dw7x7 -> LN -> FFN | MoE -> layer scale -> drop path -> residual."""

    def __init__(self, in_channels, dw_conv_cfg=dict(kernel_size=7, padding=3), norm_cfg=dict(type='LN2d', eps=1e-6),
                 act_cfg=dict(type='GELU'), mlp_ratio=4., linear_pw_conv=True, MoE_cfg=None, drop_path_rate=0.,
                 layer_scale_init_value=1e-6, use_grn=False, with_cp=False):
        super().__init__()
        if not linear_pw_conv:
            raise NotImplementedError('linear_pw_conv=False is outside the SM3Det hot path')
        if dw_conv_cfg.get('kernel_size', 7) != 7 or dw_conv_cfg.get('padding', 3) != 3:
            raise NotImplementedError('only the 7x7 / padding 3 depthwise conv is implemented')
        if not layer_scale_init_value > 0:
            raise NotImplementedError('layer_scale_init_value must be > 0')
        self.with_cp = with_cp  # accepted for API parity; activations fit easily in 288 GB, never checkpointed
        self.in_channels = in_channels
        self.depthwise_conv = DepthwiseConv7x7(in_channels, dw_conv_cfg.get('kernel_size', 7),
                                               dw_conv_cfg.get('padding', 3))
        self.linear_pw_conv = linear_pw_conv
        self.norm = build_LayerNorm2d_layer(norm_cfg, in_channels)
        mid_channels = int(mlp_ratio * in_channels)
        # the reference constructs a dense FFN first and then overwrites it (:321 then :329/:331): keep the RNG
        # consumption identical so seeded inits line up
        self.ffn = FFN(in_channels, mid_channels, act_cfg, use_grn)
        self.MoE_cfg = MoE_cfg
        if MoE_cfg is not None:
            MoE_cfg = dict(MoE_cfg)
            MoE_cfg.update({'in_channels': in_channels, 'mid_channels': mid_channels, 'use_grn': use_grn,
                            'act_cfg': act_cfg})
            self.MoE_cfg = MoE_cfg
            self.ffn = MoE_layer(MoE_cfg)
        else:
            self.ffn = FFN(in_channels, mid_channels, act_cfg, use_grn)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((in_channels)), requires_grad=True)
        self.drop_path_rate = float(drop_path_rate)

    def _rowscale(self, B, device):
        """timm DropPath: per-sample Bernoulli(keep) / keep in train mode (device RNG), None otherwise."""
        if self.drop_path_rate == 0. or not self.training:
            return None
        keep = 1.0 - self.drop_path_rate
        rs = torch.empty(B, device=device).bernoulli_(keep)
        if keep > 0.0:
            rs.div_(keep)
        return rs

    def forward_tokens(self, x, B, H, W, noise=None, drop_scale=None, forced_topk=None):
        """x (B*H*W, C) -> (out tokens, gate loss or None).  noise / drop_scale: injected randomness (tests);
        forced_topk (T, k): teacher-forced routing (precision tests: see sm3_moe_router_fwd)."""
        C = self.in_channels
        w49 = self.depthwise_conv.weight  # tap-major (49, C)
        rs = self._rowscale(B, x.device) if drop_scale is None else drop_scale.to(x.device, torch.float32)
        if self.MoE_cfg is None:
            f = self.ffn
            out = ops.dense_block(x, w49, self.depthwise_conv.bias, self.norm.weight, self.norm.bias,
                                  f.pointwise_conv1.weight, f.pointwise_conv1.bias, f.pointwise_conv2.weight,
                                  f.pointwise_conv2.bias, self.gamma, rs, self.norm.eps, B, H, W)
            return out, None
        moe = self.ffn
        g = moe.w_gate
        train = bool(moe.training and moe.noisy_gating)
        if train and noise is None:
            noise = torch.randn(x.shape[0], moe.num_experts, device=x.device)  # torch.randn_like(clean) :203
        if moe.gating == 'linear':
            gate_args, clamp_max = (g, None, moe.w_noise, None, None), 0.0
        else:
            gate_args = (g.cosine_projector.weight, g.cosine_projector.bias, moe.w_noise, g.sim_matrix, g.temperature)
            clamp_max = g.clamp_max
        out, loss, tot, offsets, top_idx = ops.moe_block(
            x, w49, self.depthwise_conv.bias, self.norm.weight, self.norm.bias, *gate_args, moe.w1, moe.b1, moe.w2,
            moe.b2, self.gamma, rs, noise if train else None, self.norm.eps, B, H, W, moe.k, train, clamp_max,
            moe.loss_coef, forced_topk)  # aux loss (:234-238) comes out of the block: coef * (cv^2(importance) + cv^2(load))
        moe.last_expert_offsets = offsets
        moe.last_importance_load = tot
        moe.last_top_idx = top_idx
        return out, loss

    def forward(self, x):
        """API parity with the reference: NCHW in, (NCHW out, loss)."""
        B, C, H, W = x.shape
        tok = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
        out, loss = self.forward_tokens(tok, B, H, W)
        return out.view(B, H, W, C).permute(0, 3, 1, 2), loss


@ROTATED_BACKBONES.register_module()
class ConvNeXt_moe(nn.Module):
    """ConvNeXt with grid-level sparse MoE FFNs (reference :407-727)."""
    arch_settings = {
        'atto': {'depths': [2, 2, 6, 2], 'channels': [40, 80, 160, 320]},
        'femto': {'depths': [2, 2, 6, 2], 'channels': [48, 96, 192, 384]},
        'pico': {'depths': [2, 2, 6, 2], 'channels': [64, 128, 256, 512]},
        'nano': {'depths': [2, 2, 8, 2], 'channels': [80, 160, 320, 640]},
        'tiny': {'depths': [3, 3, 9, 3], 'channels': [96, 192, 384, 768]},
        'small': {'depths': [3, 3, 27, 3], 'channels': [96, 192, 384, 768]},
        'base': {'depths': [3, 3, 27, 3], 'channels': [128, 256, 512, 1024]},
        'swin_large': {'depths': [2, 2, 18, 2], 'channels': [192, 384, 768, 1536]},
        'large': {'depths': [3, 3, 27, 3], 'channels': [192, 384, 768, 1536]},
        'xlarge': {'depths': [3, 3, 27, 3], 'channels': [256, 512, 1024, 2048]},
        'huge': {'depths': [3, 3, 27, 3], 'channels': [352, 704, 1408, 2816]},
    }
    _multi_input = False

    def __init__(self, arch='tiny', in_channels=3, stem_patch_size=4, norm_cfg=dict(type='LN2d', eps=1e-6),
                 act_cfg=dict(type='GELU'), linear_pw_conv=True, use_grn=False, drop_path_rate=0.,
                 layer_scale_init_value=1e-6, out_indices=[0, 1, 2, 3], MoE_Block_inds=[[], [], [], []],
                 noisy_gating=True, num_experts=2, gate='cosine', top_k=2, frozen_stages=0,
                 gap_before_final_norm=False, with_cp=False, init_cfg=None, nchw_outputs=False):
        super().__init__()
        self.init_cfg = init_cfg
        if isinstance(arch, str):
            assert arch in self.arch_settings, f'Unavailable arch, please choose from ({set(self.arch_settings)})'
            arch = self.arch_settings[arch]
        elif isinstance(arch, dict):
            assert 'depths' in arch and 'channels' in arch
        self.depths = arch['depths']
        self.channels = arch['channels']
        assert isinstance(self.depths, Sequence) and isinstance(self.channels, Sequence) and \
            len(self.depths) == len(self.channels)
        if in_channels != 3 or stem_patch_size != 4:
            raise NotImplementedError('the stem kernel handles in_channels=3, stem_patch_size=4 (all SM3Det configs)')
        self.num_stages = len(self.depths)
        if isinstance(out_indices, int):
            out_indices = [out_indices]
        out_indices = list(out_indices)
        for i, index in enumerate(out_indices):
            if index < 0:
                out_indices[i] = 4 + index
                assert out_indices[i] >= 0, f'Invalid out_indices {index}'
        self.out_indices = out_indices
        self.MoE_Block_inds = MoE_Block_inds
        self.num_experts = num_experts
        self.frozen_stages = frozen_stages
        self.gap_before_final_norm = gap_before_final_norm
        self.nchw_outputs = nchw_outputs
        self.fp16_enabled = False  # set by amp.wrap_fp16_model (mmcv.runner.wrap_fp16_model): GEMMs with fp16 operands

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(self.depths))]
        block_idx = 0
        self.downsample_layers = nn.ModuleList()
        stem = nn.Sequential(nn.Conv2d(in_channels, self.channels[0], kernel_size=stem_patch_size,
                                       stride=stem_patch_size),
                             build_LayerNorm2d_layer(norm_cfg, self.channels[0]))
        self.downsample_layers.append(stem)
        self.stages = nn.ModuleList()
        for i in range(self.num_stages):
            depth, channels = self.depths[i], self.channels[i]
            if i >= 1:
                self.downsample_layers.append(nn.Sequential(
                    build_LayerNorm2d_layer(norm_cfg, self.channels[i - 1]),
                    nn.Conv2d(self.channels[i - 1], channels, kernel_size=2, stride=2)))
            moe_ind = [list(range(depth))[q] for q in self.MoE_Block_inds[i] if q < depth]
            stage = nn.Sequential(*[
                ConvNeXtBlock(in_channels=channels, drop_path_rate=dpr[block_idx + j], norm_cfg=norm_cfg,
                              act_cfg=act_cfg,
                              MoE_cfg={'noisy_gating': noisy_gating, 'num_experts': num_experts, 'top_k': top_k,
                                       'gating': gate} if j in moe_ind else None,
                              linear_pw_conv=linear_pw_conv, layer_scale_init_value=layer_scale_init_value,
                              use_grn=use_grn, with_cp=with_cp) for j in range(depth)])
            block_idx += depth
            self.stages.append(stage)
            if i in self.out_indices:
                self.add_module(f'norm{i}', build_LayerNorm2d_layer(norm_cfg, channels))
        self._freeze_stages()

    # ---------------------------------------------------------------------------------------------- forward
    def _stem_conv(self):
        return self.downsample_layers[0][0]

    def _stem_norm(self):
        return self.downsample_layers[0][1]

    def _forward_impl(self, x, noise=None, drop_scale=None, forced_routing=None):
        """x: (B,3,H,W) NCHW fp32 on the GPU.  noise: optional list of (T,E) tensors, one per MoE block (tests);
        forced_routing: optional list of (T,k) expert-index tensors, one per MoE block (teacher-forced routing, tests)."""
        if not x.is_cuda:
            raise RuntimeError('sm3det_amd backbone runs on the MI355X only (no CPU fallback)')
        B, _, Hi, Wi = x.shape
        conv = self._stem_conv()
        C0 = self.channels[0]
        a = ops.stem_patchify(x.float())
        w64 = F.pad(conv.weight.reshape(C0, 48), (0, 16))
        tok = ops.linear(a, w64, conv.bias)
        H, W = Hi // 4, Wi // 4
        outs, gate_losses = [], []
        # optional (data-parallel comm/compute overlap, see bench.py): remember the token tensor leaving stage
        # `stage_boundary` and the per-block gate losses with their stage, so a caller can run the backward in two
        # segments (late stages first, their gradient buckets all-reduce while the early stages' backward runs)
        boundary = getattr(self, 'stage_boundary', None)
        self._boundary_tokens, self._gate_loss_terms = None, []
        if self.training and (noise is None or drop_scale is None):
            gen_noise, gen_drop = self._step_randomness(B, H, W, x.device)
            noise = gen_noise if noise is None else noise
            drop_scale = gen_drop if drop_scale is None else drop_scale
        noise_iter = iter(noise) if noise is not None else None
        drop_iter = iter(drop_scale) if drop_scale is not None else None
        forced_iter = iter(forced_routing) if forced_routing is not None else None
        for i, stage in enumerate(self.stages):
            if i == 0:
                tok = self._stem_norm().forward_tokens(tok)
            else:
                ln, dconv = self.downsample_layers[i][0], self.downsample_layers[i][1]
                Cp, Cn = self.channels[i - 1], self.channels[i]
                pm = ln.forward_tokens(tok, patch_major=True, H=H, W=W)  # (T/4, 4*Cp), columns (kh,kw,c)
                wds = dconv.weight.permute(0, 2, 3, 1).reshape(Cn, 4 * Cp)
                tok = ops.linear(pm, wds, dconv.bias)
                H, W = H // 2, W // 2
            for blk in stage:
                nz = next(noise_iter) if (noise_iter is not None and blk.MoE_cfg is not None) else None
                ds = next(drop_iter) if drop_iter is not None else None
                fr = next(forced_iter) if (forced_iter is not None and blk.MoE_cfg is not None) else None
                tok, gl = blk.forward_tokens(tok, B, H, W, noise=nz, drop_scale=ds, forced_topk=fr)
                if gl is not None:
                    gate_losses.append(gl)
                    self._gate_loss_terms.append((i, gl))
            if boundary is not None and (i == boundary or (isinstance(boundary, (list, tuple)) and i in boundary)):
                if isinstance(boundary, (list, tuple)):  # several boundaries: {stage: tokens leaving it}
                    self._boundary_tokens = dict(self._boundary_tokens or {})
                    self._boundary_tokens[i] = tok
                else:
                    self._boundary_tokens = tok
            if i in self.out_indices:
                norm_layer = getattr(self, f'norm{i}')
                C = self.channels[i]
                if self.gap_before_final_norm:
                    gap = tok.view(B, H * W, C).mean(1)
                    outs.append(norm_layer.forward_tokens(gap.contiguous()))
                else:
                    o = norm_layer.forward_tokens(tok).view(B, H, W, C).permute(0, 3, 1, 2)
                    outs.append(o.contiguous() if self.nchw_outputs else o)
        if len(gate_losses) > 0:
            return tuple(outs), torch.stack(gate_losses).mean()  # == sum(losses) / len(losses), two launches
        return tuple(outs)

    def _step_randomness(self, B, H0, W0, device):
        """All of one training step's randomness in a handful of launches instead of three per block: the gating
        noise ``randn_like(clean_logits)`` (reference :203) of every noisy MoE block as views of ONE normal draw, and the
        per-sample DropPath scales (timm ``drop_path``: Bernoulli(keep)/keep) of every block from ONE uniform draw."""
        sizes, rates = [], []
        H, W = H0, W0
        for i, stage in enumerate(self.stages):
            if i > 0:
                H, W = H // 2, W // 2
            for blk in stage:
                rates.append(blk.drop_path_rate if blk.training else 0.0)
                if blk.MoE_cfg is not None:
                    moe = blk.ffn
                    sizes.append((B * H * W, moe.num_experts) if (moe.training and moe.noisy_gating) else None)
        noise = None
        if any(sz is not None for sz in sizes):
            flat = torch.randn(sum(t * e for t, e in filter(None, sizes)), device=device)
            noise, off = [], 0
            for sz in sizes:
                if sz is None:
                    noise.append(None)
                else:
                    noise.append(flat[off:off + sz[0] * sz[1]].view(sz))
                    off += sz[0] * sz[1]
        drop = None
        if any(r > 0 for r in rates):
            key = (str(device), tuple(rates))
            if getattr(self, '_keep_cache', (None, None))[0] != key:  # built once (outside any graph capture)
                keep = torch.tensor([1.0 - r for r in rates], dtype=torch.float32).to(device)
                self._keep_cache = (key, keep.view(-1, 1), keep.clamp_min(1e-30).reciprocal().view(-1, 1))
            _, keep, inv = self._keep_cache
            scales = (torch.rand(len(rates), B, device=device) < keep).float() * inv
            drop = [scales[j] if rates[j] > 0 else None for j in range(len(rates))]
        return noise, drop

    def forward(self, x):
        return self._run(x)

    def _run(self, x, **kw):
        """`@auto_fp16`-equivalent entry: fp16-operand GEMMs when wrap_fp16_model flagged the module (or the caller is
        already inside amp.autocast), fp32 otherwise."""
        from . import amp
        from . import _lib_backbone as LB
        with amp.autocast(bool(self.fp16_enabled) or LB.COMPUTE == 1):
            return self._forward_impl(x, **kw)

    # ---------------------------------------------------------------------------------------------- misc API
    def _freeze_stages(self):
        for i in range(self.frozen_stages):
            downsample_layer = self.downsample_layers[i]
            stage = self.stages[i]
            downsample_layer.eval()
            stage.eval()
            for param in chain(downsample_layer.parameters(), stage.parameters()):
                param.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self

    def get_layer_depth(self, param_name, prefix=''):
        """reference :616-658"""
        max_layer_id = 12 if self.depths[-2] > 9 else 6
        if not param_name.startswith(prefix):
            return max_layer_id + 1, max_layer_id + 2
        param_name = param_name[len(prefix):]
        if param_name.startswith('downsample_layers'):
            stage_id = int(param_name.split('.')[1])
            if stage_id == 0:
                layer_id = 0
            elif stage_id in (1, 2):
                layer_id = stage_id + 1
            else:
                layer_id = max_layer_id
        elif param_name.startswith('stages'):
            stage_id = int(param_name.split('.')[1])
            block_id = int(param_name.split('.')[2])
            if stage_id in (0, 1):
                layer_id = stage_id + 1
            elif stage_id == 2:
                layer_id = 3 + block_id // 3
            else:
                layer_id = max_layer_id
        else:
            layer_id = max_layer_id + 1
        return layer_id, max_layer_id + 2

    def remap_pretrained_state_dict(self, _state_dict):
        """ImageNet ConvNeXt (mmcls keys) -> this backbone: reference :851-891 (dense FFN weights are cloned into
        every expert of an MoE block; the stem is handled by the MultiInput subclass)."""
        state_dict = OrderedDict()
        for k, v in _state_dict.items():
            if not k.startswith('backbone.'):
                continue
            k = k[9:]
            if self._multi_input and 'downsample_layers.0.0' in k:
                state_dict[k.replace('downsample_layers.0.0', 'dataset_stems.single')] = v
            elif self._multi_input and 'downsample_layers.0.1' in k:
                state_dict[k.replace('downsample_layers.0.1', 'downsample_layers.0.0')] = v
            elif 'pointwise_conv' in k:
                parts = k.split('.')
                stage_ind, blocks_ind = int(parts[1]), int(parts[2])
                if blocks_ind in self.MoE_Block_inds[stage_ind]:
                    for e in range(self.num_experts):
                        state_dict[k.replace('pointwise_conv', f'ffn.experts.{e}.pointwise_conv')] = v
                else:
                    state_dict[k.replace('pointwise_conv', 'ffn.pointwise_conv')] = v
            else:
                state_dict[k] = v
        if state_dict and list(state_dict.keys())[0].startswith('module.'):
            state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
        return state_dict

    def init_weights(self):
        """reference :660-727 / :824-899: only the `Pretrained` init_cfg does anything useful."""
        if self.init_cfg is None:
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=.02)
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0.)
                elif isinstance(m, nn.LayerNorm):
                    nn.init.constant_(m.weight, 1.0)
                    nn.init.constant_(m.bias, 0.)
                elif isinstance(m, MoE_layer):
                    # the reference applies the same rule to every expert's pointwise_conv{1,2} Linear (they are
                    # nn.Linear modules there); here they are fused parameters
                    for e in range(m.num_experts):
                        nn.init.trunc_normal_(m.w1.data[e], std=.02)
                        nn.init.trunc_normal_(m.w2.data[e], std=.02)
                    nn.init.constant_(m.b1, 0.)
                    nn.init.constant_(m.b2, 0.)
            return
        cfg = self.init_cfg
        assert 'checkpoint' in cfg, f'Only support specify `Pretrained` in `init_cfg` in {self.__class__.__name__}'
        ckpt = torch.load(cfg['checkpoint'], map_location='cpu')
        if 'state_dict' in ckpt:
            _sd = ckpt['state_dict']
        elif 'model' in ckpt:
            _sd = ckpt['model']
        else:
            _sd = ckpt
        return self.load_state_dict(self.remap_pretrained_state_dict(_sd), strict=False)


@ROTATED_BACKBONES.register_module()
class ConvNeXt_moe_MultiInput(ConvNeXt_moe):
    """Tri-modality input variant (reference :730-899): inputs of all modalities are concatenated along the batch
    and share one stem (``dataset_stems['single']``)."""
    _multi_input = True

    def __init__(self, arch='tiny', in_channels=3, stem_patch_size=4, datasets=None, **kwargs):
        super().__init__(arch=arch, in_channels=in_channels, stem_patch_size=stem_patch_size, **kwargs)
        norm = self.downsample_layers[0][1]
        self.downsample_layers[0] = nn.Sequential(norm)
        self.datasets = ['single']
        self.dataset_stems = nn.ModuleDict()
        self.dataset_stems['single'] = nn.Conv2d(in_channels, self.channels[0], kernel_size=stem_patch_size,
                                                 stride=stem_patch_size)

    def _stem_conv(self):
        return self.dataset_stems['single']

    def _stem_norm(self):
        return self.downsample_layers[0][0]

    def forward(self, x, datasets=['single'], noise=None, drop_scale=None, forced_routing=None):
        if len(datasets) == 1:
            x = [x]
        x = torch.cat(list(x), dim=0)
        return self._run(x, noise=noise, drop_scale=drop_scale, forced_routing=forced_routing)
