"""ctypes signatures + thin tensor-level callers for the backbone (MoE ConvNeXt) entry points of
libsm3det_hip.so (declared in include/sm3det_hip.h)."""
import ctypes

import torch

P, I, F, S = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

NT, NN, TN = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_SCALE_RES, EPI_GELU_BWD, EPI_BIAS_RELU = 0, 1, 2, 3, 4, 5


class GemmDesc(ctypes.Structure):
    """mirror of `sm3_gemm_desc` (include/sm3det_hip.h)"""
    _fields_ = [
        ('mode', ctypes.c_int32), ('epilogue', ctypes.c_int32),
        ('A', P), ('B', P), ('C', P),
        ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32),
        ('lda', ctypes.c_int32), ('ldb', ctypes.c_int32), ('ldc', ctypes.c_int32),
        ('group_offsets', P), ('num_groups', ctypes.c_int32), ('splits', ctypes.c_int32),
        ('stride_b', ctypes.c_int64), ('stride_bias', ctypes.c_int64),
        ('bias', P), ('aux_in', P), ('aux_out', P), ('gamma', P), ('rowscale', P),
        ('rows_per_scale', ctypes.c_int32), ('ld_aux', ctypes.c_int32),
        ('colsum_out', P), ('counters', P), ('tuning', ctypes.c_int32), ('compute', ctypes.c_int32),
        ('io', ctypes.c_int32),
    ]


def signatures():
    D = ctypes.POINTER(GemmDesc)
    LL = ctypes.c_long
    return {
        'sm3_gemm_f32_counter_slots': (I, []),
        'sm3_gemm_f32_workspace_bytes': (S, [D]),
        'sm3_gemm_f32': (I, [D, P, S, P]),
        'sm3_split_planes_f32': (I, [P, I, I, LL, P, LL, LL, I, P]),
        'sm3_colsum_f32': (I, [P, I, I, I, P, I, P, P]),
        'sm3_stem_patchify': (I, [P, P, I, I, I, P]),
        'sm3_layernorm_fwd': (I, [P, P, P, F, P, P, P, LL, I, I, I, I, P]),
        'sm3_cast_f32_f16': (I, [P, P, LL, P]),
        'sm3_row_reduce_workspace_bytes': (S, [I]),
        'sm3_layernorm_bwd': (I, [P, P, P, P, P, P, P, LL, I, I, I, I, I, P, S, P]),
        'sm3_dwconv7_fwd': (I, [P, P, P, P, P, I, I, I, I, I, P]),
        'sm3_dwconv7_bwd_weight': (I, [P, P, P, P, I, I, I, I, P]),
        'sm3_dwconv7_bwd_weight_acc': (I, [P, P, P, P, I, I, I, I, P]),
        'sm3_scale_bwd_prep': (I, [P, P, P, P, I, P, P, LL, I, P, S, P]),
        'sm3_conv3x3_nhwc_workspace_bytes': (S, [I, I, I, I, I, I, I]),
        'sm3_conv3x3_nhwc_fwd': (I, [P, P, P, P, I, I, I, I, I, I, I, P, S, P]),
        'sm3_relu_bwd': (I, [P, P, P, LL, P]),
        'sm3_sigmoid_f32': (I, [P, P, LL, P]),
        'sm3_midpoint_offset_encode_le90': (I, [P, P, I, P, P, P, P]),
        'sm3_delta_xywha_decode_le90': (I, [P, P, I, P, P, F, F, I, I, I, I, P, P]),
        'sm3_delta_xywha_encode_le90': (I, [P, P, I, P, P, F, I, I, P, P]),
        'sm3_rpn_decode_le90': (I, [P, P, P, P, I, P, P, F, P, P, P, P]),
        'sm3_conv3x3_nhwc_bwd_input': (I, [P, P, P, I, I, I, I, I, I, P, S, P]),
        'sm3_conv3x3_nhwc_bwd_weight_workspace_bytes': (S, [I, I, I, I, I, I]),
        'sm3_conv3x3_nhwc_bwd_weight': (I, [P, P, P, I, I, I, I, I, I, P, S, P]),
        'sm3_groupnorm_blocks': (I, [LL, I]),
        'sm3_groupnorm_workspace_bytes': (S, [I, LL, I, I]),
        'sm3_groupnorm_fwd': (I, [P, P, P, F, I, P, P, I, LL, I, I, P, S, P]),
        'sm3_groupnorm_bwd': (I, [P, P, P, P, P, I, P, P, I, LL, I, I, P, S, P]),
        'sm3_upsample2x_add': (I, [P, P, P, I, I, I, I, P]),
        'sm3_sumpool2x_add': (I, [P, P, P, I, I, I, I, P]),
        'sm3_row_partial_blocks': (I, [LL, I]),
        'sm3_row_partials_reduce': (I, [P, I, I, P, P]),
        'sm3_row_partials_reduce_multi': (I, [P, P, P, P, I, P]),
        'sm3_moe_router_partial_rows': (I, [I]),
        'sm3_moe_router_fwd': (I, [P, I, I, P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P]),
        'sm3_moe_router_bwd': (I, [P, I, I, P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
        'sm3_moe_gate_prep_fwd': (I, [P, P, P, P, P, F, I, I, I, I, P, P, P, P, P]),
        'sm3_moe_gate_prep_bwd': (I, [P, P, P, P, I, P, P, F, I, I, I, P, P, P, P, P, P]),
        'sm3_moe_aux_loss_fwd': (I, [P, I, I, F, P, P, P]),
        'sm3_moe_aux_loss_bwd': (I, [P, P, I, F, P, P, P]),
        'sm3_moe_plan_workspace_bytes': (S, [I, I]),
        'sm3_moe_plan': (I, [P, I, I, I, I, P, P, P, P, S, P]),
        'sm3_moe_dispatch': (I, [P, P, P, LL, I, P]),
        'sm3_moe_combine_fwd': (I, [P, P, P, P, P, P, I, P, LL, I, I, P]),
        'sm3_moe_combine_bwd': (I, [P, P, P, P, P, P, I, P, P, P, LL, I, I, P, S, P]),
        'sm3_moe_gather_add': (I, [P, P, P, LL, I, I, I, P]),
        'sm3_optim_chunk_elems': (I, []),
        'sm3_adamw_multi': (I, [P, P, P, P, P, P, P, I, P, P, F, F, F, F, P, P, P, P, P, F, F, I, P]),
        'sm3_dla_lr': (I, [P, I, P, I, P, P, I, P, P, I, I, I, F, F, F, F, P, P]),
        'sm3_conv3x3_set_arith': (I, [I]),
        'sm3_deform_conv_fwd_fused_supported': (I, [I] * 6),
        'sm3_deform_conv_fwd_fused': (I, [P, P, P, P] + [I] * 13 + [P]),
        'sm3_deform_im2col': (I, [P, P, P] + [I] * 13 + [LL, P]),
        'sm3_deform_col2im': (I, [P, P, P] + [I] * 13 + [LL, P]),
        'sm3_deform_col2im_nhwc': (I, [P, P, P] + [I] * 13 + [LL, P]),
        'sm3_deform_bwd_input_fused': (I, [P, P, P, P, P] + [I] * 13 + [LL, P]),
        'sm3_deform_col2im_coord': (I, [P, P, P, P] + [I] * 13 + [LL, P]),
    }


# Per-launch profiling hook (bench.py): when PROFILE is a list, every C-ABI call is bracketed by HIP events recorded
# on the stream the kernels are launched on, and (name, algorithmic flops, algorithmic bytes, ev0, ev1) is appended.
PROFILE = None
PROFILE_SHAPES = False


class _Prof:
    def __init__(self, name, flops=0.0, nbytes=0.0):
        self.rec = PROFILE is not None
        if self.rec:
            self.item = [name, float(flops), float(nbytes), torch.cuda.Event(enable_timing=True),
                         torch.cuda.Event(enable_timing=True)]

    def __enter__(self):
        if self.rec:
            self.item[3].record()
        return self

    def __exit__(self, *exc):
        if self.rec:
            self.item[4].record()
            PROFILE.append(tuple(self.item))
        return False


def call(name, *args, flops=0.0, nbytes=0.0):
    """Invoke `sm3_<name>` with tensors converted to device pointers and the current stream appended."""
    from . import _lib
    L = _lib.lib()
    conv = [(_p(a) if (a is None or isinstance(a, torch.Tensor)) else a) for a in args]
    with _Prof(name, flops, nbytes):
        _lib.check(getattr(L, 'sm3_' + name)(*conv, _lib.stream_ptr()), name)


def _p(t):
    return None if t is None else t.data_ptr()


_COUNTERS = {}
COMPUTE = 0  # 0: fp32 operands; 1: fp16 operands / fp32 accumulation (set by sm3det_amd.amp.autocast)
# How a COMPUTE == 0 (fp32) GEMM is evaluated -- sm3_gemm_desc.compute of the launch:
#   2 'bf16x3' (default): every fp32 operand element split exactly into three bf16 pieces in the loader, six
#     v_mfma_f32_32x32x16_bf16 products (i + j <= 2) with fp32 accumulation: fp32-equivalent results (24 operand bits kept,
#     dropped terms <= 2^-25 |a||b|; error vs the fp64 product measured per shape in tests/test_gemm_gpu.py and
#     profiles/r05/gemm_b3_error.txt) at 6/16 of the matrix-pipe time of the native fp32 instruction;
#   0 'f32': v_mfma_f32_32x32x2_f32 (an exact fp32 FMA chain, 157.3 TF/s peak).  SM3_GEMM_ARITH=f32 selects it.
ARITH32 = {'bf16x3': 2, 'f32': 0}[__import__('os').environ.get('SM3_GEMM_ARITH', 'bf16x3')]


def gemm_arith():
    """name of the arithmetic the fp32 GEMMs run in ('bf16x3' | 'f32'), for reports"""
    return 'bf16x3' if ARITH32 == 2 else 'f32'

TUNING = int(__import__('os').environ.get('SM3_GEMM_TUNING', '0'))  # benchmarking override forwarded to sm3_gemm_desc.tuning (scripts/gemm_sweep2.py); 0 in production


_SPARE = {}
_SPARE_COUNT = 8


def gemm_counters(device):
    """The zeroed ticket-counter array of the in-kernel split-K fix-up for torch's CURRENT stream on `device` (GEMMs on
    one stream run in order and may share it; the weight-gradient side stream gets its own).  The kernels leave it
    zeroed, so it is filled exactly once, when it is created -- and it is created OUTSIDE any hipGraph capture: an
    allocation + zero-fill made while a stream is capturing would live in that graph's private pool and be zeroed only when
    the graph replays (a later eager GEMM or another capture on the same stream would then find stale tickets, no block
    would draw the last ticket and the split-K output would silently never be written).  A stream first seen during a
    capture therefore takes an array from a spare set that the first eager call on the device zeroed.  One array must not
    be used by two graphs that are replayed CONCURRENTLY (they would share tickets); graphs replayed in order may."""
    from . import _lib
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream(device).cuda_stream)
    t = _COUNTERS.get(key)
    if t is not None:
        return t
    capturing = torch.cuda.is_current_stream_capturing()
    spare = _SPARE.get(dev)
    if spare is None:
        if capturing:
            raise _lib.SM3Error('the first GEMM on this device is being issued inside a hipGraph capture: run one warm-up '
                                'step eagerly first (the split-K ticket counters must be allocated and zeroed outside the '
                                'capture)')
        slots = _lib.lib().sm3_gemm_f32_counter_slots()
        spare = _SPARE[dev] = list(torch.zeros(_SPARE_COUNT, slots, dtype=torch.int32, device=device).unbind(0))
    if capturing:
        if not spare:
            raise _lib.SM3Error(f'more than {_SPARE_COUNT} capture streams issued GEMMs on device {dev} without an eager '
                                'warm-up on them: no zeroed ticket-counter array is left')
        t = spare.pop()
    else:
        t = torch.zeros(_lib.lib().sm3_gemm_f32_counter_slots(), dtype=torch.int32, device=device)
    _COUNTERS[key] = t
    return t


IO_A16, IO_B16, IO_C16, IO_X16 = 1, 2, 4, 8  # sm3_gemm_desc.io: tensors stored as fp16 (the AMP data path)
IO_APL, IO_BPL = 16, 32                      # ... operands stored as bf16x3 planes (fp32 path, NT / NN)


def planes(x, transpose=False, out=None, row_off=0, rows_total=None):
    """bf16x3 operand planes of the fp32 matrix x (R, K) -- or of x^T with transpose=True -- as a (3, K / 8, rows_total, 8)
    bfloat16 tensor (sm3_split_planes_f32); `out` / `row_off` place it inside a larger planes tensor (grouped operands)."""
    R, K = (x.shape[1], x.shape[0]) if transpose else (x.shape[0], x.shape[1])
    if out is None:
        out = torch.empty(3, K // 8, rows_total or R, 8, dtype=torch.bfloat16, device=x.device)
    call('split_planes_f32', x, x.shape[0], x.shape[1], x.stride(0), out, out.shape[2], row_off, int(transpose),
         nbytes=10.0 * x.numel())
    return out


def gemm(mode, A, B, C, M, N, K, *, epilogue=EPI_NONE, bias=None, aux_in=None, aux_out=None, gamma=None,
         rowscale=None, rows_per_scale=1, offsets=None, num_groups=1, splits=0, lda=None, ldb=None, ldc=None,
         ld_aux=None, colsum_out=None):
    """Enqueue one GEMM of the family on torch's current stream.  Tensors fp32 on the current device -- or, under
    COMPUTE == 1, fp16 for the operands / outputs the AMP data path stores as half: the `io` flags are derived from the
    tensors' dtypes, and a combination the library does not implement fails loudly.
    splits: 0 = let the library choose the split-K factor, 1 = none."""
    from . import _lib
    L = _lib.lib()
    d = GemmDesc()
    d.mode, d.epilogue = mode, epilogue
    d.A, d.B, d.C = _p(A), _p(B), _p(C)
    d.M, d.N, d.K = M, N, K
    if mode == NT:
        d.lda, d.ldb = lda or K, ldb or K
    elif mode == NN:
        d.lda, d.ldb = lda or K, ldb or N
    else:
        d.lda, d.ldb = lda or M, ldb or N
    d.ldc = ldc or N
    d.group_offsets = _p(offsets)
    d.num_groups = num_groups
    d.splits = splits
    if mode == NT:
        d.stride_b = N * (ldb or K)
    elif mode == NN:
        d.stride_b = K * (ldb or N)
    else:
        d.stride_b = 0
    d.stride_bias = N
    d.bias, d.aux_in, d.aux_out, d.gamma, d.rowscale = _p(bias), _p(aux_in), _p(aux_out), _p(gamma), _p(rowscale)
    d.rows_per_scale = rows_per_scale
    d.ld_aux = ld_aux or N
    d.colsum_out = _p(colsum_out)
    d.counters = _p(gemm_counters(C.device))
    d.tuning = TUNING
    d.compute = COMPUTE if COMPUTE else ARITH32
    h = torch.float16
    aux = aux_in if aux_in is not None else aux_out
    d.io = ((IO_A16 if A.dtype == h else 0) | (IO_B16 if B.dtype == h else 0) | (IO_C16 if C.dtype == h else 0) |
            (IO_X16 if (aux is not None and aux.dtype == h) else 0))
    pl = 0
    if A.dtype == torch.bfloat16 or B.dtype == torch.bfloat16:  # bf16x3 operand planes (3, K / 8, rows, 8): see `planes`
        if COMPUTE or d.compute != 2 or mode == TN:
            from ._lib import SM3Error
            raise SM3Error('bf16x3 operand planes are an operand form of the fp32 (bf16x3-arithmetic) NT / NN GEMMs only')
        if A.dtype == torch.bfloat16:
            pl |= IO_APL
            d.lda = A.shape[2]
        if B.dtype == torch.bfloat16:
            pl |= IO_BPL
            d.ldb = B.shape[2]
            d.stride_b = N  # rows between the groups' blocks
        d.io = pl
    if d.io and not COMPUTE and not pl:
        from ._lib import SM3Error
        raise SM3Error('fp16 tensors reached an fp32 GEMM (outside amp.autocast)')
    ws = None
    nbytes = L.sm3_gemm_f32_workspace_bytes(ctypes.byref(d))
    if nbytes:
        ws = _lib.workspace(nbytes, C.device)
    rows = K if mode == TN else M
    tag = ('gemm_f16_' if COMPUTE else 'gemm_f32_') + ('nt', 'nn', 'tn')[mode]  # ('f32' = fp32 tensors, either arithmetic)
    if PROFILE is not None and PROFILE_SHAPES:
        tag += f' {M}x{N}x{K} g{num_groups} e{epilogue} s{splits}'
    # algorithmic bytes: each operand read once, the output (and the epilogue's auxiliary tensor) written once
    ea, eb, ec = A.element_size(), B.element_size(), C.element_size()
    nb_alg = (ea * rows * (M if mode == TN else K) + eb * (rows if mode == TN else K) * N +
              ec * M * N * max(num_groups if mode == TN else 1, 1))
    if epilogue in (EPI_BIAS_GELU, EPI_BIAS_SCALE_RES, EPI_GELU_BWD):
        nb_alg += (aux.element_size() if aux is not None else 4.0) * M * N * (2 if epilogue == EPI_BIAS_SCALE_RES else 1)
    with _Prof(tag, 2.0 * rows * N * (M if mode == TN else K), nb_alg):
        _lib.check(L.sm3_gemm_f32(ctypes.byref(d), _p(ws), nbytes, _lib.stream_ptr()), 'gemm_f32')


def colsum(x, M, N, out, offsets=None, num_groups=1, ld=None):
    from . import _lib
    L = _lib.lib()
    with _Prof('colsum_f32', M * N, 4.0 * M * N):
        _lib.check(L.sm3_colsum_f32(_p(x), ld or N, M, N, _p(offsets), num_groups, _p(out), _lib.stream_ptr()),
                   'colsum_f32')


def row_ws(C, like):
    """scratch for the per-block column partials of the row kernels"""
    from . import _lib
    nb = _lib.lib().sm3_row_reduce_workspace_bytes(C)
    return _lib.workspace(nb, like.device), nb
