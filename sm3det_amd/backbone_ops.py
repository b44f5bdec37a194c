"""Autograd bindings of the backbone kernels (host orchestration only -- every FLOP and byte of activation
traffic happens in libsm3det_hip.so through the C ABI).

Each ``torch.autograd.Function`` below is one unit of the reference's backbone
(mmrotate/models/backbones/convnext_moe.py) with a hand-written backward:

* ``linear``           nn.Linear / the stem and downsample convs as patch GEMMs (:533-558, :783-791)
* ``layer_norm``       LayerNorm2d / block norm (:30-47, :351), optionally writing patch-major rows
* ``dense_block``      ConvNeXtBlock._inner_forward with a dense FFN (:343-372, :397-405)
* ``moe_block``        ConvNeXtBlock._inner_forward with MoE_layer (:226-248): gate, noisy top-k, dispatch,
                       grouped expert GEMMs, combine -- returns (out, importance, load)

Activations are token-major (T, C) fp32.  No host synchronisation happens anywhere in here.
"""
import os

import torch
from torch.autograd import Function

from . import _lib_backbone as LB
from ._lib import SM3Error

call, gemm, colsum = LB.call, LB.gemm, LB.colsum


class _compute:
    """run the enclosed GEMM launches in the arithmetic the forward of this autograd node used (fp32, or fp16 operands
    under sm3det_amd.amp.autocast) -- backward may execute outside the autocast block"""

    def __init__(self, mode, pair=None):
        self.mode, self.pair = mode, pair  # pair: the PAIR_DGRAD level in force when the node's forward ran (see `pairing`)

    def __enter__(self):
        global PAIR_DGRAD
        self.old, LB.COMPUTE = LB.COMPUTE, self.mode
        self.old_pair = PAIR_DGRAD
        if self.pair is not None:
            PAIR_DGRAD = self.pair

    def __exit__(self, *exc):
        global PAIR_DGRAD
        LB.COMPUTE, PAIR_DGRAD = self.old, self.old_pair
        return False


def _e(*shape, like, dtype=None):
    return torch.empty(*shape, device=like.device, dtype=dtype or torch.float32)


def _chk(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise SM3Error(f'{name}: expected a float32 tensor on the GPU (got {t.dtype} on {t.device}); '
                       'there is no CPU/eager fallback')
    return t.contiguous()


# ---- weight-gradient side stream -------------------------------------------------------------------------------
# The weight / bias gradient kernels of a block (split-K TN GEMMs, column sums, gate-parameter backward, depthwise
# weight gradient) feed nothing in the dx chain.  They are enqueued on ONE side stream, forked after the tensors they
# read exist and joined at the end of the block's backward, so the chip runs them underneath the latency-bound small
# kernels of the dx chain (router, LayerNorm, combine, depthwise) and the tails of the dgrad GEMMs.  Works the same
# eagerly and under hipGraph capture (the fork/join become graph edges).
# OFF by default since the GEMM k-loop rework: with the GEMMs filling the chip, two of them sharing it lose more than the
# small kernels gain from the overlap (training step 20.2 ms without, 20.6 ms with the side stream; it was worth 1.6 ms
# when the GEMM family ran at 0.48 of peak).  SM3_WGRAD_STREAM=1 turns it back on.
OVERLAP_WGRAD = os.environ.get('SM3_WGRAD_STREAM', '0') == '1'
_SIDE = {}
# test aid (tests/test_fullsize_gpu.py): when a list, every MoE block's backward appends its per-workgroup partial sums of
# d(scale) -- the terms of the fully cancelling sum behind d(temperature) -- so the test can state the sum's conditioning
DEBUG_DSCALE = None


def _side_stream(device):
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)
    return s


def _on_side(device, fn):
    """run fn (kernel launches only) on the side stream, after everything enqueued so far on the current stream"""
    if not OVERLAP_WGRAD:
        return fn()
    main, side = torch.cuda.current_stream(device), _side_stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        return fn()


_PAIRED = {}  # device -> a paired launch is outstanding on the side stream


# SM3_DEFER_JOIN: the side stream is joined ONCE, at the end of the backward pass (the same end-of-pass callback that runs the
# batched parameter-gradient reductions), instead of at the end of every block's backward: what the side stream produces --
# weight gradients -- is read by nobody before the pass ends (autograd ADOPTS them as p.grad; a data-parallel reducer packs
# them only after flush_deferred_reductions(), which joins).  The side stream then runs up to a block behind the input-
# gradient chain instead of stalling it at 18 joins: 16.03 -> 15.80 ms per step.  Every tensor a side-stream kernel reads
# is marked with record_stream (the caching allocator must not hand its memory out again while that kernel may be running;
# under hipGraph capture such blocks simply stay reserved until the capture ends: measured 15.1 -> 15.5 GB reserved for
# the headline step).  Joined at once when a
# weight already has a gradient (autograd would ADD to it right after the block returns) or when no end-of-pass callback
# is armed (a backward function driven by hand).
DEFER_JOIN = os.environ.get('SM3_DEFER_JOIN', '1') == '1'


def _callback_armed(device):
    """an end-of-pass callback of THIS backward pass is registered (it joins the side stream)"""
    if not (DEFER_JOIN and BATCH_REDUCE):
        return False
    try:
        task = torch._C._current_graph_task_id()
    except Exception:  # noqa: BLE001
        return False
    return task is not None and task >= 0 and _PENDING_TASK.get(device.index) == task


def _join_side(device, force=False):
    if not force and not OVERLAP_WGRAD and _PAIRED.get(device) == 'deferrable' and _callback_armed(device):
        return  # joined by flush_deferred_reductions() at the end of the pass
    if OVERLAP_WGRAD or _PAIRED.pop(device, False):
        torch.cuda.current_stream(device).wait_stream(_side_stream(device))


# SM3_PAIR_DGRAD: concurrent partners for the backward launches.  The input-gradient GEMM of an FFN layer (dh = dy . W2 with
# GELU', dx = dh . W1) and that layer's weight gradient (dW2 = dy^T . act, dW1 = dh^T . xn) read the same operand and nothing
# of each other.  Run one after the other each pays its own ramp and tail: a launch ends with CUs holding one or two
# k-loops (a lone workgroup reaches ~48 % of the matrix pipe, profiles/r06/gemm_trace_eq_prio.txt), from stage 1 on the
# input gradients are launches of one or two workgroups per CU, and every split-K weight gradient is a single round of 768.
# Issued INPUT GRADIENT FIRST on this stream and the weight gradient right behind it on the side stream, the second
# launch's workgroups take the slots the first one leaves empty.  The order matters: the older form (SM3_WGRAD_STREAM=1,
# weight gradient first) lets the single-round launch occupy every slot and the partner queue behind it -- measured SLOWER
# than no overlap at all.  Levels (same box, ms per step of the headline workload, profiles/r06/bench_pair_dgrad.txt):
#   0 off 16.55 | 1 FC1 pair 16.58* | 2 + FC2 pair 15.83 | 3 + depthwise dgrad / wgrad 15.82 | 4 + gate dgrad / wgrad 15.72
#   (* measured on another box against 16.88 for level 0)
# Not kept (measured): two / three side streams that the pairs rotate over -- 15.69 -> 16.44 / 16.76 ms (more than two GEMMs
# sharing the chip lose more than the fill gains).
# Not kept (measured): the same idea in the FORWARD pass -- the two halves of the batch as FC1(a) | FC2(a) beside FC1(b) | FC2(b)
# for the stage-2 / 3 dense blocks, whose FC2 is a launch of one workgroup per CU -- 15.59 -> 15.75 ms: half-size launches and two
# more graph edges per block cost more than the fill gains.
PAIR_DGRAD = int(os.environ.get('SM3_PAIR_DGRAD', '4'))
PAIR_MAX_OUTPUTS = int(os.environ.get('SM3_PAIR_MAX_OUTPUTS', 1 << 30))  # (A/B aid: rows x C up to which pairs are formed)


class pairing:
    """``with pairing(level):`` the blocks whose FORWARD runs inside use this SM3_PAIR_DGRAD level in their backward.  The
    whole-detector step turns it off: replayed from one hipGraph of ~2 500 nodes the cross-stream edges cost more than the
    pairs gain (56.6 -> 58.0 ms; eager 57.4 -> 57.4), where the backbone step alone gains 0.8 of 16.6 ms."""

    def __init__(self, level):
        self.level = int(level)

    def __enter__(self):
        global PAIR_DGRAD
        self.old, PAIR_DGRAD = PAIR_DGRAD, self.level

    def __exit__(self, *exc):
        global PAIR_DGRAD
        PAIR_DGRAD = self.old
        return False


def _paired(device, main_fn, side_fn, outputs, level=1, reads=(), grads_of=()):
    """-> (main_fn(), side_fn()); both may only read what is already enqueued on the current stream.  `reads`: the tensors
    side_fn's kernels read; `grads_of`: the parameters whose gradients side_fn produces (see SM3_DEFER_JOIN)"""
    if OVERLAP_WGRAD or PAIR_DGRAD < level or outputs > PAIR_MAX_OUTPUTS:
        r_side = _on_side(device, side_fn)
        return main_fn(), r_side
    main, side = torch.cuda.current_stream(device), _side_stream(device)
    ready = torch.cuda.Event()
    ready.record(main)
    r_main = main_fn()
    side.wait_event(ready)
    with torch.cuda.stream(side):
        r_side = side_fn()
    for t in reads:
        if t is not None:
            t.record_stream(side)
    # a parameter that already holds a gradient gets this one ADDED as soon as the block returns: not deferrable
    if any(getattr(q, 'grad', None) is not None for q in grads_of if q is not None) or _PAIRED.get(device) == 'now':
        _PAIRED[device] = 'now'
    else:
        _PAIRED[device] = 'deferrable'
    return r_main, r_side


# The column reductions of the row kernels' per-workgroup partials produce PARAMETER gradients (d LayerNorm weight / bias,
# d layer scale, d bias): nothing in the backward chain reads them.  They are collected while the backward pass runs and
# reduced by ONE launch when it ends (autograd's end-of-pass callback), instead of one ~5 us launch each -- 36 per
# ConvNeXt-T training step, all at the launch floor.  SM3_BATCH_REDUCE=0 restores the per-call launches.
BATCH_REDUCE = os.environ.get('SM3_BATCH_REDUCE', '1') == '1'
_PENDING_REDUCE = {}  # device index -> [(workspace, rows, columns, out)]
_PENDING_TASK = {}    # device index -> id of the backward pass (graph task) whose end-of-pass callback is registered


def flush_deferred_reductions(device=None):
    """run the collected reductions now (autograd calls this at the end of a backward pass; call it yourself if the
    backward functions of this module are driven without autograd)"""
    import ctypes
    from . import _lib
    for dev in list(_PAIRED):  # the deferred join of the side stream (SM3_DEFER_JOIN)
        _join_side(dev, force=True)
    for di in ([device.index if hasattr(device, 'index') else device] if device is not None else list(_PENDING_REDUCE)):
        items = _PENDING_REDUCE.pop(di, [])
        if not items:
            continue
        # the row kernels that produced the partials may have run on other streams than the one current here (the engine
        # runs its final callbacks under the caller's streams BEFORE it joins them with the backward streams)
        with torch.cuda.device(di):
            here = torch.cuda.current_stream()
            for st in {it[5] for it in items if it[5] is not None and it[5] != here}:
                here.wait_stream(st)
        n = len(items)
        P = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in items])
        O = (ctypes.c_void_p * n)(*[it[3] for it in items])
        NB = (ctypes.c_int * n)(*[it[1] for it in items])
        NC = (ctypes.c_int * n)(*[it[2] for it in items])
        with torch.cuda.device(di), LB._Prof('row_partials_reduce'):
            _lib.check(_lib.lib().sm3_row_partials_reduce_multi(P, O, NB, NC, n, _lib.stream_ptr()), 'row_partials_reduce_multi')


def _deferred_reduce(ws, T, C, ncols, out, params=()):
    """column reduction of a row kernel's per-workgroup partials (parameter gradients): batched at the end of the backward
    pass, or (SM3_BATCH_REDUCE=0) one launch now, on the side stream when that is on.  `params`: the parameters `out` is
    the gradient of -- if one already HAS a gradient, autograd will add `out` to it as soon as this backward function
    returns, so the reduction cannot wait (gradient accumulation; never the case in the training step, which adopts
    gradients)."""
    from . import _lib
    nblk = _lib.lib().sm3_row_partial_blocks(T, C)
    if not BATCH_REDUCE or OVERLAP_WGRAD or any(getattr(q, 'grad', None) is not None for q in params):
        _on_side(out.device, lambda: call('row_partials_reduce', ws, nblk, ncols, out))
        return
    # One end-of-pass callback per BACKWARD PASS, not per non-empty list: the engine drops its final callbacks when a pass
    # raises (an OOM that is caught and retried, an error in a later node), which would leave the list non-empty with no
    # callback behind it -- every later pass would append without ever reducing.  The pass is identified by the engine's
    # graph-task id where torch exposes it; entries left over from a pass that never flushed are reduced first (their
    # workspaces are still alive, so this is merely late, never wrong), and the flush itself is idempotent.
    di = out.device.index
    lst = _PENDING_REDUCE.setdefault(di, [])
    try:
        task = torch._C._current_graph_task_id()
    except Exception:  # noqa: BLE001  (private API: fall back to "one callback per entry", still correct)
        task = None
    if task is None or task < 0 or _PENDING_TASK.get(di) != task:
        if lst and task is not None and task >= 0:
            flush_deferred_reductions(di)  # leftovers of a pass whose callback never ran
            lst = _PENDING_REDUCE.setdefault(di, [])
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda: flush_deferred_reductions(di))
        except RuntimeError:  # not inside a backward pass (a backward function called by hand): reduce now
            call('row_partials_reduce', ws, nblk, ncols, out)
            return
        _PENDING_TASK[di] = task
    # the workspace tensor and the output's STORAGE are kept alive until the flush -- the storage, not the tensor: another
    # reference to the gradient tensor itself would make autograd's AccumulateGrad clone it (before it is filled) instead of
    # adopting it as p.grad
    lst.append((ws, nblk, ncols, out.data_ptr(), out.untyped_storage(), torch.cuda.current_stream(out.device)))


# Zero-initialised accumulation targets of a backward pass (the depthwise weight / bias gradients, which their kernel builds
# with atomics) come from ONE arena per pass, filled by ONE launch: 18 zero-fill launches per ConvNeXt-T step -> 1.  The arena
# of a pass is sized by what the previous pass used (the first pass, or a pass that needs more, simply opens another chunk);
# a pass is identified by autograd's graph-task id, and slices stay alive as long as the gradients that view them.
_ZERO_ARENA = {}  # device index -> dict(task, buf, used, need)


def _zero_slab(n, like):
    di = like.device.index
    try:
        task = torch._C._current_graph_task_id()
    except Exception:  # noqa: BLE001
        task = -1
    st = _ZERO_ARENA.get(di)
    if task < 0:
        return torch.zeros(n, device=like.device, dtype=torch.float32)
    if st is None or st['task'] != task:
        need = max(n, st['total'] if st is not None else 0)
        st = _ZERO_ARENA[di] = dict(task=task, buf=torch.zeros(need, device=like.device, dtype=torch.float32), used=0,
                                    total=0)
    if st['used'] + n > st['buf'].numel():  # (first pass of a model, or a larger one: another chunk)
        st['buf'] = torch.zeros(max(n, st['buf'].numel()), device=like.device, dtype=torch.float32)
        st['used'] = 0
    out = st['buf'][st['used']:st['used'] + n]
    st['used'] += n
    st['total'] += n
    return out


def _bucket_out(param, *shape):
    """the slice of the data-parallel gradient bucket that belongs to `param` (BucketedGradReducer hangs it on the
    parameter), as a FRESH tensor object over that memory -- or None.  A weight-gradient GEMM that writes there makes the
    reducer's pack pass a no-op for that tensor (the FFN / expert weights are 95 % of the gradient bytes), and autograd
    adopts the fresh object as p.grad without a copy (another reference to the same tensor object would make it clone)."""
    v = getattr(param, '_sm3_grad_view', None)
    if v is None or tuple(v.shape) != tuple(shape) or v.dtype != torch.float32 or not v.is_contiguous():
        return None
    # The slice may only be written when autograd will ADOPT the result.  If the parameter still holds a gradient (gradient
    # accumulation, zero_grad(set_to_none=False), or a second backward without reducer.zero_grad(): after finalize() p.grad
    # IS this slice) the GEMM would overwrite the old gradient and AccumulateGrad would then add the slice to itself --
    # 2 * new instead of old + new.  In that case the gradient goes to a fresh tensor and autograd accumulates as usual.
    if getattr(param, 'grad', None) is not None:
        return None
    return v.detach()


def _tn(dy, x, M, N, rows, out=None, **kw):
    """dW[M,N] = dy[rows,M]^T @ x[rows,N] (split-K)."""
    groups = kw.get('num_groups', 1)
    if out is None:
        out = _e(groups, M, N, like=dy) if groups > 1 or kw.get('offsets') is not None else _e(M, N, like=dy)
    gemm(LB.TN, dy, x, out, M, N, rows, **kw)  # split-K factor and fix-up chosen by the library
    return out


def _tn_bias(dy, x, M, N, rows, db, out=None, **kw):
    """dW = dy^T x and db = column sums of dy in one launch: the TN kernel sums the rows of its A operand while it
    loads them (no separate column-sum kernel, no zero-fill, dy is read once).  Only an fp16-STORED dy keeps the
    separate kernel (the C-wide gradients of the AMP data path are fp32)."""
    if dy.dtype == torch.float32:
        return _tn(dy, x, M, N, rows, out=out, colsum_out=db, **kw)
    colsum(dy, rows, M, db, offsets=kw.get('offsets'), num_groups=kw.get('num_groups', 1))
    return _tn(dy, x, M, N, rows, out=out, **kw)


# ------------------------------------------------------------------------------------------------ linear
class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x, w, b = _chk(x, 'x'), _chk(w, 'w'), _chk(b, 'b')
        M, K = x.shape
        N = w.shape[0]
        y = _e(M, N, like=x)
        gemm(LB.NT, x, w, y, M, N, K, epilogue=LB.EPI_BIAS, bias=b)
        ctx.compute = LB.COMPUTE
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, *grads):
        with _compute(ctx.compute):
            return _Linear._backward_impl(ctx, *grads)

    @staticmethod
    def _backward_impl(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x.shape
        N = w.shape[0]
        db = _e(N, like=x)

        dw = _on_side(x.device, lambda: _tn_bias(dy, x, N, K, M, db))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _e(M, K, like=x)
            gemm(LB.NN, dy, w, dx, M, K, N)
        _join_side(x.device)
        return dx, dw, db


def linear(x, w, b):
    """y = x @ w^T + b on the fp32 matrix cores; K (= x.shape[1]) must be a multiple of 32."""
    return _Linear.apply(x, w, b)


class _LinearReLU(Function):
    """y = relu(x @ w^T + b): bias + ReLU in the GEMM epilogue; backward masks dy with the saved output."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w, b = _chk(x, 'x'), _chk(w, 'w'), _chk(b, 'b')
        M, K = x.shape
        N = w.shape[0]
        y = _e(M, N, like=x)
        gemm(LB.NT, x, w, y, M, N, K, epilogue=LB.EPI_BIAS_RELU, bias=b)
        ctx.compute = LB.COMPUTE
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, *grads):
        with _compute(ctx.compute):
            return _LinearReLU._backward_impl(ctx, *grads)

    @staticmethod
    def _backward_impl(ctx, dy):
        x, w, y = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dpre = _e(M, N, like=x)
        call('relu_bwd', dy.contiguous(), y, dpre, M * N)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _e(M, K, like=x)
            gemm(LB.NN, dpre, w, dx, M, K, N)
        db = _e(N, like=x)
        dw = _tn_bias(dpre, x, N, K, M, db)
        return dx, dw, db


def linear_relu(x, w, b):
    """relu(x @ w^T + b) (nn.Linear + ReLU of the RoI head's shared fcs); K and N multiples of 32 / 4."""
    return _LinearReLU.apply(x, w, b)


def stem_patchify(x):
    """(B,3,H,W) NCHW image -> (B*H/4*W/4, 64) patch rows (no gradient: the image is a leaf input)."""
    x = _chk(x, 'image')
    B, Cin, H, W = x.shape
    if Cin != 3:
        raise SM3Error('stem_patchify expects 3 input channels')
    a = _e(B * (H // 4) * (W // 4), 64, like=x)
    call('stem_patchify', x, a, B, H, W)
    return a


# ------------------------------------------------------------------------------------------------ layer norm
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, mode, H, W):
        x, w, b = _chk(x, 'x'), _chk(w, 'w'), _chk(b, 'b')
        T, C = x.shape
        y = _e(T // 4, 4 * C, like=x) if mode == 1 else _e(T, C, like=x)
        mean, rstd = _e(T, like=x), _e(T, like=x)
        call('layernorm_fwd', x, w, b, float(eps), y, mean, rstd, T, C, mode, H, W, nbytes=8.0 * T * C)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.meta = (mode, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        mode, H, W = ctx.meta
        T, C = x.shape
        dx = _e(T, C, like=x)
        dwdb = _e(2, C, like=x)
        ws, nb = LB.row_ws(C, x)
        call('layernorm_bwd', dy.contiguous(), x, w, mean, rstd, dx, dwdb, T, C, mode, H, W, 0, ws, nb,
             nbytes=12.0 * T * C)
        return dx, dwdb[0], dwdb[1], None, None, None, None


def layer_norm(x, w, b, eps, patch_major=False, H=0, W=0):
    return _LayerNorm.apply(x, w, b, eps, 1 if patch_major else 0, H, W)


# ------------------------------------------------------------------------------------------------ shared block pieces
def _half():
    """AMP data path (amp.autocast / wrap_fp16_model): the LayerNorm output that feeds the GEMMs and the three 4C-wide
    tensors of a block (GELU output, GELU', their gradient) live in HBM as fp16 -- what autocast makes of the reference's
    FFN (nn.Linear outputs are half tensors, mmcv/mmcv/runner/fp16_utils.py:71-149).  Weights, biases, the residual stream,
    LayerNorm statistics, router, combine and every C-wide gradient stay fp32.  SM3_AMP_STORAGE=fp32 keeps the round-2
    behaviour (fp32 tensors, operands rounded in the GEMM loader) for A/B measurements."""
    return torch.float16 if (LB.COMPUTE == 1 and AMP_HALF_STORAGE) else torch.float32


AMP_HALF_STORAGE = os.environ.get('SM3_AMP_STORAGE', 'fp16') != 'fp32'
# fp16 SHADOWS of the FFN / expert weights as the B operand of the four GEMMs of a block (what the half model of
# `wrap_fp16_model` holds next to the fp32 master weights): a third fewer operand bytes through the L1, which bounds the
# fp16-operand launches.  ON by default (bit-identical results: the loader rounds the fp32 weights to the
# same halves -- tests/test_amp_gpu.py), 11.85 -> 11.71 ms per AMP step with a cast per forward, less with the shadows kept by
# the optimizer (profiles/r04); SM3_AMP_W16=0 reads the fp32 weights.
AMP_W16 = os.environ.get('SM3_AMP_W16', '1') == '1'


def _shadow(w):
    """the tensor the fp16-operand GEMMs read for weight `w`: `w` itself, or its fp16 shadow.  The shadow is a persistent
    tensor hung on the parameter (`w._sm3_shadow`): cast here when it does not exist yet or `w` was modified by a torch op
    since (`w._version` moved: a checkpoint load, an initialiser, another optimizer), and from then on kept current by
    MultiTensorAdamW, whose update writes the rounded new value next to the fp32 master (optim.hip) -- the training step
    carries no cast pass (0.44 ms per step on config #3 when every forward cast its weights)."""
    if not (LB.COMPUTE == 1 and AMP_HALF_STORAGE and AMP_W16):
        return w
    s = getattr(w, '_sm3_shadow', None)
    if s is not None and w._sm3_shadow_version == w._version and s.device == w.device:
        return s
    if s is None or s.shape != w.shape or s.device != w.device:
        s = torch.empty(w.shape, dtype=torch.float16, device=w.device)
    call('cast_f32_f16', w, s, w.numel(), nbytes=6.0 * w.numel())
    try:
        w._sm3_shadow, w._sm3_shadow_version = s, w._version
    except AttributeError:  # (not a plain tensor object: keep casting per call)
        pass
    return s


# AMP data path: the depthwise output `u` as fp16 in HBM -- what autocast makes of ConvNeXtBlock.depthwise_conv
# (convnext_moe.py:347: nn.Conv2d returns half; F.layer_norm re-promotes, :30-47): the convolution's write, the LayerNorm's
# read and the LayerNorm backward's read of `u` move 2 B per element instead of 4.  LDS-tiled depthwise kernels only
# (C % 32 == 0, H and W multiples of 16: every stage at 1024^2).  OFF by default, SM3_AMP_DW16=1 turns it on: parity-green
# (tests/test_amp_gpu.py bit-equality of the kernels; the full-size AMP case of config #2 passes with it) but it buys nothing --
# same box, config #3: 11.21 ms with `u` in fp32, 11.22 ms in half (depthwise forward 0.806 / 0.806, LayerNorm forward 0.382 /
# 0.379, LayerNorm backward 0.605 / 0.658 ms: these launches are not bound by their bytes, and 8-byte loads per lane are the
# less efficient access) -- profiles/r05/bench_amp_dw16_{0,1}.json.
AMP_DW16 = os.environ.get('SM3_AMP_DW16', '0') == '1'


def _u_half(H, W, C):
    return (AMP_DW16 and _half() == torch.float16 and C % 32 == 0 and H % 16 == 0 and W % 16 == 0
            and H * W * C * 4 < 0x7fff0000)


def _dw_ln_forward(x, w49, bdw, lnw, lnb, eps, B, H, W, C):
    T = B * H * W
    u16 = _u_half(H, W, C)
    u = _e(T, C, like=x, dtype=torch.float16 if u16 else torch.float32)
    call('dwconv7_fwd', x, w49, bdw, None, u, B, H, W, C, 32 if u16 else 0, nbytes=(4.0 + u.element_size()) * T * C)
    hd = _half()
    xn = _e(T, C, like=x, dtype=hd)
    mean, rstd = _e(T, like=x), _e(T, like=x)
    call('layernorm_fwd', u, lnw, lnb, float(eps), xn, mean, rstd, T, C,
         (2 if hd == torch.float16 else 0) | (16 if u16 else 0), H, W,
         nbytes=(u.element_size() + xn.element_size()) * T * C)
    return u, xn, mean, rstd


def _dw_ln_backward(dxn, dout, x, u, w49, lnw, mean, rstd, B, H, W, C):
    """dxn: grad wrt the LN output; dout: grad wrt the block output (residual branch).  Returns
    dx, dw49, dbdw, dlnw, dlnb."""
    T = B * H * W
    du = _e(T, C, like=x)
    dwdb = _e(2, C, like=x)
    ws, nb = LB.row_ws(C, x)
    call('layernorm_bwd', dxn, u, lnw, mean, rstd, du, None, T, C, 16 if u.dtype == torch.float16 else 0, H, W, 0, ws, nb,
         nbytes=(8.0 + u.element_size()) * T * C)
    _deferred_reduce(ws, T, C, 2 * C, dwdb, params=(lnw,))  # d(ln weight) | d(ln bias); joined by the caller
    dwb = _zero_slab(50 * C, x).view(50, C)  # [dw49 (49,C); dbias (C)]: zeros from the pass's arena (one fill per pass)
    dw49, dbdw = dwb[:49], dwb[49]
    dx = _e(T, C, like=x)
    _paired(x.device,
            lambda: call('dwconv7_fwd', du, w49, None, dout, dx, B, H, W, C, 1, nbytes=12.0 * T * C),  # flip=1: reversed taps
            lambda: call('dwconv7_bwd_weight_acc', x, du, dw49, dbdw, B, H, W, C, nbytes=8.0 * T * C),
            T * C, level=3, reads=(x, du, dwb))  # joined by the caller (or at the end of the pass)
    return dx, dw49, dbdw, dwdb[0], dwdb[1]


# ------------------------------------------------------------------------------------------------ dense block
class _DenseBlock(Function):
    @staticmethod
    def forward(ctx, x, w49, bdw, lnw, lnb, w1, b1, w2, b2, gamma, rs, eps, B, H, W):
        x = _chk(x, 'x')
        T, C = x.shape
        Hd = w1.shape[0]
        u, xn, mean, rstd = _dw_ln_forward(x, w49, bdw, lnw, lnb, eps, B, H, W, C)
        hpre, act = _e(T, Hd, like=x, dtype=xn.dtype), _e(T, Hd, like=x, dtype=xn.dtype)
        ctx.wparams = (w1, w2)  # the parameters themselves (their data-parallel bucket slices receive the gradients)
        w1, w2 = _shadow(w1), _shadow(w2)
        gemm(LB.NT, xn, w1, act, T, Hd, C, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=hpre)  # hpre := gelu'(h)
        y, out = _e(T, C, like=x), _e(T, C, like=x)
        gemm(LB.NT, act, w2, out, T, C, Hd, epilogue=LB.EPI_BIAS_SCALE_RES, bias=b2, aux_in=x, aux_out=y,
             gamma=gamma, rowscale=rs, rows_per_scale=H * W)
        ctx.compute, ctx.pair = LB.COMPUTE, PAIR_DGRAD
        ctx.save_for_backward(x, u, mean, rstd, xn, hpre, act, y, w49, lnw, w1, w2, gamma, rs)
        ctx.meta = (B, H, W)
        return out

    @staticmethod
    def backward(ctx, *grads):
        with _compute(ctx.compute, ctx.pair):
            return _DenseBlock._backward_impl(ctx, *grads)

    @staticmethod
    def _backward_impl(ctx, dout):
        x, u, mean, rstd, xn, hpre, act, y, w49, lnw, w1, w2, gamma, rs = ctx.saved_tensors
        B, H, W = ctx.meta
        T, C = x.shape
        Hd = w1.shape[0]
        dout = dout.contiguous()
        dy, dgdb = _e(T, C, like=x), _e(2, C, like=x)
        ws, nb = LB.row_ws(C, x)
        dev = x.device
        call('scale_bwd_prep', dout, y, gamma, rs, H * W, dy, None, T, C, ws, nb, nbytes=12.0 * T * C)
        _deferred_reduce(ws, T, C, 2 * C, dgdb, params=(gamma,))
        dgamma, db2 = dgdb[0], dgdb[1]
        pw1, pw2 = getattr(ctx, 'wparams', (None, None))
        dh, db1 = _e(T, Hd, like=x, dtype=hpre.dtype), _e(Hd, like=x)
        _, dw2 = _paired(dev, lambda: gemm(LB.NN, dy, w2, dh, T, Hd, C, epilogue=LB.EPI_GELU_BWD, aux_in=hpre, colsum_out=db1),
                         lambda: _tn(dy, act, C, Hd, T, out=_bucket_out(pw2, C, Hd)), T * C, level=2, reads=(dy, act),
                         grads_of=(pw1, pw2))
        dxn = _e(T, C, like=x)  # not dy's buffer: the side-stream wgrad may still be reading dy
        _, dw1 = _paired(dev, lambda: gemm(LB.NN, dh, w1, dxn, T, C, Hd),
                         lambda: _tn(dh, xn, Hd, C, T, out=_bucket_out(pw1, Hd, C)), T * C, reads=(dh, xn), grads_of=(pw1, pw2))
        dx, dw49, dbdw, dlnw, dlnb = _dw_ln_backward(dxn, dout, x, u, w49, lnw, mean, rstd, B, H, W, C)
        _join_side(dev)
        return dx, dw49, dbdw, dlnw, dlnb, dw1, db1, dw2, db2, dgamma, None, None, None, None, None


def dense_block(x, w49, bdw, lnw, lnb, w1, b1, w2, b2, gamma, rs, eps, B, H, W):
    return _DenseBlock.apply(x, w49, bdw, lnw, lnb, w1, b1, w2, b2, gamma, rs, eps, B, H, W)


# ------------------------------------------------------------------------------------------------ MoE block
def _pad_rows(n):
    """rows of the fused [Wp; Wn^T; 0] gate matrix: next multiple of 32."""
    return (n + 31) // 32 * 32


class _MoEBlock(Function):
    """One MoE ConvNeXt block.  Inputs are the REFERENCE parameters (cosine_projector weight/bias `wp`/`bp`, `w_noise`
    (C,E), `sim_matrix` (P,E), `temperature`); the fused gate operands are built by one prep launch and the auxiliary
    loss by one more, so a block costs no torch elementwise launches."""

    @staticmethod
    def forward(ctx, x, w49, bdw, lnw, lnb, wp, bp, wn, sim, temp, w1, b1, w2, b2, gamma, rs, noise, eps, B, H, W, k,
                train, clamp_max, loss_coef, forced_topk=None):
        from . import _lib
        x = _chk(x, 'x')
        T, C = x.shape
        E, Hd = w1.shape[0], w1.shape[1]
        linear = sim is None  # gating='linear' (reference :195-196): wp is w_gate (C, E), no bias / sim_matrix / temperature
        if linear:
            # clean logits = x @ w_gate come straight out of the fused projection: wcat = [w_gate^T (E rows, padded to a
            # multiple of 4); w_noise^T; 0].  The (<= 2E x C) operand is assembled with three tiny torch copies -- no
            # SM3Det config uses this gate, so it has no prep kernel of its own.
            P = (E + 3) // 4 * 4
            PC = _pad_rows(P + E)
            wcat, bcat = x.new_zeros(PC, C), x.new_zeros(PC)
            wcat[:E].copy_(wp.t())
            wcat[P:P + E].copy_(wn.t())
            snorm = scale = None
        else:
            P = wp.shape[0]
            PC = _pad_rows(P + E)
            # gate operands: wcat = [Wp; Wn^T; 0], bcat = [bp; 0], snorm = normalize(sim, dim=0), scale = exp(clamp(t))
            wcat, bcat = _e(PC, C, like=x), _e(PC, like=x)
            snorm, scale = _e(P, E, like=x), _e(1, like=x)
            call('moe_gate_prep_fwd', wp.contiguous(), bp.contiguous(), wn.contiguous(), sim.contiguous(), temp,
                 float(clamp_max), P, C, E, PC, wcat, bcat, snorm, scale)
        m = min(k + 1, E)
        S = T * k
        u, xn, mean, rstd = _dw_ln_forward(x, w49, bdw, lnw, lnb, eps, B, H, W, C)
        # gate projection (+ noise projection) in one GEMM: hcat = [h | raw | 0]
        hcat = _e(T, PC, like=x)
        gemm(LB.NT, xn, wcat, hcat, T, PC, C, epilogue=LB.EPI_BIAS, bias=bcat)
        top_idx = _e(T, m, like=x, dtype=torch.int32)
        top_val, gates = _e(T, m, like=x), _e(T, k, like=x)
        clean, hnorm = _e(T, E, like=x), _e(T, like=x)
        sigma = _e(T, E, like=x) if train else None
        nblk = _lib.lib().sm3_moe_router_partial_rows(T)
        partials = _e(nblk, 2 * E, like=x)
        if forced_topk is not None:
            forced_topk = forced_topk.to(device=x.device, dtype=torch.int32).contiguous()
            if forced_topk.shape != (T, k):
                raise SM3Error(f'forced_topk must be ({T}, {k})')
        call('moe_router_fwd', hcat, PC, P, snorm, scale, noise, T, E, k, int(train), top_idx, top_val, gates,
             clean, sigma, hnorm, partials, forced_topk, nbytes=4.0 * T * (PC + 4 * E))
        tot, loss = _e(2 * E, like=x), _e(1, like=x)
        call('moe_aux_loss_fwd', partials, nblk, E, float(loss_coef), tot, loss)
        # dispatch tables (no host sync)
        offsets = _e(E + 1, like=x, dtype=torch.int32)
        slot_token = _e(S, like=x, dtype=torch.int32)
        token_slot = _e(T, k, like=x, dtype=torch.int32)
        nb = _lib.lib().sm3_moe_plan_workspace_bytes(T, E)
        ws = _lib.workspace(nb, x.device)
        call('moe_plan', top_idx, m, T, E, k, offsets, slot_token, token_slot, ws, nb)
        xslot = _e(S, C, like=x, dtype=xn.dtype)
        if xn.dtype == torch.float16:  # a row gather: half rows of C elements move as C / 2 32-bit words
            call('moe_dispatch', xn.view(torch.float32), slot_token, xslot.view(torch.float32), S, C // 2,
                 nbytes=4.0 * S * C)
        else:
            call('moe_dispatch', xn, slot_token, xslot, S, C, nbytes=8.0 * S * C)
        # experts: grouped GEMM pair over the expert-major slots
        hpre, act = _e(S, Hd, like=x, dtype=xn.dtype), _e(S, Hd, like=x, dtype=xn.dtype)
        ctx.wparams = (w1, w2)  # the parameters themselves (their data-parallel bucket slices receive the gradients)
        w1, w2 = _shadow(w1), _shadow(w2)
        gemm(LB.NT, xslot, w1, act, S, Hd, C, epilogue=LB.EPI_BIAS_GELU, bias=b1, aux_out=hpre, offsets=offsets,
             num_groups=E)
        yslot = _e(S, C, like=x)
        gemm(LB.NT, act, w2, yslot, S, C, Hd, epilogue=LB.EPI_BIAS, bias=b2, offsets=offsets, num_groups=E)
        out = _e(T, C, like=x)
        call('moe_combine_fwd', yslot, token_slot, gates, x, gamma, rs, H * W, out, T, C, k,
             nbytes=4.0 * (k + 2) * T * C)
        ctx.compute = LB.COMPUTE
        ctx.save_for_backward(x, u, mean, rstd, xn, hcat, top_idx, top_val, gates, clean, sigma, hnorm, offsets,
                              token_slot, xslot, hpre, act, yslot, w49, lnw, wcat, snorm, scale, w1, w2, gamma, rs,
                              noise, sim, temp, tot)
        ctx.meta = (B, H, W, P, k, train, float(clamp_max), float(loss_coef))
        ctx.pair = PAIR_DGRAD
        ctx.mark_non_differentiable(tot, offsets, top_idx)
        ctx.set_materialize_grads(False)  # else autograd fills a zero tensor per non-differentiable output per block
        return out, loss.reshape(()), tot, offsets, top_idx

    @staticmethod
    def backward(ctx, *grads):
        with _compute(ctx.compute, ctx.pair):
            return _MoEBlock._backward_impl(ctx, *grads)

    @staticmethod
    def _backward_impl(ctx, dout, dloss, _dtot, _doff, _dtop):
        from . import _lib
        (x, u, mean, rstd, xn, hcat, top_idx, top_val, gates, clean, sigma, hnorm, offsets, token_slot, xslot, hpre,
         act, yslot, w49, lnw, wcat, snorm, scale, w1, w2, gamma, rs, noise, sim, temp, tot) = ctx.saved_tensors
        B, H, W, P, k, train, clamp_max, loss_coef = ctx.meta
        T, C = x.shape
        E, Hd = w1.shape[0], w1.shape[1]
        PC = wcat.shape[0]
        S = T * k
        dout = x.new_zeros(T, C) if dout is None else dout.contiguous()
        # aux loss backward -> dimp | dload
        dimp_load = _e(2 * E, like=x)
        if dloss is None:
            dimp_load.zero_()
        else:
            call('moe_aux_loss_bwd', tot, dloss.contiguous().float(), E, loss_coef, dimp_load, dimp_load[E:])
        # combine backward
        dyslot, dgate, dgamma = _e(S, C, like=x), _e(T, k, like=x), _e(C, like=x)
        ws, nb = LB.row_ws(C, x)
        call('moe_combine_bwd', dout, yslot, token_slot, gates, gamma, rs, H * W, dyslot, dgate, None, T, C, k, ws, nb,
             nbytes=4.0 * (2 * k + 1) * T * C)
        _deferred_reduce(ws, T, C, C, dgamma, params=(gamma,))
        # experts backward (every expert gets a -- possibly zero -- gradient: DDP-safe); weight gradients on the side
        # stream, the dx chain on this one
        dev = x.device
        db2 = _e(E, C, like=x)

        pw1, pw2 = getattr(ctx, 'wparams', (None, None))
        dh, db1 = _e(S, Hd, like=x, dtype=hpre.dtype), _e(E, Hd, like=x)
        _, dw2 = _paired(dev, lambda: gemm(LB.NN, dyslot, w2, dh, S, Hd, C, epilogue=LB.EPI_GELU_BWD, aux_in=hpre,
                                           offsets=offsets, num_groups=E, colsum_out=db1),
                         lambda: _tn_bias(dyslot, act, C, Hd, S, db2, out=_bucket_out(pw2, E, C, Hd), offsets=offsets,
                                          num_groups=E), S * C, level=2, reads=(dyslot, act, offsets, db2), grads_of=(pw1, pw2))
        dxslot = _e(S, C, like=x)  # not dyslot's buffer: the side-stream wgrad may still be reading it
        _, dw1 = _paired(dev, lambda: gemm(LB.NN, dh, w1, dxslot, S, C, Hd, offsets=offsets, num_groups=E),
                         lambda: _tn(dh, xslot, Hd, C, S, out=_bucket_out(pw1, E, Hd, C), offsets=offsets, num_groups=E),
                         S * C, reads=(dh, xslot, offsets), grads_of=(pw1, pw2))
        # router backward
        nblk = _lib.lib().sm3_moe_router_partial_rows(T)
        dhcat, dcn, ds_part = _e(T, PC, like=x), _e(T, E, like=x), _e(nblk, like=x, dtype=torch.float64)
        ds_sq = _e(nblk, like=x, dtype=torch.float64) if DEBUG_DSCALE is not None else None
        call('moe_router_bwd', hcat, PC, P, snorm, scale, noise, T, E, k, int(train), top_idx, top_val, gates, clean,
             sigma, hnorm, dgate, dimp_load, dimp_load[E:], dhcat, dcn, ds_part, ds_sq, nbytes=4.0 * T * (2 * PC + 6 * E))
        if DEBUG_DSCALE is not None:  # (per-workgroup sums, per-workgroup sums of the squared per-token terms)
            DEBUG_DSCALE.append((ds_part, ds_sq))
        # gate parameters ([Wp; Wn^T] rows, normalize and exp(clamp) backward in one launch) -- side stream
        if sim is None:  # linear gate: wp is w_gate (C, E)
            dwp, dwn = _e(C, E, like=x), _e(C, E, like=x)
            dbp = dsim = dtemp = None
        else:
            dwp, dbp, dwn = _e(P, C, like=x), _e(P, like=x), _e(C, E, like=x)
            dsim, dtemp = _e(P, E, like=x), _e(1, like=x)

        linear = sim is None

        def gate_wgrad():
            dbcat = _e(PC, like=x)
            dwcat = _tn_bias(dhcat, xn, PC, C, T, dbcat)
            if linear:  # d w_gate = (d hcat[:, :E])^T x,  d w_noise = (d hcat[:, P:P+E])^T x -- rows of dwcat
                dwp.copy_(dwcat[:E].t())
                dwn.copy_(dwcat[P:P + E].t())
                return
            # d snorm-product = h^T . dcn (P x E, reduction over tokens); E is padded to the 16-byte vector width
            E4 = (E + 3) // 4 * 4
            dcn4 = dcn if E4 == E else torch.nn.functional.pad(dcn, (0, E4 - E))
            dsn4 = _e(P, E4, like=x)
            gemm(LB.TN, hcat, dcn4, dsn4, P, E4, T, lda=PC, ldb=E4)
            dsn = dsn4 if E4 == E else dsn4[:, :E].contiguous()
            call('moe_gate_prep_bwd', dwcat, dbcat, dsn, ds_part, nblk, sim.contiguous(), temp, clamp_max, P, C, E,
                 dwp, dbp, dwn, dsim, dtemp)
        dxn = _e(T, C, like=x)

        def gate_dgrad():
            gemm(LB.NN, dhcat, wcat, dxn, T, C, PC)
            call('moe_gather_add', dxslot, token_slot, dxn, T, C, k, 1, nbytes=4.0 * (k + 2) * T * C)
        _paired(dev, gate_dgrad, gate_wgrad, T * C, level=4,
                reads=(dhcat, xn, hcat, dcn, ds_part, sim, temp, dwp, dbp, dwn, dsim, dtemp))
        dx, dw49, dbdw, dlnw, dlnb = _dw_ln_backward(dxn, dout, x, u, w49, lnw, mean, rstd, B, H, W, C)
        _join_side(dev)
        return (dx, dw49, dbdw, dlnw, dlnb, dwp, dbp, dwn, dsim, None if dtemp is None else dtemp.reshape(temp.shape),
                dw1, db1, dw2, db2, dgamma,
                None, None, None, None, None, None, None, None, None, None, None)


def moe_block(x, w49, bdw, lnw, lnb, wp, bp, wn, sim, temp, w1, b1, w2, b2, gamma, rs, noise, eps, B, H, W, k, train,
              clamp_max, loss_coef=1e-2, forced_topk=None):
    """Returns (out (T,C), aux loss (scalar), [importance | load] (2E, non-differentiable), expert slot offsets
    (E+1, int32, non-differentiable), top-(k+1) expert indices per token (T, min(k+1,E), int32, non-differentiable))."""
    return _MoEBlock.apply(x, w49, bdw, lnw, lnb, wp, bp, wn, sim, temp, w1, b1, w2, b2, gamma, rs, noise, eps, B, H,
                           W, k, train, clamp_max, loss_coef, forced_topk)
