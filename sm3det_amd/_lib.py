"""ctypes binding of ``libsm3det_hip.so`` (the C ABI declared in ``include/sm3det_hip.h``).

PyTorch is used here only as plumbing: device memory (``Tensor.data_ptr()``), the current HIP stream and
(elsewhere) ``torch.distributed``.  There is NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing an operator fails loudly -- a product path must never silently run on eager PyTorch or on
the CPU oracle.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SM3DET_HIP_LIB: A/B aid for kernel work (another build of the same ABI); never a fallback -- it must exist and resolve
LIB_PATH = os.environ.get('SM3DET_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libsm3det_hip.so')

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t


class SM3Error(RuntimeError):
    pass


_lib = None


def _declare(lib):
    P, I, F, S = c_void_p, c_int, c_float, c_size_t
    sig = {
        'sm3_version': (ctypes.c_char_p, []),
        'sm3_compiler_version': (ctypes.c_char_p, []),
        'sm3_error_string': (ctypes.c_char_p, [I]),
        'sm3_box_iou_rotated': (I, [P, P, P, I, I, I, I, P]),
        'sm3_transpose_f32': (I, [P, P, I, I, I, P]),
        'sm3_transpose_add_f32': (I, [P, P, I, I, I, P]),
        'sm3_max_iou_assign_workspace_bytes': (S, [I, I]),
        'sm3_max_iou_assign': (I, [P, I, I, P, I, I, I, F, F, F, I, P, P, P, P, P, S, P]),
        'sm3_max_iou_assign_masked': (I, [P, I, I, P, P, I, I, I, F, F, F, I, P, P, P, P, P, S, P]),
        'sm3_argsort_desc_workspace_bytes': (S, [I]),
        'sm3_argsort_desc_f32': (I, [P, I, P, P, S, P]),
        'sm3_topk_desc_workspace_bytes': (S, [I]),
        'sm3_topk_desc_f32': (I, [P, I, I, P, P, S, P]),
        'sm3_nms_workspace_bytes': (S, [I]),
        'sm3_nms': (I, [P, P, P, I, F, I, P, P, P, S, P]),
        'sm3_nms_rotated_workspace_bytes': (S, [I]),
        'sm3_nms_rotated': (I, [P, I, P, P, I, F, I, P, P, P, S, P]),
        'sm3_roi_align_rotated_forward': (I, [P, P, P, I, I, I, I, I, I, I, F, I, I, I, I, P]),
        'sm3_roi_align_rotated_backward': (I, [P, P, P, I, I, I, I, I, I, I, F, I, I, I, I, P]),
        'sm3_roi_align_rotated_multilevel_forward': (I, [P, P, P, P, I, F, P, P, P, I, I, I, I, I, I, I, I, P]),
        'sm3_roi_align_rotated_multilevel_backward': (I, [P, P, P, P, P, P, I, F, I, I, I, I, I, I, I, I, P]),
        'sm3_random_sample_workspace_bytes': (S, []),
        'sm3_rcnn_gather_samples': (I, [P, P, I, I, P, I, P, P, P, P, P, I, I, F, ctypes.c_long, P, P, P, P, P]),
        'sm3_random_sample_fixed': (I, [P, P, I, I, I, F, P, P, P, P, P, P, S, P]),
        'sm3_roi_align_rotated_backward_tiled_workspace_bytes': (S, [I, I, I, I, I, I, P, P, I]),
        'sm3_roi_align_rotated_backward_tiled': (I, [P, P, P, P, P, P, I, F, I, I, I, I, I, I, I, I, I, P, S, P]),
        'sm3_atss_assign': (I, [P, I, P, P, I, P, I, P, I, P, P]),
        'sm3_atss_decode': (I, [P, I, P, P, P]),
        'sm3_gfl_loss_fwd': (I, [P, P, P, I, I, I, I, P, P, I, P, P, P, P, P, F, F, P, P, P]),
        'sm3_gfl_loss_bwd': (I, [P, P, P, I, I, I, I, P, P, I, P, P, P, P, P, F, F, P, P, P, P]),
    }
    from . import _lib_backbone, det_losses
    sig.update(_lib_backbone.signatures())
    sig.update(det_losses.signatures())
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError => the .so is stale: fail loudly
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTED = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the extension was not built."""
    global _lib, EXPORTED
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SM3Error(f'{LIB_PATH} not found: run `python -m sm3det_amd.build` '
                           '(hipcc --offload-arch=gfx950); there is no CPU/eager fallback')
        handle = ctypes.CDLL(LIB_PATH)
        EXPORTED = _declare(handle)
        _lib = handle
        # the 3x3 convolutions of the neck / heads follow the arithmetic of the fp32 GEMM family (SM3_GEMM_ARITH: 'bf16x3'
        # by default, 'f32' = the native fp32 matrix instruction) -- _lib_backbone.ARITH32
        from . import _lib_backbone
        handle.sm3_conv3x3_set_arith(_lib_backbone.ARITH32)
    return _lib


def check(code, what):
    if code != 0:
        raise SM3Error(f'{what}: {lib().sm3_error_string(code).decode()} (code {code})')


def stream_ptr():
    """Raw hipStream_t of torch's current stream on the current device."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SM3Error('sm3det_amd operators run on the MI355X only: got a CPU tensor '
                           '(there is deliberately no CPU fallback in the product path)')


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
