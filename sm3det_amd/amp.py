"""Mixed precision on the MI355X hot path -- the mirror of the reference's AMP switch.

Reference: ``fp16 = dict(loss_scale='dynamic')`` (local_configs/SM3Det_convnext_t.py:8, SM3Det_convnext_b.py:8) makes
``Fp16OptimizerHook`` (mmcv/mmcv/runner/hooks/optimizer.py:198-306) call ``wrap_fp16_model`` -- every module that has
``fp16_enabled`` gets it set -- and ``@auto_fp16`` (mmcv/mmcv/runner/fp16_utils.py:71-149) then runs the detector's
forward under ``torch.cuda.amp.autocast``: ``nn.Linear`` / conv inputs are rounded to fp16 and multiplied with fp32
accumulation, LayerNorm / softmax / the MoE combine (``.float()``, convnext_moe.py:279-283) stay fp32, the loss is
scaled by a ``GradScaler`` and the optimizer step is skipped when a gradient overflowed.

Here: ``autocast()`` switches every GEMM of the family to fp16 OPERANDS with fp32 accumulation
(``v_mfma_f32_32x32x16_f16``, 16x the fp32 matrix rate) and -- since round 3 -- to the fp16 DATA PATH: the tensors that
autocast turns into halves in the reference (the outputs of the FFN's ``nn.Linear``s and what feeds them) are stored as
fp16 in HBM: the LayerNorm output ``xn``, the dispatched expert inputs ``xslot``, the GELU output ``act``, the saved
GELU' ``hpre`` and the 4C-wide gradient ``dh`` (written as halves by the producing kernels' epilogues, read as halves by
the GEMM loaders).  Master weights, biases, the residual stream, every C-wide gradient, the expert
outputs and everything the router computes stay fp32; fp32 operands are rounded in the GEMM's loader, so no cast kernels
exist.  ``SM3_AMP_STORAGE=fp32`` restores the all-fp32 storage of round 2 for A/B runs.  The dynamic loss scale (GradScaler semantics: init 65536, x0.5 on overflow, x2 after 2000 clean steps) lives on the device
inside ``MultiTensorAdamW`` (optim.py): unscale, overflow check, clip, skip and scale update cost no host sync.
"""
import contextlib

from . import _lib_backbone as LB


@contextlib.contextmanager
def autocast(enabled=True):
    """Run the enclosed forward AND backward GEMMs with fp16 operands (the flag is read when a GEMM is launched, so wrap
    ``loss.backward()`` as well -- or use ``wrap_fp16_model`` + the module's own forward, which records the mode for
    its backward)."""
    old = LB.COMPUTE
    LB.COMPUTE = 1 if enabled else 0
    try:
        yield
    finally:
        LB.COMPUTE = old


def wrap_fp16_model(model):
    """mmcv.runner.wrap_fp16_model: flag every module that declares ``fp16_enabled``."""
    for m in model.modules():
        if hasattr(m, 'fp16_enabled'):
            m.fp16_enabled = True
    return model
