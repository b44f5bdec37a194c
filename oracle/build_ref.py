"""Build recipe for ``oracle/_ref``: the REFERENCE's own CPU operators, compiled
from the sources where they lie under ``/root/reference`` (never copied into this
repo).  TEST INFRASTRUCTURE ONLY -- nothing under ``sm3det_amd/`` imports this.

What is compiled (all paths relative to
``/root/reference/mmcv/mmcv/ops/csrc``):

* ``pytorch/cpu/{box_iou_rotated,nms_rotated,roi_align_rotated,nms,deform_conv}.cpp``
* ``pytorch/{box_iou_rotated,roi_align_rotated,nms,deform_conv}.cpp`` (dispatchers)
* headers from ``common/`` (``box_iou_rotated_utils.hpp``, ``pytorch_cpp_helper.hpp``,
  ``pytorch_device_registry.hpp``)

The reference's own build system (``mmcv/setup.py``) is NOT run.  Same-named files
in ``pytorch/`` and ``pytorch/cpu/`` would collide as object names, so each one is
pulled in through a generated one-line ``#include "<abs path>"`` wrapper written
into ``oracle/_ref/src`` (git-ignored).  A ~30-line pybind shim (generated too)
re-declares the entry points the way ``pytorch/pybind.cpp`` does for these ops.

Output: ``oracle/_ref/sm3det_ref_ops.so`` (git-ignored, NOT gpurun-ignored: it
travels to the GPU box, where ``/root/reference`` does not exist).

Also compiled (``py_compile``, bytecode only -- no source text is copied) into
``oracle/_ref/pyc``: the reference's own PYTHON wrappers of these ops,
``mmcv/mmcv/ops/{box_iou_rotated,nms,roi_align_rotated,deform_conv}.py`` plus the two
helper files they import (``mmcv/mmcv/utils/{ext_loader,misc}.py``), so that the GPU
tests can drive the REFERENCE's unmodified wrappers into ``sm3det_amd.mmcv_ext`` on a box
where ``/root/reference`` does not exist (``oracle/ref_mmcv_ops.py`` loads them).

Usage:  python oracle/build_ref.py          (no-op when the .so is up to date or
                                             /root/reference is absent)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get('SM3DET_REFERENCE', '/root/reference')
CSRC = os.path.join(REF_ROOT, 'mmcv', 'mmcv', 'ops', 'csrc')
OUT = os.path.join(HERE, '_ref')
NAME = 'sm3det_ref_ops'

CPU_UNITS = ['box_iou_rotated', 'nms_rotated', 'roi_align_rotated', 'nms',
             'deform_conv']
DISPATCH_UNITS = ['box_iou_rotated', 'roi_align_rotated', 'nms', 'deform_conv']

SHIM = r'''
#include <torch/extension.h>
using at::Tensor;
// declarations mirror mmcv/mmcv/ops/csrc/pytorch/pybind.cpp (lines cited in include/sm3det_hip.h)
void box_iou_rotated(const Tensor boxes1, const Tensor boxes2, Tensor ious,
                     const int mode_flag, const bool aligned);
Tensor nms_rotated_cpu(const Tensor dets, const Tensor scores, const float iou_threshold);
Tensor nms(Tensor boxes, Tensor scores, float iou_threshold, int offset);
void roi_align_rotated_forward(Tensor input, Tensor rois, Tensor output, int pooled_height,
                               int pooled_width, float spatial_scale, int sampling_ratio,
                               bool aligned, bool clockwise);
void roi_align_rotated_backward(Tensor grad_output, Tensor rois, Tensor grad_input,
                                int pooled_height, int pooled_width, float spatial_scale,
                                int sampling_ratio, bool aligned, bool clockwise);
void deform_conv_forward(Tensor input, Tensor weight, Tensor offset, Tensor output,
                         Tensor columns, Tensor ones, int kW, int kH, int dW, int dH, int padW,
                         int padH, int dilationW, int dilationH, int group,
                         int deformable_group, int im2col_step);
void deform_conv_backward_input(Tensor input, Tensor offset, Tensor gradOutput,
                                Tensor gradInput, Tensor gradOffset, Tensor weight,
                                Tensor columns, int kW, int kH, int dW, int dH, int padW,
                                int padH, int dilationW, int dilationH, int group,
                                int deformable_group, int im2col_step);
void deform_conv_backward_parameters(Tensor input, Tensor offset, Tensor gradOutput,
                                     Tensor gradWeight, Tensor columns, Tensor ones, int kW,
                                     int kH, int dW, int dH, int padW, int padH, int dilationW,
                                     int dilationH, int group, int deformable_group,
                                     float scale, int im2col_step);
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("box_iou_rotated", &box_iou_rotated);
  m.def("nms_rotated_cpu", &nms_rotated_cpu);
  m.def("nms", &nms);
  m.def("roi_align_rotated_forward", &roi_align_rotated_forward);
  m.def("roi_align_rotated_backward", &roi_align_rotated_backward);
  m.def("deform_conv_forward", &deform_conv_forward);
  m.def("deform_conv_backward_input", &deform_conv_backward_input);
  m.def("deform_conv_backward_parameters", &deform_conv_backward_parameters);
}
'''


PY_UNITS = {  # module name in the shim package -> path under /root/reference/mmcv/mmcv
    'mmcv.utils.ext_loader': 'utils/ext_loader.py',
    'mmcv.utils.misc': 'utils/misc.py',
    'mmcv.ops.box_iou_rotated': 'ops/box_iou_rotated.py',
    'mmcv.ops.nms': 'ops/nms.py',
    'mmcv.ops.roi_align_rotated': 'ops/roi_align_rotated.py',
    'mmcv.ops.deform_conv': 'ops/deform_conv.py',
}
PYC = os.path.join(OUT, 'pyc')
BACKBONE_PYC = 'mmrotate.models.backbones.convnext_moe.pyc'


def build_pyc():
    """bytecode of the reference's python wrappers -> oracle/_ref/pyc/<module>.pyc (needs /root/reference)"""
    import py_compile
    root = os.path.join(REF_ROOT, 'mmcv', 'mmcv')
    if not os.path.isdir(root):
        return None
    os.makedirs(PYC, exist_ok=True)
    for mod, rel in PY_UNITS.items():
        src, dst = os.path.join(root, rel), os.path.join(PYC, mod + '.pyc')
        if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            py_compile.compile(src, cfile=dst, dfile=f'<reference>/mmcv/mmcv/{rel}', doraise=True)
    # the reference BACKBONE module itself (oracle/ref_moe.py loads it from this bytecode where /root/reference is absent:
    # bench.py's cpu_baseline then times the reference's own code on the GPU box's host -- kind "reference")
    src = os.path.join(REF_ROOT, 'mmrotate', 'models', 'backbones', 'convnext_moe.py')
    dst = os.path.join(PYC, BACKBONE_PYC)
    if os.path.exists(src) and (not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src)):
        py_compile.compile(src, cfile=dst, dfile='<reference>/mmrotate/models/backbones/convnext_moe.py', doraise=True)
    return PYC


def so_path():
    return os.path.join(OUT, NAME + '.so')


def build(verbose=False):
    """Compile oracle/_ref if the reference tree is present. Returns the .so path or None."""
    build_pyc()
    if not os.path.isdir(CSRC):
        return so_path() if os.path.exists(so_path()) else None
    if os.path.exists(so_path()):
        newest = max(os.path.getmtime(os.path.join(CSRC, 'pytorch', 'cpu', u + '.cpp'))
                     for u in CPU_UNITS)
        if os.path.getmtime(so_path()) > max(newest, os.path.getmtime(__file__)):
            return so_path()
    src = os.path.join(OUT, 'src')
    os.makedirs(src, exist_ok=True)
    sources = []
    for u in CPU_UNITS:
        p = os.path.join(src, f'ref_cpu_{u}.cpp')
        with open(p, 'w') as f:
            f.write(f'#include "{CSRC}/pytorch/cpu/{u}.cpp"\n')
        sources.append(p)
    for u in DISPATCH_UNITS:
        p = os.path.join(src, f'ref_dispatch_{u}.cpp')
        with open(p, 'w') as f:
            f.write(f'#include "{CSRC}/pytorch/{u}.cpp"\n')
        sources.append(p)
    # pytorch/nms_rotated.cpp only forwards to nms_rotated_cpu (file:line 17-32); the shim binds
    # nms_rotated_cpu directly, exactly what that dispatcher calls for CPU tensors.
    p = os.path.join(src, 'ref_shim.cpp')
    with open(p, 'w') as f:
        f.write(SHIM)
    sources.append(p)
    from torch.utils.cpp_extension import load
    load(name=NAME, sources=sources, extra_include_paths=[os.path.join(CSRC, 'common')],
         extra_cflags=['-O2', '-w'], build_directory=OUT, verbose=verbose, is_python_module=False)
    return so_path()


def load_ref():
    """Import the compiled reference ops (python module). Raises if it was never built."""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    path = so_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f'{path} missing: run `python oracle/build_ref.py` where '
                                f'{REF_ROOT} exists')
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    out = build(verbose='-v' in sys.argv)
    print('oracle/_ref:', out)
