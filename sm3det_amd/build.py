"""Build libsm3det_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

    python -m sm3det_amd.build [-f] [-v]

One object per ``csrc/*.hip`` (rebuilt only when the source or a header is newer), linked into
``sm3det_amd/csrc/libsm3det_hip.so``.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libsm3det_hip.so')
ARCH = 'gfx950'

BASE_FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# Per-file flags.  The detection ops must round exactly like the reference's scalar CPU code (bit-exact NMS),
# so no FMA contraction there; the backbone kernels keep the default (-ffp-contract=fast-honor-pragmas).
FILE_FLAGS = {
    'ops_rotated.hip': ['-ffp-contract=off'],
    'deform_conv.hip': ['-ffp-contract=off'],
    'rpn.hip': ['-ffp-contract=off'],
    'deform_fused.hip': ['-ffp-contract=off'],
    # bf16x3 GEMMs: the operand split is scalar fp32 arithmetic beside MFMAs; SLP-packing it into v_pk_add_f32 costs more issue
    # time than the scalar form (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
    'gemm_b3_nt.hip': ['-fno-slp-vectorize'], 'gemm_b3_nn.hip': ['-fno-slp-vectorize'], 'gemm_b3_tn.hip': ['-fno-slp-vectorize'],
}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


# A/B variants: the same ABI compiled with an extra define into its own objects and `libsm3det_hip_<name>.so`; selected at
# run time with SM3DET_HIP_LIB=<path> (sm3det_amd/_lib.py).  Measurement aids, never loaded by default.
VARIANTS = {
    'gelu_exact': ['-DSM3_GELU_EXACT=1'],  # GELU epilogues through ocml erff/expf instead of the A&S 7.1.26 polynomial
    'f16_occ2': ['-DSM3_F16_OCC=2'],       # fp16-operand GEMMs at two workgroups per CU (round-2 occupancy)
    # phase ablations of the fp16-operand GEMM loop (results are WRONG by construction: timing only)
    'abl_noload': ['-DSM3_ABL_NOLOAD=1'], 'abl_nostore': ['-DSM3_ABL_NOSTORE=1'], 'abl_nomfma': ['-DSM3_ABL_NOMFMA=1'],
    # per-workgroup phase timestamps of the GEMM kernel (scripts/gemm_trace.py) / static wave priority by residency slot
    'trace': ['-DSM3_TRACE=1'], 'prio1': ['-DSM3_PRIO=1'], 'prio2': ['-DSM3_PRIO=2'], 'stagger': ['-DSM3_STAGGER=1'],
    'trace_stagger': ['-DSM3_TRACE=1', '-DSM3_STAGGER=1'],
    'lpt8': ['-DSM3_ROUTER_LPT=8'], 'lpt2': ['-DSM3_ROUTER_LPT=2'],  # lanes per token of the MoE router kernels
    'abl_nocvt': ['-DSM3_ABL_NOCVT=1'],  # bf16x3 loop without the split arithmetic (stores kept)
    'b3_nomix': ['-DSM3_B3_NOMIX=1'], 'b3_occ2': ['-DSM3_B3_OCC=2'], 'b3_unsigned': ['-DSM3_B3_SIGNED=0'], 'f16_spills': ['-DSM3_F16_KEEP_SPILLS=1'], 'aux_temporal': ['-DSM3_AUX_TEMPORAL=1'], 'b3_slp': ['-fslp-vectorize'], 'b3_two_sets': ['-DSM3_B3_SIGNED=1'], 'dw_onetile': ['-DSM3_DW_MULTITILE=0'], 'dw_wgrad_1024': ['-DSM3_DW_WGRAD_ONE_ROUND=0'],
 # bf16x3 GEMMs at three workgroups per CU (register budget 168 instead of 256)
    'abl_noepi': ['-DSM3_ABL_NOEPI=1'], 'abl_loop_only_mfma': ['-DSM3_ABL_NOLOAD=1', '-DSM3_ABL_NOSTORE=1', '-DSM3_ABL_NOEPI=1'],
}


def build_variant(name, verbose=False, only=None):
    """only: substring of the translation units the variant's define touches -- the others are linked from the main build's
    objects (python -m sm3det_amd.build --variant b3_occ3 --only gemm_b3)"""
    flags = VARIANTS[name]
    vdir = os.path.join(CSRC, '_variant_' + name)
    os.makedirs(vdir, exist_ok=True)
    lib = os.path.join(CSRC, f'libsm3det_hip_{name}.so')
    objs, procs = [], []
    for s in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
        if only and only not in os.path.basename(s):
            objs.append(s[:-4] + '.o')
            continue
        o = os.path.join(vdir, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        cmd = [_hipcc()] + BASE_FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + flags + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n{out.decode()}')
    subprocess.check_call([_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', lib] + objs)
    return lib


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdrs = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    hdr_m = max([os.path.getmtime(h) for h in hdrs] + [os.path.getmtime(__file__)])
    objs, rebuilt = [], False
    procs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            cmd = [_hipcc()] + BASE_FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n{out.decode()}')
        if verbose and out:
            print(out.decode())
    if rebuilt or not os.path.exists(LIB):
        cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    if '--variant' in sys.argv:
        print(build_variant(sys.argv[sys.argv.index('--variant') + 1], verbose='-v' in sys.argv,
                            only=sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else None))
    else:
        print(build(force='-f' in sys.argv, verbose='-v' in sys.argv))
