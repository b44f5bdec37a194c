"""Generate sm3det_amd/configs/baseline_configs.json from the reference's own config files.

Run in the build container only (the GPU box has no /root/reference):

    python scripts/make_bench_configs.py

For each BASELINE.json config the file under /root/reference/local_configs is read UNCHANGED by
sm3det_amd.config.Config.fromfile (python exec + `_base_` merge) and the parts `bench.py --config NAME` needs are
stored: the whole `model` dict (backbone / neck / heads / train_cfg exactly as written), `fp16` (the AMP switch),
`optimizer`, `optimizer_config`, `lr_config` (the dynamic-lr policy), `data.samples_per_gpu`.  tests/test_config_cpu.py re-derives the file from the live
reference and compares, so the committed copy cannot drift.
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('SM3DET_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'sm3det_amd', 'configs', 'baseline_configs.json')

# name on the bench command line -> (BASELINE.json config #, file pattern under local_configs/)
CONFIGS = {
    'simple_joint': (1, 'main_convnext_t_orcnn_gfl_simple_joint.py'),
    'main_SM3Det': (2, 'main_SM3Det.py'),
    'SM3Det_convnext_t': (3, 'SM3Det_convnext_t.py'),
    'e16t2': (4, 'ablation_moe_et_*e16t2_last2blocks.py'),
    'SM3Det_convnext_b': (5, 'SM3Det_convnext_b.py'),
}


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def derive():
    from sm3det_amd.config import Config
    out = {}
    for name, (num, pat) in CONFIGS.items():
        files = sorted(glob.glob(os.path.join(REF, 'local_configs', pat)))
        assert len(files) == 1, (pat, files)
        cfg = Config.fromfile(files[0])
        out[name] = dict(
            baseline_config=num, file='local_configs/' + os.path.basename(files[0]),
            model=_plain(cfg['model']), fp16=_plain(cfg.get('fp16')), optimizer=_plain(cfg.get('optimizer')),
            optimizer_config=_plain(cfg.get('optimizer_config')),
            lr_config=_plain(cfg.get('lr_config')),
            samples_per_gpu=(cfg.get('data') or {}).get('samples_per_gpu'))
    return out


if __name__ == '__main__':
    cfgs = derive()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, 'w') as f:
        json.dump(cfgs, f, indent=1, sort_keys=True)
        f.write('\n')
    for n, c in cfgs.items():
        bb = c['model']['backbone']
        print(f"{n}: cfg #{c['baseline_config']} {c['file']} arch={bb['arch']} E={bb.get('num_experts')} "
              f"k={bb.get('top_k')} fp16={c['fp16']} opt={c['optimizer']} clip={c['optimizer_config']} "
              f"bs/gpu={c['samples_per_gpu']}")
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')
