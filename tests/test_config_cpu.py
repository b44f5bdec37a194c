"""CPU tier: the plug-in surface against the REAL config files of the reference (north_star: "so
local_configs/main_SM3Det.py still constructs ... the model unchanged").  The files are read from /root/reference
unmodified by sm3det_amd.config.Config.fromfile (python exec + `_base_` merge + `_delete_`, mmcv semantics) and every
`model` sub-dict whose `type` this package implements is built through the registry with the dict UNCHANGED (unknown
keys such as loss_cls / anchor_generator / init_cfg included).  Skipped where /root/reference is absent (GPU box)."""
import os

import pytest
import torch

REF = os.environ.get('SM3DET_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'local_configs')),
                                reason='/root/reference not present')


def _load(name):
    from sm3det_amd.config import Config
    return Config.fromfile(os.path.join(REF, 'local_configs', name))


def test_main_sm3det_config_parses_with_bases_and_builds_every_implemented_piece():
    from sm3det_amd.config import build_detector_pieces
    from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
    cfg = _load('main_SM3Det.py')
    # values that only exist after python evaluation / _base_ merging
    assert cfg.model.type == 'TriSourceDetector'
    assert cfg.model.backbone.MoE_Block_inds == [[], [0, 2], [0, 2, 4, 6, 8], [0, 2]]
    assert cfg.model.sar_bbox_head.num_classes == 26 and cfg.angle_version == 'le90'
    assert 'data' in cfg and 'optimizer' in cfg and 'lr_config' in cfg  # from the three _base_ files
    assert cfg.optimizer.type == 'AdamW'
    pieces = build_detector_pieces(cfg.model)
    for name in ('backbone', 'neck', 'rgb_rpn_head', 'ifr_rpn_head', 'rgb_roi_head.bbox_roi_extractor',
                 'rgb_roi_head.bbox_head', 'ifr_roi_head.bbox_roi_extractor', 'ifr_roi_head.bbox_head'):
        assert name in pieces, (name, sorted(pieces))
    bb = pieces['backbone']
    assert isinstance(bb, ConvNeXt_moe_MultiInput)
    assert bb.init_cfg['checkpoint'].endswith('convnext-tiny.pth')  # carried, not loaded, until init_weights()
    n_moe = sum(1 for st in bb.stages for b in st if b.MoE_cfg is not None)
    assert n_moe == 9
    assert pieces['neck'].num_outs == 5
    assert pieces['rgb_rpn_head'].num_anchors == 3
    assert pieces['rgb_roi_head.bbox_head'].fc_cls.out_features == 27
    assert pieces['rgb_roi_head.bbox_roi_extractor'].featmap_strides == [4, 8, 16, 32]
    if 'sar_bbox_head' in pieces:  # GFL tower (mmdet semantics restated; parity unpinned)
        assert pieces['sar_bbox_head'].num_classes == 26


@pytest.mark.parametrize('name,arch,experts,n_moe', [
    ('SM3Det_convnext_t.py', 'tiny', 8, 9),                                   # BASELINE config #3 (AMP)
    ('SM3Det_convnext_b.py', 'base', 8, None),                                # config #5: `_base_` needs the fallback
    ('main_convnext_t_orcnn_gfl_simple_joint.py', 'tiny', None, 0),           # config #1: fully dumped, no _base_
])
def test_other_baseline_configs_parse_and_build_backbone(name, arch, experts, n_moe):
    from sm3det_amd.config import build_detector_pieces
    cfg = _load(name)
    assert cfg.model.backbone.type == 'ConvNeXt_moe_MultiInput' and cfg.model.backbone.arch == arch
    pieces = build_detector_pieces(cfg.model)
    bb = pieces['backbone']
    got = sum(1 for st in bb.stages for b in st if b.MoE_cfg is not None)
    if n_moe is not None:
        assert got == n_moe
    else:
        assert got > 9
    if experts is not None:
        assert bb.num_experts == experts
    if name.startswith('SM3Det_convnext'):
        assert cfg.fp16 == dict(loss_scale='dynamic')  # the AMP switch of configs #3 / #5
        assert 'data' in cfg  # bases resolved (for _b through the documented fallback)


def test_e16_ablation_config_parses():
    import glob
    files = glob.glob(os.path.join(REF, 'local_configs', 'ablation_moe_et_*e16t2_last2blocks.py'))
    assert files, 'config #4 file not found'
    from sm3det_amd.config import Config, build_detector_pieces
    cfg = Config.fromfile(files[0])
    assert cfg.model.backbone.num_experts == 16 and cfg.model.backbone.top_k == 2
    bb = build_detector_pieces(cfg.model)['backbone']
    assert bb.num_experts == 16


def test_delete_key_and_duplicate_base_keys(tmp_path):
    from sm3det_amd.config import Config
    (tmp_path / 'a.py').write_text("model = dict(head=dict(type='A', x=1, y=2))\nlr = 0.1\n")
    (tmp_path / 'b.py').write_text("_base_ = './a.py'\nmodel = dict(head=dict(_delete_=True, type='B', z=3))\n"
                                   "ks = [i * 2 for i in range(3)]\n")
    c = Config.fromfile(str(tmp_path / 'b.py'))
    assert c.model.head == dict(type='B', z=3) and c.lr == 0.1 and c.ks == [0, 2, 4]
    (tmp_path / 'c.py').write_text("_base_ = ['./a.py', './a2.py']\n")
    (tmp_path / 'a2.py').write_text("lr = 0.2\n")
    with pytest.raises(KeyError):
        Config.fromfile(str(tmp_path / 'c.py'))


def test_committed_bench_config_fixture_equals_the_live_reference_configs():
    """sm3det_amd/configs/baseline_configs.json (what `bench.py --config` reads on the GPU box, where the reference tree
    is absent) is exactly what scripts/make_bench_configs.py derives from the reference files today."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('make_bench_configs', os.path.join(root, 'scripts', 'make_bench_configs.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    with open(m.OUT) as f:
        committed = json.load(f)
    assert committed == json.loads(json.dumps(m.derive()))
    assert committed['main_SM3Det']['model']['backbone']['MoE_Block_inds'] == [[], [0, 2], [0, 2, 4, 6, 8], [0, 2]]
    assert committed['SM3Det_convnext_b']['fp16'] == {'loss_scale': 'dynamic'}
    assert committed['e16t2']['model']['backbone']['num_experts'] == 16


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v) for v in x)
    return x


def test_config_loader_equals_the_references_own_mmcv_config_fromfile():
    """Every file of local_configs/ through the REFERENCE'S OWN loader -- mmcv/mmcv/utils/config.py `Config.fromfile`,
    imported unmodified (oracle/ref_config.py; only `addict.Dict` and `yapf`, absent from this image, are stand-ins) -- and
    through sm3det_amd.config.Config.fromfile: the merged dicts are identical, key for key, value for value and container
    type for container type.  Files whose `_base_` path does not exist relative to the file make the reference loader
    raise FileNotFoundError; those are the ones this package's documented fallback resolves."""
    import glob

    from oracle import ref_config as RC
    from sm3det_amd.config import Config
    if not RC.available():
        pytest.skip('mmcv/utils/config.py not present')
    ref_mod = RC.load()
    files = sorted(glob.glob(os.path.join(REF, 'local_configs', '*.py')))
    assert len(files) >= 30
    same, fallback = 0, []
    for f in files:
        mine = _plain(dict(Config.fromfile(f)))
        mine.pop('filename', None)
        try:
            ref = _plain(ref_mod.Config.fromfile(f)._cfg_dict)
        except FileNotFoundError as e:
            assert '_base_' in str(e), (f, e)
            fallback.append(os.path.basename(f))
            assert 'model' in mine  # resolved through <tree>/configs/_base_/...
            continue
        assert mine == ref, os.path.basename(f)
        same += 1
    assert same >= 30 and 'main_SM3Det.py' not in fallback and 'SM3Det_convnext_t.py' not in fallback
    assert 'SM3Det_convnext_b.py' in fallback
