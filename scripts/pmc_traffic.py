"""Per-kernel HBM traffic from the two PMC passes (FETCH_SIZE x2 on gfx950 for wide coalesced reads + WRITE_SIZE, both
in KB: MI355X_MICROARCH.md, HBM / rocprofv3 section) next to the ALGORITHMIC bytes of the bench line, and the MFMA-pipe
utilisation of the GEMM family from the SQ pass.

    python scripts/pmc_traffic.py FETCH_summary.json WRITE_summary.json bench.json [mfma_summary.json] > pmc_traffic.json
"""
import json
import sys

def _load(path):
    d = json.load(open(path))
    return {k.replace('sm3gemm::', ''): v for k, v in d.items()} if isinstance(d, dict) and 'metric' not in d else d


fetch, write, bench = (_load(a) for a in sys.argv[1:4])
mfma = _load(sys.argv[4]) if len(sys.argv) > 4 else None
# rocprof kernel name -> name of the C-ABI call in bench.py's kernels_ms_per_step
ALIAS = {'gemm_f32_kernel': 'gemm_f32', 'dwconv7_lds_fwd_kernel': 'dwconv7_fwd',
         'dwconv7_lds_bwd_weight_kernel': 'dwconv7_bwd_weight', 'layernorm_fwd_kernel': 'layernorm_fwd',
         'layernorm_bwd_kernel': 'layernorm_bwd', 'moe_router_fwd_kernel': 'moe_router_fwd',
         'moe_router_bwd_kernel': 'moe_router_bwd', 'moe_combine_fwd_kernel': 'moe_combine_fwd',
         'moe_combine_bwd_kernel': 'moe_combine_bwd', 'adamw_multi_kernel': 'adamw_multi',
         'scale_bwd_prep_kernel': 'scale_bwd_prep', 'moe_dispatch_kernel': 'moe_dispatch',
         'moe_gather_add_kernel': 'moe_gather_add', 'splitk_reduce_kernel': 'splitk_reduce',
         'partials_reduce_kernel': 'row_partials_reduce', 'colsum_kernel': 'colsum_f32'}
out = {'correction': 'gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled; WRITE_SIZE as is; KB',
       'kernels': {}}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, {}).get('FETCH_SIZE')
    w = write.get(k, {}).get('WRITE_SIZE')
    if not f or not w:
        continue
    hbm = (2.0 * f['mean'] + w['mean']) * 1024.0
    out['kernels'][k] = dict(launches=f['launches'], fetch_kb_mean=round(f['mean'], 1), write_kb_mean=round(w['mean'], 1),
                             hbm_bytes_per_launch=round(hbm))
r = bench.get('roofline') or {}
g = out['kernels'].get('gemm_f32_kernel')
sr = out['kernels'].get('splitk_reduce_kernel')
if g and r.get('algorithmic_bytes_per_launch'):
    # the bench brackets a TN GEMM together with its slice-reduce pass: count that pass's traffic with the family
    extra = sr['hbm_bytes_per_launch'] * sr['launches'] / g['launches'] if sr else 0.0
    out['gemm_family'] = dict(hbm_bytes_per_launch=round(g['hbm_bytes_per_launch'] + extra),
                              algorithmic_bytes_per_launch=r['algorithmic_bytes_per_launch'],
                              ratio=round((g['hbm_bytes_per_launch'] + extra) / r['algorithmic_bytes_per_launch'], 3))
if mfma and 'gemm_f32_kernel' in mfma:
    m = mfma['gemm_f32_kernel']
    busy, wave = m.get('SQ_VALU_MFMA_BUSY_CYCLES'), m.get('SQ_BUSY_CYCLES')
    ga = m.get('GRBM_GUI_ACTIVE')
    if busy and ga:
        # same formula as profiles/r01/pmc_mfma.json: MFMA-busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE
        # over the 8 XCDs (calibration: 64.0 busy cycles per v_mfma_f32_32x32x2_f32 on an isolated GEMM)
        out['gemm_family_mfma_util'] = round(busy['total'] / ((ga['total'] / 8.0) * 1024.0), 4)
        wc = m.get('SQ_WAVE_CYCLES')
        if wc:
            out['gemm_family_wait_inst_any_over_wave_cycles'] = round(m['SQ_WAIT_INST_ANY']['total'] / wc['total'], 3)
            out['gemm_family_wait_any_over_wave_cycles'] = round(m['SQ_WAIT_ANY']['total'] / wc['total'], 3)
# per-kernel: counted HBM bytes vs the algorithmic bytes the bench line states for the same C-ABI call
alg, nl = bench.get('kernels_algorithmic_mb_per_step', {}), bench.get('kernels_launches_per_step', {})
out['vs_algorithmic'] = {}
for rk, bk in ALIAS.items():
    k = out['kernels'].get(rk)
    if k and bk in alg and nl.get(bk):
        a = alg[bk] * 1e6 / nl[bk]
        out['vs_algorithmic'][bk] = dict(hbm_bytes_per_launch=k['hbm_bytes_per_launch'],
                                         algorithmic_bytes_per_launch=round(a), ratio=round(k['hbm_bytes_per_launch'] / a, 3))
print(json.dumps(out, indent=1))
