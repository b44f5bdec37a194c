"""profiles/<round>/ops_pmc.json from the per-operator rocprofv3 passes of scripts/gpu_r03_ops_profile.sh
(gpurun_out/<tag>_ops/<op>_{stats.csv,sq.json,fetch.json,write.json}): per operator the dominant kernels with their
average duration, the VALU lane-instructions per pair test (SQ_INSTS_VALU x 64 lanes / pairs), the counted HBM bytes
(FETCH_SIZE x 2 + WRITE_SIZE, KB: MI355X_MICROARCH.md) and, for the NMS operators, the serial sweep's duration.

    python scripts/ops_pmc_summary.py gpurun_out/<tag>_ops > profiles/<round>/ops_pmc.json
"""
import csv
import json
import os
import sys

d = sys.argv[1]
PAIRS = {'box_iou_rotated_2000x512': 2000 * 512, 'box_iou_rotated_2000x64': 2000 * 64,
         'nms_rotated_2000': 2000 * 1999 / 2, 'nms_rotated_10000': 10000 * 9999 / 2, 'nms_8768': 8768 * 8767 / 2}
PAIR_KERNELS = ('box_iou_rotated_kernel', 'nms_rotated_mask_kernel', 'nms_mask_kernel')
OURS = ('box_iou', 'nms_', 'roi_align', 'roi_bwd', 'deform_', 'gemm_f32_kernel', 'transpose_f32', 'sort_', 'sm3_zero',
        'at::native::vectorized_elementwise_kernel')  # (the torch fill of the gradient map is part of a backward call)


def key(name):
    n = name.replace('void ', '').replace('(anonymous namespace)::', '').replace('sm3gemm::', '')
    return n.split('(')[0].split('<')[0].strip()


def load(op, suffix):
    p = os.path.join(d, f'{op}_{suffix}.json')
    return json.load(open(p)) if os.path.exists(p) else {}


out = {}
for f in sorted(os.listdir(d)):
    if not f.endswith('_stats.csv'):
        continue
    op = f[:-10]
    rows = list(csv.DictReader(open(os.path.join(d, f))))
    kern = {}
    for r in rows:
        k = key(r['Name'])
        if k.startswith(OURS):
            e = kern.setdefault(k, dict(calls=0, total_us=0.0))
            e['calls'] += int(r['Calls'])
            e['total_us'] += float(r['TotalDurationNs']) / 1e3
    reps = 6  # ops_profile.py: 1 warm-up + 5 timed invocations under the stats pass
    entry = dict(kernels={k: dict(launches_per_call=round(v['calls'] / reps, 2), us_per_call=round(v['total_us'] / reps, 1))
                          for k, v in sorted(kern.items(), key=lambda kv: -kv[1]['total_us'])})
    sq, fe, wr = load(op, 'sq'), load(op, 'fetch'), load(op, 'write')
    hbm = 0.0
    for k in kern:
        fk, wk = fe.get(k, {}).get('FETCH_SIZE'), wr.get(k, {}).get('WRITE_SIZE')
        if fk and wk:
            hbm += (2.0 * fk['mean'] + wk['mean']) * 1024.0 * kern[k]['calls'] / reps
    if hbm:
        entry['hbm_bytes'] = round(hbm)
    if op in PAIRS:
        for k in PAIR_KERNELS:
            if k in sq and 'SQ_INSTS_VALU' in sq[k]:
                entry['valu_lane_instr_per_pair'] = round(sq[k]['SQ_INSTS_VALU']['mean'] * 64.0 / PAIRS[op], 1)
                wc, act = sq[k].get('SQ_WAVE_CYCLES'), sq[k].get('SQ_ACTIVE_INST_VALU')
                busy, ga = sq[k].get('SQ_BUSY_CYCLES'), sq[k].get('GRBM_GUI_ACTIVE')
                if act and ga:  # VALU-active quad-cycles x 4 over SIMD-cycles (1024 SIMDs; GRBM summed over 8 XCDs)
                    entry['valu_busy'] = round(act['mean'] * 4.0 / ((ga['mean'] / 8.0) * 1024.0), 3)
        for k in kern:
            if k.startswith('nms_sweep'):
                entry['serial_floor_us'] = round(kern[k]['total_us'] / reps, 1)
    out[op] = entry
out['_method'] = ('rocprofv3 --kernel-trace --stats and separate --pmc passes per operator (scripts/gpu_r03_ops_profile.sh, '
                  'scripts/ops_profile.py); FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request), WRITE_SIZE as is')
print(json.dumps(out, indent=1))
