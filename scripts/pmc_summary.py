"""Summarise a rocprofv3 --pmc counter_collection.csv: per-kernel mean counter value per launch.

    python scripts/pmc_summary.py <dir with *counter_collection.csv>   -> <dir>/summary.json + a top-20 table

Kernel names are reduced to the bare function name: `void`, the `(anonymous namespace)::` qualifier and the template /
argument lists are stripped BEFORE the name is cut at the first parenthesis (round 1 cut first, which collapsed every
anonymous-namespace kernel into one empty key).  All `gemm_f32_kernel<...>` / `gemm_bf16_kernel<...>` instantiations
are pooled per family."""
import csv
import glob
import json
import re
import sys


def kernel_key(name):
    n = name.replace('void ', '').replace('(anonymous namespace)::', '').replace('sm3gemm::', '')
    n = re.sub(r'<.*', '', n.split('(')[0]).strip()
    return n[:60] or name[:60]


GEMM_MODE = {'0': 'nt', '1': 'nn', '2': 'tn'}


def summarise(d):
    f = glob.glob(d + '/*counter_collection.csv')[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        keys = [kernel_key(r['Kernel_Name'])]
        m = re.search(r'gemm_f32_kernel<(\d)', r['Kernel_Name'])
        if m:  # the pooled family AND the family split by operand form (first template argument)
            keys.append('gemm_f32_kernel/' + GEMM_MODE.get(m.group(1), m.group(1)))
        for key in keys:
            a = agg.setdefault((key, r['Counter_Name']), [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    out = {}
    for (k, c), (n, s) in agg.items():
        out.setdefault(k, {})[c] = dict(launches=n, mean=s / n, total=s)
    return out


if __name__ == '__main__':
    d = sys.argv[1]
    out = summarise(d)
    json.dump(out, open(d + '/summary.json', 'w'), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -list(kv[1].values())[0]['total'])[:20]:
        print(k, {c: (x['launches'], round(x['mean'], 1)) for c, x in v.items()})
