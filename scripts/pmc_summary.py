"""Summarise a rocprofv3 --pmc counter_collection.csv: per-kernel-name mean counter value per launch."""
import csv, glob, json, sys
d = sys.argv[1]
f = glob.glob(d + '/*counter_collection.csv')[0]
agg = {}
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    key = 'gemm_f32_kernel' if 'gemm_f32_kernel' in n else n.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[:60]
    a = agg.setdefault((key, r['Counter_Name']), [0, 0.0])
    a[0] += 1; a[1] += float(r['Counter_Value'])
out = {}
for (k, c), (n, s) in agg.items():
    out.setdefault(k, {})[c] = dict(launches=n, mean=s / n, total=s)
json.dump(out, open(d + '/summary.json', 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -list(kv[1].values())[0]['total'])[:12]:
    print(k, {c: (x['launches'], round(x['mean'], 1)) for c, x in v.items()})
