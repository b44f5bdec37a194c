"""sum rocprofv3 --pmc counters per kernel: python scripts/pmc_sum.py <dir> [name substring]"""
import csv, glob, sys, collections
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'gemm_f32_kernel')
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if sub not in k: continue
        acc[k[:70]][r['Counter_Name']] += float(r['Counter_Value']); 
        n[(k[:70], r['Counter_Name'])] += 1
for k, c in acc.items():
    print(k)
    for cn, v in sorted(c.items()):
        print(f'   {cn:32s} {v / n[(k, cn)]:16.1f} per launch ({n[(k, cn)]} launches)')
