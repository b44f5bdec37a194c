#!/bin/bash
# round 6, call L: full detector step with the GFL kernels (+ kernel stats)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06l; mkdir -p $O
cd $R
python bench.py --workload full_model --steps 10 --warmup 3 --no-cpu-baseline --no-ops > $O/full_model.json 2> $O/full_model.err
python -c "
import json; d=json.loads(open('$O/full_model.json').read().strip().splitlines()[-1]); print('full_model', d['ms_per_step'], d['value'], {k: d[k] for k in d if 'loss' in k.lower()} )"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fm -o p -- python $R/bench.py --workload full_model --steps 6 --warmup 2 --no-cpu-baseline --no-ops --no-graph > /dev/null 2>&1
f=$(find /tmp/fm -name "*kernel_stats.csv" | head -1); cp $f $O/full_model_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/full_model_kernel_stats.csv')))
steps=8
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6/steps
torch_=[r for r in rows if 'at::native' in r['Name'] or 'rocprim' in r['Name'] or 'Cijk' in r['Name']]
print('kernel ms/step', round(tot,2), 'launches/step', sum(int(r['Calls']) for r in rows)/steps)
print('torch launches/step', sum(int(r['Calls']) for r in torch_)/steps, 'ms/step', round(sum(float(r['TotalDurationNs']) for r in torch_)/1e6/steps,2))
print('Cijk rows', [r['Name'][:40] for r in rows if 'Cijk' in r['Name']])
for r in rows[:14]: print(round(float(r['TotalDurationNs'])/1e6/steps,3), int(r['Calls'])/steps, r['Name'][:90])
PY
