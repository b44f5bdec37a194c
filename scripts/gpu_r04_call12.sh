#!/bin/bash
# round 4, call 12: where the step's idle time sits -- kernel trace (start/end stamps) of the replayed step
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r04c12; S=$O/${T}_summary.txt; : > $S
export TMPDIR=/tmp
rm -rf /tmp/prof_tr; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -o tr -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-ops > $O/${T}_bench.json 2> $O/${T}_bench.err
f=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
echo "trace $f $(wc -l < $f)" | tee -a $S
python scripts/kernel_gaps.py $f | tee -a $S
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
PY
