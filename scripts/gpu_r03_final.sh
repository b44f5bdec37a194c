#!/bin/bash
# full -m gpu suite + smoke + the round's artifact collection
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/c19_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/c19_pytest.log)" | tee $O/c19_summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/c19_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/c19_smoke.log)" | tee -a $O/c19_summary.txt
bash $R/scripts/collect_artifacts.sh c19
