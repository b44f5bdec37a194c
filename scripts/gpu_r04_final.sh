#!/bin/bash
# round 4: full -m gpu suite + smoke + the round's artifact collection (bench lines of every config, rocprofv3 stats, PMC passes)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
TAG=${1:-r04f}
timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/${TAG}_pytest.log)" | tee $O/${TAG}_summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/${TAG}_smoke.log)" | tee -a $O/${TAG}_summary.txt
bash $R/scripts/collect_artifacts.sh $TAG
