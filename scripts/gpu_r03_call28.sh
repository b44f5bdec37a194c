#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/c28_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/c28_pytest.log)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/c28_bench.json 2> $O/c28_bench.err; head -c 260 $O/c28_bench.json; echo
