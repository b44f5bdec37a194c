#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_amp_gpu.py tests/test_backbone_gpu.py -m gpu -q -x > $O/c17_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/c17_pytest.log)"
timeout 300 python scripts/gemm_sweep_amp.py --cold --default-only 2>&1 | tail -1
python bench.py --config SM3Det_convnext_t --no-cpu-baseline --no-ops > $O/c17_bench_amp.json 2>$O/c17_bench_amp.err; head -c 330 $O/c17_bench_amp.json; echo
python bench.py --config SM3Det_convnext_b --no-cpu-baseline --no-ops > $O/c17_bench_amp_b.json 2>$O/c17_bench_amp_b.err; head -c 330 $O/c17_bench_amp_b.json; echo
python bench.py --no-cpu-baseline --no-ops > $O/c17_bench.json 2>$O/c17_bench.err; head -c 330 $O/c17_bench.json; echo
