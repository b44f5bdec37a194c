#!/bin/bash
# round 6, call B: checkerboard one-set accumulators -- accuracy (probe, gemm tests, full-size parity) and step time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O
cd $R
U=$R/sm3det_amd/csrc/libsm3det_hip_b3_unsigned.so
T=$R/sm3det_amd/csrc/libsm3det_hip_b3_two_sets.so
python scripts/probes/b3_bias.py > $O/bias_checker.txt 2>&1
SM3DET_HIP_LIB=$U python scripts/probes/b3_bias.py > $O/bias_one_set.txt 2>&1
SM3DET_HIP_LIB=$T python scripts/probes/b3_bias.py > $O/bias_two_sets.txt 2>&1
python -m pytest tests/test_gemm_gpu.py tests/test_backbone_gpu.py -q -m gpu -x > $O/tests_gemm.txt 2>&1
python -m pytest tests/test_fullsize_gpu.py -q -m gpu > $O/tests_fullsize.txt 2>&1
mkdir -p $O/fullsize; cp gpurun_out/fullsize_*.json $O/fullsize/ 2>/dev/null
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops"
for i in 1 2; do
$B > $O/checker_$i.json 2> $O/checker_$i.err
SM3DET_HIP_LIB=$T $B > $O/two_$i.json 2> $O/two_$i.err
SM3DET_HIP_LIB=$U $B > $O/uns_$i.json 2> $O/uns_$i.err
done
tail -n 3 $O/tests_gemm.txt; tail -n 8 $O/tests_fullsize.txt
cat $O/bias_checker.txt $O/bias_one_set.txt
for f in $O/*.json; do echo "$(basename $f) $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline'].get('gemm_ms_per_step'), d['roofline'].get('other_kernels_ms_per_step'))" 2>&1 | tail -1)"; done
