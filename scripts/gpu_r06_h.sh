#!/bin/bash
# round 6, call H: scratch-free rotated-IoU kernels -- parity tests, two-process test, operator timings
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06h; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py tests/test_assign_gpu.py tests/test_oracle_ops.py tests/test_detector_gpu.py -q -m gpu > $O/tests_ops.txt 2>&1
tail -n 4 $O/tests_ops.txt
python scripts/ops_bench.py > $O/ops_bench.json 2> $O/ops_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06h/ops_bench.json').read())
for k in ('box_iou_rotated_2000x64','box_iou_rotated_2000x512','nms_rotated_2000','nms_rotated_10000','nms_8768','max_iou_assign_rpn_261888x8','max_iou_assign_rcnn_2000x8_rotated','orpn_proposals_one_image'):
    print(k, d.get(k) if k in d else d.get('ops_us',{}).get(k))
PY
