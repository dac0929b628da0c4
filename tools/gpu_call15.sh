#!/bin/bash
mkdir -p gpurun_out/c15 && O=$PWD/gpurun_out/c15
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "iou_counts or config5 or g4_big or frame_pointer" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for args in "--no-pipeline" "--pipeline --parts 2"; do
timeout 200 python bench.py --config 5 --steps 20 --warmup 3 --no-extras $args | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg5 $args', d['value'], d['ms_per_step'], 'cost', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'mix', d['roofline_mix']['avg_launch_ms'], d['roofline_layer']['b_cost_basis']['frac'])"
done
CFG=5 timeout 200 python tools/stage_timing.py 512 2>&1 | tail -8
