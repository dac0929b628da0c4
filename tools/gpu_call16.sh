#!/bin/bash
mkdir -p gpurun_out/c16 && O=$PWD/gpurun_out/c16
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=3 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 200 python bench.py --config 5 --steps 20 --warmup 3 --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg5', d['value'], d['ms_per_step'], 'cost', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'mix', d['roofline_mix']['avg_launch_ms'], d['roofline_layer']['b_cost_basis']['frac'])"
timeout 200 python bench.py --config 5 --steps 20 --warmup 3 --no-extras --pipeline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg5 2-lane', d['value'], d['ms_per_step'], 'cost', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline_layer']['b_cost_basis']['frac'])"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-traffic | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_mix']['frac'], d['latency'])"
CFG=5 timeout 200 python tools/stage_timing.py 512 2>&1 | tail -2
