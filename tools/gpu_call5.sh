#!/bin/bash
mkdir -p gpurun_out/c5 && O=gpurun_out/c5
export TMPDIR=/tmp
for args in "--no-pipeline" "--pipeline --parts 2" "--pipeline --parts 4" "--pipeline --parts 8" "--no-pipeline --f32-out" "--no-pipeline --frames 256"; do
  echo "== config 5 $args" >> $O/cfg5.log
  timeout 200 python bench.py --config 5 --steps 20 --warmup 3 --no-extras $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], 'cost', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'mix', d.get('roofline_mix',{}).get('avg_launch_ms'), d.get('roofline_mix',{}).get('achieved'), 'layer', d['roofline_layer']['b_cost_basis']['frac'], d['roofline_layer']['b_layer_basis']['frac'])" >> $O/cfg5.log 2>&1
done
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $O/bench2.json 2>$O/bench2.err
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cat $O/cfg5.log; python -c "
import json
d=json.load(open('$O/bench2.json'))
print(d['value'], d['roofline']['frac'], d['roofline_mix']['frac'], d['latency'], {k:v['ms'] for k,v in d['batch_sweep'].items()})"
tail -12 $O/pytest.log
