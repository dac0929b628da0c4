#!/bin/bash
mkdir -p gpurun_out/c11 && O=$PWD/gpurun_out/c11
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
cd /tmp
for b in 1 4; do
  rm -rf /tmp/lt$b
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt$b -- python $R/tools/latency_trace.py $b > $O/lt$b.log 2>&1
  f=$(find /tmp/lt$b -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_overlap.py $f 10 > $O/lat$b.txt 2>&1
  echo "== B=$b"; cat $O/lat$b.txt
done
cd $R
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $O/bench2.json 2>$O/bench2.err
python -c "
import json
d=json.load(open('$O/bench2.json'))
print(d['value'], d['roofline']['frac'], d['roofline_mix']['frac'], d['latency'], {k:v['ms'] for k,v in d['batch_sweep'].items()})"
timeout 200 python bench.py --config 5 --steps 20 --warmup 3 --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg5', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_layer']['b_cost_basis']['frac'])"
timeout 300 python tools/encoder_layer_table.py $O/encoder_layer_table.md > $O/layer.log 2>&1; tail -3 $O/layer.log | cut -c1-200
