#!/bin/bash
mkdir -p gpurun_out/c12 && O=$PWD/gpurun_out/c12
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
cd /tmp
for b in 1 4; do
  rm -rf /tmp/lt$b
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt$b -- python $R/tools/latency_trace.py $b > $O/lt$b.log 2>&1
  f=$(find /tmp/lt$b -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_overlap.py $f 5 > $O/lat$b.txt 2>&1
  echo "== B=$b"; cat $O/lat$b.txt
done
rm -rf /tmp/tr2
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr2 -- python $R/bench.py --steps 4 --warmup 2 --no-extras > $O/trace2.log 2>&1
f=$(find /tmp/tr2 -name '*kernel_trace.csv' | head -1); python $R/tools/trace_overlap.py $f 8
cd $R
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-traffic > $O/bench2.json 2>$O/bench2.err
python -c "
import json
d=json.load(open('$O/bench2.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_mix']['frac'], d['latency'], {k:v['ms'] for k,v in d['batch_sweep'].items()})"
timeout 200 python tools/frame_loop_timing.py > $O/frame_loop.log 2>&1; tail -3 $O/frame_loop.log
