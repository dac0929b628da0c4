#!/bin/bash
# What the driver does at round end, in one gpurun call: GPU tests, smoke, default bench.
mkdir -p gpurun_out/final && O=gpurun_out/final
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=3 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_source','')[:40], d['latency'], d['cpu_baseline']['value'])"
