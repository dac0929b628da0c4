#!/bin/bash
mkdir -p gpurun_out/c8 && O=$PWD/gpurun_out/c8
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python tools/encoder_profile_fast.py eager > $O/encoder_fast_table.md 2>&1; tail -3 $O/encoder_fast_table.md
timeout 300 python bench.py --config 3 --steps 100 --warmup 10 > $O/bench3.json 2>$O/bench3.err; python -c "
import json; d=json.load(open('$O/bench3.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms'], d['roofline']['achieved'])"
