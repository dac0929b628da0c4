#!/bin/bash
mkdir -p gpurun_out/c10 && O=$PWD/gpurun_out/c10
export TMPDIR=/tmp
timeout 600 python tools/encoder_layer_table.py $O/encoder_layer_table.md > $O/layer.log 2>&1; tail -22 $O/layer.log | cut -c1-160
timeout 300 python bench.py --config 3 --steps 100 --warmup 10 > $O/bench3.json 2>$O/bench3.err; python -c "
import json; d=json.load(open('$O/bench3.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms'], d['roofline']['achieved'])"
CHANNELS_LAST=1 timeout 300 python tools/config4_timing.py > $O/config4_cl.log 2>&1; tail -2 $O/config4_cl.log
ls -la dmm_net_amd/miopen_db/; cp dmm_net_amd/miopen_db/*.txt $O/
