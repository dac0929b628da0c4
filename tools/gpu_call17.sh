#!/bin/bash
for w in 1024 2048 4096 8192; do
DMM_COST_TL_WGS=$w timeout 200 python bench.py --config 5 --steps 20 --warmup 3 --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('TL_WGS $w', d['value'], 'cost', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
