#!/bin/bash
mkdir -p gpurun_out/c4 && O=$PWD/gpurun_out/c4
export TMPDIR=/tmp
R=$PWD
cd /tmp
for cfg in 5 2; do
  rm -rf /tmp/tr$cfg
  FR=256; [ $cfg = 2 ] && FR=1024
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$cfg -- python $R/bench.py --config $cfg --frames $FR --steps 4 --warmup 2 --no-extras --pipeline > $O/trace$cfg.log 2>&1
  f=$(find /tmp/tr$cfg -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_overlap.py $f 14 > $O/overlap$cfg.txt 2>&1
  cat $O/overlap$cfg.txt
done
