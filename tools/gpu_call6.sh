#!/bin/bash
mkdir -p gpurun_out/c6 && O=$PWD/gpurun_out/c6
export TMPDIR=/tmp
R=$PWD
cd /tmp
for b in 1 4; do
  rm -rf /tmp/lt$b
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt$b -- python $R/tools/latency_trace.py $b > $O/lt$b.log 2>&1
  f=$(find /tmp/lt$b -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_overlap.py $f 14 > $O/lat$b.txt 2>&1
  echo "== B=$b"; cat $O/lat$b.txt
done
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
