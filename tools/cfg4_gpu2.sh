#!/bin/bash
# TrainEncoder: tests, then the probe variants (graph vs eager check), then kernel stats of the shipped form
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/cfg4b; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train_encoder.py -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
cd /tmp
for v in train train_nowgrad train_find train_nowgrad_find; do
  timeout 900 python $R/tools/cfg4_probe.py $v --steps 10 $( [ $v = train ] && echo --check ) $( [[ $v == *find ]] && echo --find ) > $O/probe_$v.log 2>&1; grep '^{' $O/probe_$v.log
done
d=/tmp/prof_train; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/tools/cfg4_probe.py train --steps 6 > $O/prof_train.log 2>&1
f=$(find $d -name '*_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_train.csv
tail -3 $O/prof_train.log
