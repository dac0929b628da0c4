#!/bin/bash
mkdir -p gpurun_out/c9 && O=$PWD/gpurun_out/c9
export TMPDIR=/tmp
R=$PWD
export MIOPEN_USER_DB_PATH=$O/miopen_db; mkdir -p $MIOPEN_USER_DB_PATH
BENCH=1 timeout 600 python tools/encoder_profile_fast.py eager > $O/encoder_fast_table_find.md 2>&1; tail -3 $O/encoder_fast_table_find.md; head -14 $O/encoder_fast_table_find.md | cut -c1-150
ls -la $MIOPEN_USER_DB_PATH | head
# config 4 per-GPU training step: first run fills MIOpen's find-db (its search runs naive reference kernels), second is traced
STEPS=3 timeout 600 python tools/encoder_profile.py step4 > $O/step4_warm.log 2>&1
(cd /tmp && rm -rf /tmp/p4 && STEPS=6 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -- python $R/tools/encoder_profile.py step4 > $O/prof4.log 2>&1)
find /tmp/p4 -name '*_kernel_stats.csv' -exec cp {} $O/cfg4_step_kernel_stats.csv \;
head -25 $O/cfg4_step_kernel_stats.csv | cut -c1-150
timeout 300 python tools/config4_timing.py > $O/config4_timing.log 2>&1; cat $O/config4_timing.log | tail -3
