#!/bin/bash
# round-2 GPU call 1: full GPU test-suite + the three bench configs + encoder evidence
mkdir -p gpurun_out/c1 && O=gpurun_out/c1
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python bench.py --steps 100 --warmup 10 > $O/bench2.json 2> $O/bench2.err
timeout 300 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench5.json 2> $O/bench5.err
timeout 300 python bench.py --config 3 --steps 50 --warmup 5 > $O/bench3.json 2> $O/bench3.err
timeout 300 python tools/encoder_profile.py table $O/encoder_table.md > $O/encoder_table.log 2>&1
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc3 -- python $R/tools/encoder_profile.py replay3 > $R/$O/prof3.log 2>&1)
find /tmp/prof_enc3 -name '*_kernel_stats.csv' -exec cp {} $O/encoder_cfg3_kernel_stats.csv \;
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step4 -- python $R/tools/encoder_profile.py step4 > $R/$O/prof4.log 2>&1)
find /tmp/prof_step4 -name '*_kernel_stats.csv' -exec cp {} $O/cfg4_step_kernel_stats.csv \;
timeout 200 python tools/config5_timing.py > $O/config5_timing.log 2>&1
tail -5 $O/pytest.log; cat $O/bench2.json | head -c 3000; echo; cat $O/bench5.json | head -c 1500; echo; cat $O/bench3.json | head -c 1500; echo; tail -3 $O/config5_timing.log
