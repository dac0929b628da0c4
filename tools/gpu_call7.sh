#!/bin/bash
mkdir -p gpurun_out/c7 && O=$PWD/gpurun_out/c7
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_gpu_features.py -q -x -k "bias_act or fast_channels" > $O/pytest_enc.log 2>&1; tail -15 $O/pytest_enc.log
timeout 300 python tools/encoder_profile_fast.py eager > $O/encoder_fast_table.md 2>&1; head -45 $O/encoder_fast_table.md | cut -c1-170
timeout 300 python bench.py --config 3 --steps 50 --warmup 5 > $O/bench3.json 2>$O/bench3.err; cat $O/bench3.json | head -c 1200; echo
timeout 300 python bench.py --config 3 --steps 50 --warmup 5 --nchw-encoder > $O/bench3_nchw.json 2>>$O/bench3.err; cat $O/bench3_nchw.json | head -c 700; echo
(cd /tmp && rm -rf /tmp/pf && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python $R/tools/encoder_profile_fast.py > $O/prof_fast.log 2>&1)
find /tmp/pf -name '*_kernel_stats.csv' -exec cp {} $O/encoder_cfg3_kernel_stats_fast.csv \;
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
