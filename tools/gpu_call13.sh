#!/bin/bash
mkdir -p gpurun_out/c13 && O=$PWD/gpurun_out/c13
export TMPDIR=/tmp
timeout 1500 python tools/collect_profiles.py r02 $O/profiles > $O/collect.log 2>&1; tail -5 $O/collect.log | cut -c1-600
timeout 300 python tools/frame_loop_timing.py > $O/frame_loop.log 2>&1; tail -2 $O/frame_loop.log
FIND=1 timeout 400 python tools/frame_loop_timing.py > $O/frame_loop_find.log 2>&1; tail -2 $O/frame_loop_find.log
cp dmm_net_amd/miopen_db/*.txt $O/
