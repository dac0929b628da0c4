#!/bin/bash
mkdir -p gpurun_out/c3 && O=gpurun_out/c3
export TMPDIR=/tmp
timeout 300 python tools/solver_timing.py > $O/solver_timing.log 2>&1
timeout 300 python tools/config5_timing.py > $O/config5_timing.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cat $O/solver_timing.log $O/config5_timing.log; tail -30 $O/pytest.log
