#!/bin/bash
mkdir -p gpurun_out/c2 && O=gpurun_out/c2
export TMPDIR=/tmp
timeout 300 python tools/solver_timing.py > $O/solver_timing.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cat $O/solver_timing.log; tail -40 $O/pytest.log
