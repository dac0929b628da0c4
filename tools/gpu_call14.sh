#!/bin/bash
mkdir -p gpurun_out/c14 && O=$PWD/gpurun_out/c14
export TMPDIR=/tmp
rm -f gpurun_out/parity_achieved.jsonl
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
cat gpurun_out/parity_achieved.jsonl | head -40
timeout 300 python tools/config4_timing.py > $O/config4.log 2>&1; tail -2 $O/config4.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
