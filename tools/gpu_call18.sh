#!/bin/bash
mkdir -p gpurun_out/c18 && O=$PWD/gpurun_out/c18
export TMPDIR=/tmp
timeout 600 python tools/collect_pmc_misc.py r02 $O 2>&1 | tail -6
BENCH_ARGS="--config 5" SUFFIX=_config5 timeout 600 python tools/collect_pmc_misc.py r02 $O 2>&1 | tail -6
