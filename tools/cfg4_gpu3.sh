#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/cfg4c; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_train_encoder.py "tests/test_gpu_distributed.py::test_config4_per_gpu_share_trains_through_dmm_model" -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
cd /tmp
timeout 900 python $R/bench.py --config 4 --steps 8 --warmup 2 > $O/bench_cfg4_f32.json 2> $O/bench_cfg4_f32.err; tail -c 1500 $O/bench_cfg4_f32.json | head -c 1500; echo
timeout 900 python $R/bench.py --config 4 --steps 8 --warmup 2 --bf16 > $O/bench_cfg4_bf16.json 2> $O/bench_cfg4_bf16.err; python - <<PY
import json
for f in ("f32","bf16"):
    try:
        d=json.load(open("$O/bench_cfg4_%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["config"]["stage_ms"], d["config"]["repeats"]["ms_per_step_min_median_max"])
    except Exception as e: print(f, "ERR", e); print(open("$O/bench_cfg4_%s.err"%f).read()[-1500:])
PY
