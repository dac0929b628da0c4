#!/bin/bash
# config 4 encoder step: the variant matrix, then rocprofv3 kernel stats of the settings that matter
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/cfg4; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 2400 python $R/tools/cfg4_probe.py matrix $O/matrix.jsonl > $O/matrix.log 2>&1
prof() {  # name, env..., -- variant args
  name=$1; shift
  d=/tmp/prof_$name; rm -rf $d

  ( export "${ENVV[@]}"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/tools/cfg4_probe.py "${ARGS[@]}" > $O/prof_$name.log 2>&1 )
  f=$(find $d -name '*_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$name.csv
}
ENVV=(X=1); ARGS=(f32_nchw --steps 6); prof f32_nchw
ENVV=(X=1); ARGS=(ac_nchw --steps 6); prof ac_nchw
ENVV=(PYTORCH_MIOPEN_SUGGEST_NHWC=1); ARGS=(ac_nhwc --steps 6); prof ac_nhwc_suggest
ENVV=(PYTORCH_MIOPEN_SUGGEST_NHWC=1); ARGS=(bf16_nhwc --steps 6); prof bf16_nhwc_suggest
cat $O/matrix.jsonl
