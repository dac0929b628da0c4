cd /tmp && export TMPDIR=/tmp
for d_ in FWD BWD WRW; do export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_$d_=0; done
rm -rf /tmp/p4; rocprofv3 --kernel-trace --output-format csv -d /tmp/p4 -- python $GRAFT_REPO_ROOT/bench.py --config 4 --bf16 --steps 6 --warmup 2 --repeats 1 --settle 3 --no-extras > $GRAFT_REPO_ROOT/gpurun_out/p4.log 2>&1
f=$(find /tmp/p4 -name "*_kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-6000:]
with gzip.open("/root/repo/gpurun_out/p4_trace_tail.csv.gz", "wt") as f:
    w = csv.writer(f)
    w.writerow(["name", "start", "end", "queue", "stream", "gx", "wx"])
    for r in rows:
        w.writerow([r["Kernel_Name"][:160], r["Start_Timestamp"], r["End_Timestamp"], r.get("Queue_Id", ""), r.get("Stream_Id", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))])
PY
