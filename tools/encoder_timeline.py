#!/usr/bin/env python3
"""Timeline of the last replay in a rocprofv3 --kernel-trace CSV of tools/encoder_profile_fast.py: every kernel's start /
end (us from the replay's first kernel), queue and stream, so that the overlap between the body chain and the heads'
side stream can be read off.  The replay starts at the last copy of the input into the graph's static buffer.
(Under the tracer every kernel runs a few us longer than untraced; the ORDER and the overlap are what to read.)
    python tools/encoder_timeline.py kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Queue_Id", "?"),
             r.get("Stream_Id", "?")) for r in rows)
starts = [i for i, k in enumerate(ks) if "copyBuffer" in k[2]]
last = ks[starts[-1]:] if starts else ks[-116:]
t0 = last[0][0]
busy = 0
for s, e, name, q, st in last:
    print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})  q={q} s={st}  {name}")
    busy += e - s
print(f"span {(max(e for _, e, *_ in last) - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
