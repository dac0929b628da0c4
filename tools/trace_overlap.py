#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of a 2-lane run: per kernel name the mean duration, and for the last
step a timeline (start / end in us relative to the first kernel of that step) so that overlap between the streaming
lane and the latency lane can be read off.   python tools/trace_overlap.py kernel_trace.csv [n_last] [all]
("all": every kernel of the trace, not only the library's -- the tensor-op kernels between the library's launches)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60],
       r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows
      if (len(sys.argv) > 3 and sys.argv[3] == "all") or r["Kernel_Name"].startswith(("void dmm::", "dmm::"))]
ks.sort()
last = ks[-nlast:]
t0 = last[0][0]
for s, e, n, q, st in last:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:8.1f})  q={q} s={st}  {n}")
