#!/usr/bin/env python3
"""Instruction-mix / occupancy counters of the bench's kernels (separate rocprofv3 --pmc passes, --kernel-trace only):

    python tools/collect_pmc_misc.py r01 profiles_dir

Writes <tag>_pmc_instruction_mix.json: per dmm:: kernel the per-launch averages of each counter plus derived figures
(VALU issue share = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch cycles) with the launch duration from the kernel trace)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag, out = sys.argv[1], os.path.abspath(sys.argv[2])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
bench = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras"] + \
    os.environ.get("BENCH_ARGS", "").split()
suffix = os.environ.get("SUFFIX", "")
PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
          ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_WAVES"],
          ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"]]
kernels = {}
for i, ctrs in enumerate(PASSES):
    d = f"/tmp/prof_misc{i}"
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--"] + bench,
                   cwd="/tmp", env=env, capture_output=True, text=True)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            name = name[5:] if name.startswith("void ") else name
            if not name.startswith("dmm::"):
                continue
            kernels.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
res = {"round": tag, "command": "rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 3 --warmup 1 "
                                "--no-extras " + os.environ.get("BENCH_ARGS", "") + " (three separate passes)", "passes": PASSES, "kernels": {}}
for name, cs in kernels.items():
    res["kernels"][name] = {c: {"avg_per_launch": sum(v) / len(v), "samples": len(v)} for c, v in cs.items()}
json.dump(res, open(os.path.join(out, f"{tag}_pmc_instruction_mix{suffix}.json"), "w"), indent=1)
for name, cs in res["kernels"].items():
    print(name, {c: round(v["avg_per_launch"]) for c, v in cs.items()})
