#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py dmm_net_amd/csrc/dmm_cost.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
err = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = None
for line in err.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*?):\s+(\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
        if m.group(1).startswith("LDS Size"):
            if flt in cur["name"]:
                print(f"{cur['name'][:72]:72s} VGPR {cur.get('VGPRs'):>4s} SGPR {cur.get('TotalSGPRs'):>4s} "
                      f"spill(s/v) {cur.get('SGPRs Spill')}/{cur.get('VGPRs Spill')} scratch {cur.get('ScratchSize [bytes/lane]')} "
                      f"waves/SIMD {cur.get('Occupancy [waves/SIMD]')} LDS {cur.get('LDS Size [bytes/block]')}")
            cur = None
