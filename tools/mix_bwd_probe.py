#!/usr/bin/env python3
"""dmm_mask_mix_bwd at the training bench's shape (50 proposals x 10 rows, 255 x 255, train-mode supports): HIP-event time
per launch of the union kernel (and the row kernel, MIX_SHARED = 0) and -- with --pmc -- SQ / traffic counters from separate
rocprofv3 passes of this script (--kernel-trace --pmc only).  (Round 5 also ran an LDS-staged fp32 MFMA form through this
probe: profiles/r05_mix_bwd_mfma_nogo.md.)

    python tools/mix_bwd_probe.py [--frames 512] [--pmc] [--steps 1,2]"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=512)
ap.add_argument("--pmc", action="store_true")
ap.add_argument("--child", action="store_true")
ap.add_argument("--steps", default="1", help="MIX_SHARED_STEPS values to try (the backward takes twice as many 4 KiB steps per workgroup)")
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--rows", type=int, default=10, help="templates per frame (10 = the training bench's shape, 5 = the product's)")
ap.add_argument("--lockstep", type=int, default=-1, help="--pmc: pin MIX_SHARED_LOCKSTEP for the counter passes")
ap.add_argument("--xcd", type=int, default=-1, help="pin MIX_XCD (1 = plain frame / step mapping of the union kernels, 3 = one "
                                                    "whole frame per XCD); default: both, one after the other")
args = ap.parse_args()

PASSES = [["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAVES",
           "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"],
          ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
           "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
          ["FETCH_SIZE"], ["WRITE_SIZE"]]


def run():
    import torch
    from dmm_net_amd import _lib, ops
    if args.lockstep >= 0 and args.child:                            # counter passes of one setting: no A/B inside
        _lib.set_option("MIX_SHARED_LOCKSTEP", args.lockstep)
    if args.xcd >= 0 and args.child:
        _lib.set_option("MIX_XCD", args.xcd)
    dev = torch.device("cuda", 0)
    B, N, M, H, W = args.frames, 50, args.rows, 255, 255
    g = torch.Generator(device=dev).manual_seed(7)
    pm = torch.rand((B, N, H, W), generator=g, device=dev)
    dout = torch.rand((B, M, H, W), generator=g, device=dev)
    # train-mode support: ~13 of 50 proposals per row, 48 distinct planes per frame (bench_train's statistics)
    Rb = torch.rand((B, M, N), generator=g, device=dev)
    Rb = torch.where(torch.rand((B, M, N), generator=g, device=dev) < 0.27, Rb, torch.zeros_like(Rb))
    union = int((Rb != 0).any(1).sum())
    alg = (union + B * M) * H * W * 4
    out = {"frames": B, "union_planes_per_frame": round(union / B, 2), "algorithmic_bytes": alg, "runs": {}}

    def ms(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    ms(lambda: ops.mask_mix_bwd(Rb, pm, dout), 10)                       # clocks up before the first figure
    for visit in (16,):                                              # (32-byte visits: built, measured, removed -- r06_mix_bwd_visit32_nogo.md)
      for steps in [int(v) for v in args.steps.split(",")]:
        for lock in ((args.lockstep,) if args.lockstep >= 0 else (0, 1)):
            for xcd in ((args.xcd,) if args.xcd >= 0 else (1, 3, 1, 3)):
                with _lib.options(MIX_SHARED_STEPS=steps, MIX_SHARED_LOCKSTEP=lock, MIX_XCD=xcd):
                    t = ms(lambda: ops.mask_mix_bwd(Rb, pm, dout), args.reps)
                    tf = ms(lambda: ops.mask_mix(Rb, pm, shared=True), args.reps)
                for name, tt in (("bwd_union", t), ("fwd_union", tf)):
                    out["runs"].setdefault(f"{name}_visit{visit}_steps{steps}_lockstep{lock}_xcd{xcd}", []).append(
                        {"ms": round(tt, 4), "frac_of_8TBps": round(alg / tt / 1e6 / 8000, 4)})
    with _lib.options(MIX_SHARED=0):
        t = ms(lambda: ops.mask_mix_bwd(Rb, pm, dout), 3)
    out["runs"]["bwd_rows"] = {"ms": round(t, 4), "frac_of_8TBps": round(alg / t / 1e6 / 8000, 4)}
    print(json.dumps(out))


if args.child or not args.pmc:
    run()
    if not args.pmc:
        sys.exit(0)
if args.pmc and not args.child:
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    for i, ctrs in enumerate(PASSES):
        d = f"/tmp/mixbwd_pmc{i}"
        shutil.rmtree(d, ignore_errors=True)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--",
                            sys.executable, os.path.abspath(__file__), "--child", "--frames", str(args.frames), "--reps", "3",
                            "--steps", args.steps.split(",")[0], "--lockstep", str(args.lockstep), "--xcd", str(args.xcd)],
                           cwd="/tmp", env=env,
                           capture_output=True, text=True)
        if r.returncode != 0:
            print("pass failed:", ctrs, r.stderr[-400:])
            continue
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0]
                if "mask_mix_bwd" not in name and "mask_mix_shared" not in name:
                    continue
                name = name.replace("void ", "")
                res.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if "mask_mix_bwd" in name or "mask_mix_shared" in name:
                    res.setdefault(name, {}).setdefault("duration_us_profiled", []).append(
                        (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
    for name, cs in res.items():
        print(name)
        print("   ", {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())})
