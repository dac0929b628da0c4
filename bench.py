#!/usr/bin/env python3
"""bench.py -- frames/sec of the DMM-Net cost+match layer on MI355X.

Metric (BASELINE.json): frames/sec (cost+match layer) at N=50 proposals, M=10 templates, 255x255,
20 outer x 5 inner solver iterations, fp32 (configs[1]).  One "step" = one pass of the whole layer
(IoU cost tables -> cosine + relaxed assignment -> assignment-weighted mask mix; forward, is_test=1)
over a batch of ``--frames`` synthetic frames already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL): every rank
owns its own B frames (weak scaling; the forward has no exchange step), timing is bracketed by
barrier + synchronize on both sides and the MAX over ranks is used.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU per step (B)")
    ap.add_argument("--no-pipeline", action="store_true", help="single-stream schedule")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                                                      "exercise the multi-rank control flow on a single-GPU box)")
    return ap.parse_args()


def cpu_baseline(seconds):
    """Oracle (plain-C port of the reference layer) timed on the host cores of this box on the same workload: one
    frame per call, one worker thread per core (ctypes releases the GIL; the C routine is re-entrant)."""
    import threading
    import oracle
    from dmm_net_amd import synth
    c = synth.CONFIGS[2]
    fr = synth.make_frame(c["P"], c["O"], c["H"], c["W"], c["D"], seed=99, kind="uniform")

    def one():
        oracle.match_forward(fr.proposed_mask, fr.mask_last_occurence, fr.proposed_feature, fr.template_feature,
                             fr.proposal_score, max_iter=20, proj_iter=5, is_test=1)
    one()                                                    # warm-up
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    counts = [0] * cores
    t0 = time.perf_counter()
    deadline = t0 + seconds

    def worker(k):
        while time.perf_counter() < deadline:
            one()
            counts[k] += 1
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    n = sum(counts)
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} frames of the same workload (N=50, M=10, 255x255, 20x5 iters) in {dt:.1f} s, "
                      f"oracle/dmm_oracle.c, {cores} threads (one frame per call per thread)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % max(torch.cuda.device_count(), 1)       # identity on a full node; lets 2 test ranks share 1 GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)

    from dmm_net_amd import _lib, ops, synth
    _lib.load()                                              # loud failure if the HIP extension is missing
    c = synth.CONFIGS[2]
    B, N, M, H, W, D = args.frames, c["P"], c["O"], c["H"], c["W"], c["D"]
    HW = H * W
    g = torch.Generator(device=dev).manual_seed(synth.BASE_SEED + 2 + 1000 * rank)
    pm = torch.rand((B, N, H, W), generator=g, device=dev)
    tm = torch.rand((B, M, H, W), generator=g, device=dev)
    pf = torch.randn((B, N, D), generator=g, device=dev)
    tf = torch.randn((B, M, D), generator=g, device=dev)
    sc = torch.rand((B, N), generator=g, device=dev)

    # pre-allocated plan: nothing is allocated in the timed region.  pipeline=True = streaming lane (cost, mix)
    # on the current stream + latency lane (normalise, cosine, solver) on a side stream (ops.ForwardPlan).
    plan = ops.ForwardPlan(B, N, M, H, W, D, dev, pipeline=not args.no_pipeline)
    halves = plan.halves if plan.pipeline else [(0, B)]
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in halves]
          for _ in range(args.steps)]
    if not plan.pipeline:
        # unpipelined: time the cost kernel through the granular C-ABI calls on the current stream
        L = _lib.load()
        Pp = ops.padded_width(N, M)
        i32, f32 = dict(dtype=torch.int32, device=dev), dict(dtype=torch.float32, device=dev)
        counts = torch.empty((B * M * N + B * N + B * M,), **i32)
        inter, ap, at = counts[:B * M * N], counts[B * M * N:B * M * N + B * N], counts[B * M * N + B * N:]
        pn, tn, cosv = torch.empty_like(pf), torch.empty_like(tf), torch.empty((B, M, N), **f32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        P = lambda t: t.data_ptr()

    def step(k=None):
        if plan.pipeline:
            plan.cost_events = ev[k] if k is not None else None
            plan.run(pm, tm, pf, tf, sc, score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
            return
        if k is not None:
            ev[k][0][0].record()
        rc = L.dmm_iou_counts(P(pm), P(tm), 0, B, N, M, HW, N * HW, HW, M * HW, HW, None, None, P(inter), P(ap), P(at),
                              stream)
        if k is not None:
            ev[k][0][1].record()
        rc |= L.dmm_feature_normalize_f32(P(pf), B * N, D, P(pn), None, stream)
        rc |= L.dmm_feature_normalize_f32(P(tf), B * M, D, P(tn), None, stream)
        rc |= L.dmm_cosine_f32(P(tn), P(pn), B, N, M, D, None, None, P(cosv), stream)
        rc |= L.dmm_relax_match_f32(P(cosv), P(inter), P(ap), P(at), P(sc), B, N, M, None, None, 0.3, 20, 5, 0.1, 1,
                                    P(plan.sim), None, P(plan.Rb), P(plan.match_score), P(plan.det_score),
                                    P(plan.iters), None, stream)
        rc |= L.dmm_mask_mix(P(plan.Rb), P(pm), 0, B, N, M, Pp, HW, N * HW, HW, None, None, P(plan.full_outmask),
                             M * HW, HW, stream)
        if rc:
            raise RuntimeError(f"libdmm_match call failed: {rc}")

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # every frame ran the solver; a frame may take the reference's data-dependent early exit (relax_match.py:96-98)
    it_mean = float(plan.iters.float().mean())
    assert int(plan.iters.max()) == 20 and it_mean > 19.5, f"work was skipped inside the timed region ({it_mean})"
    assert bool(torch.isfinite(plan.full_outmask[-1]).all()) and float(plan.full_outmask[-1].abs().sum()) > 0
    # dominant kernel = dmm::iou_counts_kernel: HIP events around each of its launches, on its own stream
    # (the pipelined plan launches it twice per step, once per half of the batch)
    cost_ms = float(np.mean([a.elapsed_time(b) for per_step in ev for (a, b) in per_step]))
    frames_per_launch = B / len(halves)
    alg_bytes = int(frames_per_launch * ((N + M) * HW * 4 + M * N * 4))   # B_cost x frames per launch
    achieved = alg_bytes / (cost_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch_at_frames", {}).get(str(int(frames_per_launch)))
        except Exception:
            traffic = None
    out = {
        "metric": "frames/sec (cost+match layer) at N=50 proposals, M=10 templates, 255x255",
        "value": round(world * B * args.steps / elapsed, 1), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 50 proposals x 10 templates, 255x255 fp32 masks, D=512, "
                               "20 outer x 5 inner relax iterations, forward is_test=1, uniform-random masks",
                   "frames_per_gpu_per_step": B, "mean_outer_iterations": round(it_mean, 3), "sharding": f"frames x{world} (no collective in the forward)",
                   "schedule": "streaming lane (cost, mix) + latency lane (normalise, cosine, solver) on 2 HIP streams"
                               if plan.pipeline else "single stream"},
        "roofline": {"bound": "hbm", "kernel": "dmm::iou_counts_kernel<float,16,1>", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                     "frames_per_launch": int(frames_per_launch), "avg_launch_ms": round(cost_ms, 4)},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
