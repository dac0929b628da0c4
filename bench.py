#!/usr/bin/env python3
"""bench.py -- frames/sec of the DMM-Net cost+match layer on MI355X.

Default workload = BASELINE.json configs[1] (the configuration the metric is quoted on): N=50 proposals,
M=10 templates, 255x255 fp32 masks, 20 outer x 5 inner solver iterations.  One "step" = one pass of the whole layer
(IoU cost tables -> cosine + relaxed assignment -> assignment-weighted mask mix; forward, is_test=1) over a batch of
``--frames`` synthetic frames already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B] [--config 2|3|4|5|loop]

``--config 5`` = configs[4] (N=200, M=20, fp16 mask planes, fp32 accumulation), ``--config 3`` = configs[2]
(ResNet-50 + heads in bf16 -> fused 4-level ROIAlign+mean -> the layer, 8 frames), ``--config 4`` = configs[3]'s per-GPU
share (ResNet-101 training step through DMM_Model, gradient mean over RCCL with the bucketed overlapped all-reduce;
``--gpus 8`` is the BASELINE configuration), ``--config loop`` = the evaluator's frame loop (video.FrameLoop: encoder +
two-phase proposal prep + ROI features + matching + label merge, 4 videos of 255x448, eval solver setting 40x5); same
JSON schema, ``config.workload`` names the configuration.  The default N = 1 line also carries compact results of
configs 3, 5 and the frame loop under ``other_configs`` (``--no-others`` skips them).

N > 1 runs one rank per GPU over RCCL: launched by the driver through torch.distributed.run, or -- when ``--gpus N`` is
given WITHOUT a launcher around it (WORLD_SIZE unset) -- by this script re-executing itself under torch.distributed.run;
a world size other than N is an error, never a number.  Every rank owns its own B
frames (weak scaling; the forward has no exchange step), timing is bracketed by barrier + synchronize on both sides and
the MAX over ranks is used.  Rank 0 prints ONE JSON line.

At N = 1 the line also carries (``--no-extras`` skips them): ``cpu_baseline`` (the C oracle on the host cores, bounded
sample), ``batch_sweep`` / ``latency`` (frames per launch sequence in {1, 4, 8, 64, 512, 1024}), and ``roofline.traffic``
measured by two child ``rocprofv3 --pmc`` passes of this same command (``traffic_source`` says where the number is from).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md); never the 2:1-sparsity headline
METRIC = "frames/sec (cost+match layer) at N=50 proposals, M=10 templates, 255x255"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="2", choices=("2", "3", "4", "5", "loop", "train", "dropin"),
                    help="BASELINE configs index + 1; loop = the evaluator's frame loop; train = the layer's training "
                         "form (forward + backward) with per-kernel rooflines; dropin = host-inclusive cost per call of "
                         "the drop-in classes themselves (MatchModel.forward, DMM_Model)")
    ap.add_argument("--no-others", action="store_true", help="default run: skip the compact other_configs entries")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (default 1024 / 8 / 512 for "
                                                          "config 2 / 3 / 5)")
    ap.add_argument("--no-pipeline", action="store_true", help="single-stream schedule")
    ap.add_argument("--pipeline", action="store_true", help="force the 2-lane schedule")
    ap.add_argument("--parts", type=int, default=0, help="slices of the batch in the 2-lane schedule (default 2)")
    ap.add_argument("--nchw-encoder", action="store_true", help="config 3: the round-1 NCHW / all-MIOpen encoder")
    ap.add_argument("--contiguous-planes", action="store_true", help="config 5: packed [B,K,H,W] planes (round 3's layout) "
                                                                     "instead of the 128-byte-aligned plane stride")
    ap.add_argument("--f32-out", action="store_true", help="config 5: write full_outmask in fp32 instead of fp16")
    ap.add_argument("--f16-solver", action="store_true", help="config 5: the fp16-state solver (tolerance mode: R within 1e-2, "
                    "argmax identical where the fp32 top-2 gap exceeds 0.02) instead of the default bit-exact fp32-state one on the "
                    "same fp16 planes; it buys ~1 % (round 5), so the bit-exact form is the default")
    ap.add_argument("--f32-solver", action="store_true", help="(the default since round 6; accepted for old command lines)")
    ap.add_argument("--bf16", action="store_true", help="config 4: the encoder as train_encoder.TrainEncoder -- bf16 channels-last, "
                    "fp32 master weights, HIP-graph replays, own BatchNorm / weight-gradient kernels (the shipped bf16 training form)")
    ap.add_argument("--bn-groups", type=int, default=0, help="config 4 --bf16: BatchNorm statistics groups of the encoder call "
                    "(0 = the clip length 3 when it divides the frames: statistics per frame step, as the reference's per-frame "
                    "encoder calls compute them; 1 = over all frames of the call)")
    ap.add_argument("--autocast", action="store_true", help="config 4: run the encoder under bf16 autocast (the reference "
                                                          "trains in fp32, which is the default here)")
    ap.add_argument("--repeats", type=int, default=3, help="config 4: timed repeats of K steps each; the line carries the "
                                                           "median with min / max beside it")
    ap.add_argument("--settle", type=int, default=8, help="config 4: untimed steps before the W warm-ups (MIOpen solver "
                                                          "picks, allocator working set, bucketer steady mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="no cpu_baseline / batch_sweep / in-run PMC traffic")
    ap.add_argument("--no-traffic", action="store_true", help="do not spawn the rocprofv3 --pmc child passes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                                                      "exercise the multi-rank control flow on a single-GPU box)")
    a = ap.parse_args()
    a.config = a.config if a.config in ("loop", "train", "dropin") else int(a.config)
    return a


# ---------------------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds, ci=2):
    """Oracle (plain-C port of the reference layer) timed on the host cores of this box on the same workload: one
    frame per call, one worker thread per core (ctypes releases the GIL; the C routine is re-entrant)."""
    import threading
    import oracle
    from dmm_net_amd import synth
    c = synth.CONFIGS[ci]
    fr = synth.make_frame(c["P"], c["O"], c["H"], c["W"], c["D"], seed=99, kind="uniform")
    pm, tm = fr.proposed_mask, fr.mask_last_occurence
    if ci == 5:                                      # fp16 storage: the port computes on the rounded values in fp32
        pm, tm = pm.astype(np.float16).astype(np.float32), tm.astype(np.float16).astype(np.float32)

    def one():
        oracle.match_forward(pm, tm, fr.proposed_feature, fr.template_feature, fr.proposal_score, max_iter=20,
                             proj_iter=5, is_test=1)
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
    out = {"value": round(n / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
           "sample": f"{n} frames of the same workload (N={c['P']}, M={c['O']}, 255x255, 20x5 iters) in {dt:.1f} s, "
                     f"oracle/dmm_oracle.c, {cores} threads (one frame per call per thread)"}
    # SURVEY 8(d) / north_star: "the reference PyTorch-CPU path timed on the host cores" -- the op-for-op torch restatement
    # (oracle/torch_ref.py: the reference's own tensor ops per frame, expands and .item() syncs included) with intra-op
    # parallelism over all cores, one frame at a time like the reference's per-frame call; a few seconds of frames
    try:
        from oracle import torch_ref
        old_threads = torch.get_num_threads()
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        args_t = (tt(fr.proposed_feature), tt(pm), tt(fr.template_feature), tt(tm), tt(fr.proposal_score))
        by_threads = {}
        # all host cores as north_star says -- and 16 threads: the layer is ~40 small tensor ops and 100+ .item() syncs per
        # frame, and torch's intra-op pool on 256 threads spends its time waking threads (measured 2.6 s per frame on 256
        # threads of an EPYC 9575F against ~0.1-0.2 s on 16)
        for nt in sorted({cores, min(16, cores)}, reverse=True):
            torch.set_num_threads(nt)
            torch_ref.match_forward(*args_t, max_iter=20, proj_iter=5, is_test=1)
            times = []
            t1 = time.perf_counter()
            while len(times) < 12 and time.perf_counter() - t1 < max(2.5, seconds / 4):
                t2 = time.perf_counter()
                torch_ref.match_forward(*args_t, max_iter=20, proj_iter=5, is_test=1)
                times.append(time.perf_counter() - t2)
            by_threads[nt] = (len(times), float(np.mean(times)), float(np.std(times)))
        torch.set_num_threads(old_threads)
        best = min(by_threads, key=lambda k: by_threads[k][1])
        out["torch_ops"] = {"value": round(1.0 / by_threads[best][1], 3), "unit": "frames/s", "cores": best,
                            "kind": "port (op-for-op torch restatement of match_model.py:24-148, oracle/torch_ref.py)",
                            "sample": "; ".join(f"{n} frames on {nt} torch intra-op threads: {m * 1e3:.0f} +- {sd * 1e3:.0f} ms "
                                                "per frame" for nt, (n, m, sd) in by_threads.items())
                                      + f" -- value = the faster setting ({best} threads), one frame per call"}
    except Exception as e:                                   # a side figure must not cost the line
        out["torch_ops"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def pmc_traffic(extra_args, kernel_prefixes):
    """HBM bytes per launch of the named kernels from two SEPARATE child passes ``rocprofv3 --kernel-trace --pmc
    FETCH_SIZE`` / ``--pmc WRITE_SIZE`` of this same command (3 steps), as MI355X_MICROARCH.md's HBM section
    prescribes: counter values are KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide streaming read (x2).
    Returns {prefix: bytes per launch} or None when rocprofv3 is unavailable / fails."""
    if shutil.which("rocprofv3") is None:
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    sums = {p: {"FETCH_SIZE": [], "WRITE_SIZE": []} for p in kernel_prefixes}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"dmm_pmc_{ctr}_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras"]
            r = subprocess.run(cmd + extra_args, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                return None
            for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != ctr:
                        continue
                    name = row["Kernel_Name"]
                    name = name[5:] if name.startswith("void ") else name
                    for p in kernel_prefixes:
                        if name.startswith(p):
                            sums[p][ctr].append(float(row["Counter_Value"]))
        except (subprocess.TimeoutExpired, OSError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for p, v in sums.items():
        if not v["FETCH_SIZE"] or not v["WRITE_SIZE"]:
            return None
        fa = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
        wa = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        out[p] = int(fa * 1024 * 2 + wa * 1024)
    return out


def profile_traffic(tag_file, key):
    path = os.path.join(ROOT, "profiles", tag_file)
    try:
        return json.load(open(path)).get("hbm_bytes_per_launch_at_frames", {}).get(str(key)), "profiles/" + tag_file
    except Exception:
        return None, None


def self_launch(args):
    """``python bench.py --gpus N`` (N > 1) without a launcher around it: become ``python -m torch.distributed.run
    --nproc-per-node N ... bench.py <same arguments>`` (one rank per GPU, rendezvous on 127.0.0.1), so that a plain
    invocation can never print a 1-GPU number for an N-GPU request.  Under a launcher (WORLD_SIZE set) nothing happens
    here; ``Runner`` then insists on WORLD_SIZE == N."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and ndev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} over RCCL needs {args.gpus} visible GPUs, this box has {ndev} "
                         "(RCCL refuses two ranks on one device; --backend gloo only exercises the control flow)")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)                        # does not return


def bind_to_gpu_numa_node(index):
    """Pin this process's host threads to the CPUs local to GPU ``index`` (``distributed.bind_host_threads_to_gpu``): on
    an 8-GPU node the launch path of a rank then does not cross sockets.  Returns the NUMA node or None; every failure (no
    sysfs entry, no permission, a container without the topology) is silent -- the bench runs unpinned."""
    from dmm_net_amd.distributed import bind_host_threads_to_gpu
    return bind_host_threads_to_gpu(index)[0]


class Runner:
    """dist init, fences and the timed loop shared by the three workloads."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            # never a number for another rank count than the one asked for
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        local = local % max(torch.cuda.device_count(), 1)   # identity on a full node; lets 2 test ranks share 1 GPU
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        # host threads next to this rank's GPU when several ranks share the box (silent if the topology is unknown); a
        # single rank keeps every core -- its cpu_baseline leg uses them
        self.numa = bind_to_gpu_numa_node(local) if self.world > 1 else None
        self.dist = None
        self.rccl_ranks = 1
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)   # RCCL over xGMI
            else:
                dist.init_process_group(args.backend)
            # the rank count the COLLECTIVE sees, not the one the environment claims: a sum of ones, on the device under RCCL
            one = torch.ones(1, dtype=torch.float32, device=self.dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(one)
            got = int(round(float(one.item())))
            if got != self.world:
                raise SystemExit(f"bench.py: all_reduce(1) over the process group gave {got}, WORLD_SIZE is {self.world}")
            self.rccl_ranks = got if args.backend == "nccl" else 0

    def fence(self):
        torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            self.dist.barrier()
            torch.cuda.synchronize(self.dev)

    def timed(self, step, steps, warmup):
        """W untimed warm-up steps, then EXACTLY K steps between two fences; MAX over ranks."""
        for _ in range(warmup):
            step(None)
        self.fence()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        self.fence()
        elapsed = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if self.args.backend == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def gather(self, obj):
        """One python object per rank, on every rank (the per-rank lines of an N-GPU run)."""
        if self.dist is None:
            return [obj]
        got = [None] * self.world
        self.dist.all_gather_object(got, obj)
        return got

    def rank_spread(self, out, fps_rank, frac=None):
        """n_gpus comes from the process group actually formed; per-rank min / max so that a slow rank is visible."""
        per = self.gather({"fps": float(fps_rank), "frac": None if frac is None else float(frac),
                           "dev": torch.cuda.current_device(), "numa": self.numa})
        assert len(per) == self.world == out["n_gpus"]
        devices = sorted({p["dev"] for p in per})
        if self.dist is not None and self.args.backend == "nccl":
            # one rank per GPU (train_101.sh:27-28): RCCL refuses two ranks on one device anyway; say so before it hangs
            assert len(devices) == self.world, f"{self.world} RCCL ranks on {len(devices)} distinct devices {devices}"
        out["rccl_ranks"] = self.rccl_ranks                  # measured: all_reduce(ones) at start-up (0 = not RCCL)
        out["per_rank"] = {"frames_per_s_min": round(min(p["fps"] for p in per), 1),
                           "frames_per_s_max": round(max(p["fps"] for p in per), 1),
                           "devices": devices, "backend": self.args.backend if self.dist else None,
                           "numa_nodes": [p.get("numa") for p in per]}
        if frac is not None:
            out["per_rank"]["roofline_frac_min"] = round(min(p["frac"] for p in per), 4)
            out["per_rank"]["roofline_frac_max"] = round(max(p["frac"] for p in per), 4)

    def finish(self, out):
        if self.dist is not None:
            assert self.dist.get_world_size() == self.args.gpus == out["n_gpus"]
        if self.dist is not None:
            self.dist.destroy_process_group()
        if self.rank == 0:
            emit_line(json.dumps(out))


def quick_ms(fn, n, warm=3, dev=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / n


# ---------------------------------------------------------------------------------------------------------------------
# configs[1] (default) and configs[4]: the matching layer on resident mask batches
# ---------------------------------------------------------------------------------------------------------------------
def bench_layer(R, ci):
    from dmm_net_amd import _lib, ops, synth
    args, dev, rank, world = R.args, R.dev, R.rank, R.world
    _lib.load()                                              # loud failure if the HIP extension is missing
    c = synth.CONFIGS[ci]
    B = args.frames or (1024 if ci == 2 else 512)
    N, M, H, W, D = c["P"], c["O"], c["H"], c["W"], c["D"]
    HW = H * W
    mdt = torch.float32 if ci == 2 else torch.float16
    es = 4 if ci == 2 else 2
    odt = mdt if (ci == 5 and not args.f32_out) else torch.float32   # config 5: matched masks stay in the storage type
    g = torch.Generator(device=dev).manual_seed(synth.BASE_SEED + ci + 1000 * rank)

    # config 5: the 16-bit planes sit at a 128-byte-aligned plane stride (ops.alloc_planes; the C ABI takes the stride) --
    # what a producer that owns its buffers hands over; --contiguous-planes = the packed [B,K,H,W] layout of round 3,
    # where every other 255 x 255 fp16 plane starts 2 bytes off a dword
    aligned = ci == 5 and not args.contiguous_planes

    def planes(b, k):
        if not aligned:
            t = torch.rand((b, k, H, W), generator=g, device=dev)
            return t if mdt == torch.float32 else t.to(mdt)
        t = ops.alloc_planes(b, k, H, W, mdt, dev, 128, fill=0)
        for b0 in range(0, b, 64):
            t[b0:b0 + 64].copy_(torch.rand((min(64, b - b0), k, H, W), generator=g, device=dev))
        return t

    def make_inputs(b):
        pm, tm = planes(b, N), planes(b, M)
        return (pm, tm, torch.randn((b, N, D), generator=g, device=dev), torch.randn((b, M, D), generator=g, device=dev),
                torch.rand((b, N), generator=g, device=dev))
    inputs = make_inputs(B)
    # pre-allocated plan: nothing is allocated in the timed region.  pipeline = streaming lane (cost, mix) on the
    # current stream + latency lane (normalise, cosine, solver) on a side stream (ops.ForwardPlan)
    # config 5 = "fp16 Sinkhorn with fp32 accumulate": fp16 planes, fp32 accumulation.  The solver keeps its state in fp32 -- bit
    # exact against the reference like every other configuration -- unless --f16-solver asks for the fp16-state form (tolerance
    # mode, include/dmm_match.h (3c)), which is ~1 % faster: VERDICT r5 item 7
    sstate = "f16" if (ci == 5 and getattr(args, "f16_solver", False)) else "f32"
    plan = ops.ForwardPlan(B, N, M, H, W, D, dev, mask_dtype=mdt, pipeline=False if args.no_pipeline else (True if args.pipeline else None),
                           time_kernels=True, out_dtype=odt, parts=args.parts or 2, solver_state=sstate,
                           out_plane_align=128 if aligned else 0)
    kw = dict(score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
    ev = []

    def step(k):
        plan.kernel_events = None
        if k is not None:
            plan.kernel_events = {}
            ev.append(plan.kernel_events)
        plan.run(*inputs, **kw)

    elapsed = R.timed(step, args.steps, args.warmup)
    # every frame ran the solver; a frame may take the reference's data-dependent early exit (relax_match.py:96-98)
    it_mean = float(plan.iters.float().mean())
    assert int(plan.iters.max()) == 20 and it_mean > 19.5, f"work was skipped inside the timed region ({it_mean})"
    assert bool(torch.isfinite(plan.full_outmask[-1].float()).all()) and float(plan.full_outmask[-1].float().abs().sum()) > 0

    def avg_ms(name):
        v = [a.elapsed_time(b) for per_step in ev for (a, b) in per_step.get(name, [])]
        return float(np.mean(v)) if v else None
    # dominant kernel = the IoU count kernel: HIP events around each of its launches, on the stream it runs on (the
    # pipelined plan launches it once per half of the batch)
    cost_ms, mix_ms = avg_ms("cost"), avg_ms("mix")
    launches = len(plan.halves) if plan.pipeline else 1
    fpl = B / launches
    b_cost = (N + M) * HW * es + M * N * 4                                 # B_cost per frame
    b_mix = M * HW * es + M * HW * plan.full_outmask.element_size()       # test mode: M selected planes in, M planes out
    alg_bytes = int(fpl * b_cost)
    achieved = alg_bytes / (cost_ms * 1e-3) / 1e9
    cost_kernel = "dmm::iou_counts_kernel<float,16,1>" if ci == 2 else "dmm::iou_counts_tl_kernel<__half,...>"
    fps_rank = B * args.steps / elapsed
    mix_gbs = fpl * b_mix / (mix_ms * 1e-3) / 1e9 if mix_ms else None
    out = {
        "metric": METRIC if ci == 2 else "frames/sec (cost+match layer) at N=200 proposals, M=20 templates, 255x255, "
                                         "fp16 mask planes (BASELINE configs[4])",
        "value": round(world * fps_rank, 1), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if ci == 2 else ("f16 planes (in and out)" + (", f16 solver state" if sstate == "f16" else "") + ", f32 accumulate"),
        "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[{ci - 1}]: {N} proposals x {M} templates, 255x255 "
                                f"{'fp32' if ci == 2 else 'fp16'} masks, D=512, 20 outer x 5 inner relax iterations, "
                                "forward is_test=1, uniform-random masks"
                                + ("; planes at a 128-byte-aligned plane stride (65088 elements), in and out" if aligned else "")),
                   "frames_per_gpu_per_step": B, "mean_outer_iterations": round(it_mean, 3),
                   "sharding": f"frames x{world} (no collective in the forward)", "schedule": plan.schedule_name(),
                   "solver_state": sstate},
        "roofline": {"bound": "hbm", "kernel": cost_kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "frames_per_launch": int(fpl),
                     "avg_launch_ms": round(cost_ms, 4)},
        # whole layer on SURVEY 8d's two bases, per GPU: B_cost x frames/s, and cost + test-mode mix bytes x frames/s
        "roofline_layer": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                           "b_cost_basis": {"bytes_per_frame": b_cost, "achieved": round(b_cost * fps_rank / 1e9, 1),
                                            "frac": round(b_cost * fps_rank / 1e9 / HBM_PEAK_GBS, 4)},
                           "b_layer_basis": {"bytes_per_frame": b_cost + b_mix,
                                             "achieved": round((b_cost + b_mix) * fps_rank / 1e9, 1),
                                             "frac": round((b_cost + b_mix) * fps_rank / 1e9 / HBM_PEAK_GBS, 4)}},
    }
    R.rank_spread(out, fps_rank, achieved / HBM_PEAK_GBS)
    if mix_ms:
        out["roofline_mix"] = {"bound": "hbm", "kernel": "dmm::mask_mix_rows_kernel", "achieved": round(mix_gbs, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(mix_gbs / HBM_PEAK_GBS, 4),
                               "algorithmic_bytes_per_launch": int(fpl * b_mix), "avg_launch_ms": round(mix_ms, 4)}
    extras = rank == 0 and world == 1 and not args.no_extras
    if extras:
        # frames per launch sequence: 1 and 4 are the product's sizes (one call per video / per training batch)
        sweep = {}
        for b in (1, 4, 8, 64, 512, 1024):
            if b > B:
                continue
            p2 = plan if b == B else ops.ForwardPlan(b, N, M, H, W, D, dev, mask_dtype=mdt, out_dtype=odt,
                                                     graph=b <= 32 and not aligned, solver_state=sstate,
                                                     out_plane_align=128 if aligned else 0)
            inp = inputs if b == B else tuple(t[:b] for t in inputs)
            ms = quick_ms(lambda: p2.run(*inp, **kw), 200 if b <= 64 else 30, dev=dev)
            sweep[str(b)] = {"ms": round(ms, 4), "frames_per_s": round(b / ms * 1e3, 1), "schedule": p2.schedule_name()}
        out["batch_sweep"] = sweep
        out["latency"] = {"B1_ms": sweep["1"]["ms"], "B4_ms": sweep["4"]["ms"],
                          "note": "one launch sequence (HIP-graph replay where the plan captured one), device-side "
                                  "time per call averaged over back-to-back calls"}
        del inputs, plan
        torch.cuda.empty_cache()
        if not args.no_traffic:
            pref = "dmm::iou_counts_kernel" if ci == 2 else "dmm::iou_counts_tl_kernel"
            extra = ["--config", str(ci), "--frames", str(B)] + (["--no-pipeline"] if args.no_pipeline else []) + \
                (["--contiguous-planes"] if args.contiguous_planes else [])
            t = pmc_traffic(extra, [pref, "dmm::mask_mix_rows_kernel"])
            if t is not None:
                out["roofline"]["traffic"] = t[pref]
                out["roofline"]["traffic_source"] = ("measured in this run: child passes `rocprofv3 --kernel-trace "
                                                     "--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` of this command (3 "
                                                     "steps), KiB x 1024, FETCH x 2 (gfx950)")
                if "roofline_mix" in out:
                    out["roofline_mix"]["traffic"] = t["dmm::mask_mix_rows_kernel"]
        if out["roofline"]["traffic"] is None and ci == 2:
            tv, src = profile_traffic("r01_pmc_traffic.json", int(fpl))
            out["roofline"]["traffic"] = tv
            out["roofline"]["traffic_source"] = f"NOT measured in this run; from {src}" if src else None
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, ci)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# configs[2]: ResNet-50 + heads (bf16, MIOpen) -> fused 4-level ROIAlign+mean -> the layer, batch of 8 frames
# ---------------------------------------------------------------------------------------------------------------------
def conv_flops(module, x):
    """2 x MACs of every Conv2d in one forward (hooks); the encoder's matrix work."""
    total = [0]
    hooks = []

    def hook(m, inp, out):
        total[0] += 2 * out.numel() * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
    for m in module.modules():
        if isinstance(m, torch.nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        module(x)
    for h in hooks:
        h.remove()
    return total[0]


def bench_config3(R):
    from dmm_net_amd import _lib, ops, synth
    from dmm_net_amd.encoder import FastEncoder, FeatureEncoder, GraphedEncoder, fold_batchnorm
    from dmm_net_amd.proposals import SimpleBoxList
    from dmm_net_amd.roi_features import FeatureExtractor
    args, dev, rank, world = R.args, R.dev, R.rank, R.world
    _lib.load()
    B, P, O, H, W, D = args.frames or 8, 50, 10, 255, 255, 512
    torch.manual_seed(0)
    g = torch.Generator(device=dev).manual_seed(synth.BASE_SEED + 3 + 1000 * rank)
    enc_fp32 = fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval())
    img = torch.randn(B, 3, H, W, device=dev)
    flops = conv_flops(enc_fp32, img)
    if args.nchw_encoder:        # round-1 form: NCHW, every convolution through MIOpen
        enc = GraphedEncoder(enc_fp32, weights_dtype=torch.bfloat16)
    else:
        enc = GraphedEncoder(FastEncoder(enc_fp32), miopen_find=True)   # find-db shipped in dmm_net_amd/miopen_db
    fe = FeatureExtractor()
    pm = torch.rand((B, P, H, W), generator=g, device=dev)
    tm = torch.rand((B, O, H, W), generator=g, device=dev)
    sc = torch.rand((B, P), generator=g, device=dev)

    def boxes(n):
        x1 = torch.rand(n, generator=g, device=dev) * (W - 40)
        y1 = torch.rand(n, generator=g, device=dev) * (H - 40)
        w = 8 + torch.rand(n, generator=g, device=dev) * 100
        h = 8 + torch.rand(n, generator=g, device=dev) * 100
        return torch.stack([x1, y1, (x1 + w).clamp(max=W - 1), (y1 + h).clamp(max=H - 1)], 1)
    pbox = [SimpleBoxList(boxes(P), (W, H)) for _ in range(B)]
    tbox = [SimpleBoxList(boxes(O), (W, H)) for _ in range(B)]
    plan = ops.ForwardPlan(B, P, O, H, W, D, dev)
    kw = dict(score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=1)
    ev = []

    def step(k):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if k is not None else None
        if e:
            e[0].record()
        f = enc(img)["backbone_feature"]
        if e:
            e[1].record()
        with torch.no_grad():
            a, b = fe(f, pbox).view(B, P, D), fe(f, tbox).view(B, O, D)
        if e:
            e[2].record()
        plan.run(pm, tm, a, b, sc, **kw)
        if e:
            e[3].record()
            ev.append(e)
    elapsed = R.timed(step, args.steps, args.warmup)
    assert int(plan.iters.max()) >= 1 and bool(torch.isfinite(plan.full_outmask).all())
    enc_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    roi_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    lay_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
    tflops = flops / (enc_ms * 1e-3) / 1e12
    out = {
        "metric": "frames/sec (ResNet-50 encoder + ROI features + cost+match layer), batch of 8 frames, bf16 encoder "
                  "(BASELINE configs[2])",
        "value": round(world * B * args.steps / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16 encoder, f32 matching layer", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: {B} frames of 255x255, ResNet-50 + sk/prop heads (BatchNorm "
                               "folded, bf16 weights, one HIP graph, MIOpen), 60 rois/frame x 4 levels fused "
                               "ROIAlign+mean, 50 proposals x 10 templates cost + 20x5 solver + mix; random-init weights",
                   "frames_per_gpu_per_step": B, "sharding": f"frames x{world}",
                   "stage_ms": {"encoder": round(enc_ms, 4), "roi_features": round(roi_ms, 4),
                                "matching_layer": round(lay_ms, 4)}, "schedule": plan.schedule_name()},
        "roofline": {"bound": "mfma", "kernel": "encoder graph replay: hipBLASLt GEMMs (1x1) + MIOpen implicit-GEMM (3x3, 7x7) + fused epilogues",
                     "achieved": round(tflops, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 5), "traffic": None,
                     "algorithmic_flops_per_launch": int(flops), "avg_launch_ms": round(enc_ms, 4),
                     "note": "2 x MACs of every convolution of one 8-frame forward / HIP-event time of the graph "
                             "replay; per-layer table in profiles/r02_encoder_layer_table.md"},
    }
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the evaluator's frame loop (video.FrameLoop): what DMM-Net actually runs per frame at inference
# ---------------------------------------------------------------------------------------------------------------------
def bench_frame_loop(R, T=12, reps=3, boxlist_path=True):
    from dmm_net_amd import _lib, video
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd.encoder import FastEncoder, FeatureEncoder, GraphedEncoder, fold_batchnorm
    from dmm_net_amd.proposals import SimpleBoxList
    from dmm_net_amd.roi_features import FeatureExtractor
    args, dev, rank, world = R.args, R.dev, R.rank, R.world
    _lib.load()
    B, O, H, W, P = args.frames or 4, 5, 255, 448, 50
    rng = np.random.default_rng(1000 * rank)
    torch.manual_seed(0)
    cfgs = {"matching": {"algo": "relax"}, "relax_max_iter": 40, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
            "score_weight": 0.3}
    enc = GraphedEncoder(FastEncoder(fold_batchnorm(FeatureEncoder("resnet50").to(dev).eval())), miopen_find=True)

    def raw(n):
        x1, y1 = rng.uniform(0, W - 40, n), rng.uniform(0, H - 40, n)
        bx = np.stack([x1, y1, np.minimum(x1 + rng.uniform(10, 150, n), W - 1), np.minimum(y1 + rng.uniform(10, 100, n), H - 1)], 1)
        bl = SimpleBoxList(torch.from_numpy(bx.astype(np.float32)), (W, H))
        bl.add_field("scores", torch.from_numpy(rng.random(n).astype(np.float32)))
        bl.add_field("mask", torch.from_numpy((rng.random((n, 1, 28, 28)) * 0.6 + 0.4).astype(np.float32)))
        return bl
    frames = torch.randn(B, T, 3, H, W, device=dev)
    props = [[raw(P).to(dev) for _ in range(T)] for _ in range(B)]
    first = torch.zeros(B, O, H, W, device=dev)
    for b in range(B):
        for o in range(3 + b % 3):
            y0, x0 = int(rng.integers(0, H - 60)), int(rng.integers(0, W - 60))
            first[b, o, y0:y0 + 50, x0:x0 + 55] = 1.0
    first = first.view(B, O, H * W)

    def make(slots):
        lp = video.FrameLoop(enc, DMM_Model(cfgs, is_test=1, feature_extractor=FeatureExtractor()), nms_thresh=0.4,
                             max_proposals=50)
        lp.slots = lp.graph = slots
        return lp
    loop = make(True)
    seen = []

    def clip(lp, nxt=None):
        seen.clear()
        lp.run(frames, first, props, on_labels=lambda b, t, lab: seen.append(t), next_frames=nxt)
    clip(loop)                                                    # warm-up: graph captures, MIOpen find-db lookups
    single = R.timed(lambda k: clip(loop), reps, 1) / (reps * T) * 1e3    # every clip on its own (pipeline fill included)
    # the evaluator walks a list of clips (evaluator.py:63-70): each run is told the next clip's frames and issues their
    # first encoder chunk under its own last steps
    elapsed = R.timed(lambda k: clip(loop, frames), reps, 1)
    assert len(seen) == B * T
    ms = elapsed / (reps * T) * 1e3
    out = {
        "metric": "frames/sec (evaluator frame loop: ResNet-50 encoder + proposal paste/NMS + ROI features + cost+match "
                  "layer + label merge), 4 videos of 255x448",
        "value": round(world * B * T * reps / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": reps * T,
        "warmup": T, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 encoder, f32 matching layer", "data": "synthetic",
        "config": {"workload": f"video.FrameLoop: {B} videos x {T} frames of 255x448, {P} raw 28x28 proposals per frame "
                               "(two-phase paste + NMS(0.4) + top-50 on the device), 5 template slots, eval solver "
                               "setting 40 outer x 5 inner, label merge; refine decoder out of scope (None); one step = "
                               "one frame of all videos; fixed-slot step replayed from one HIP graph, encoder chunks of "
                               f"{loop._frames_per_chunk(T, True)} frames on a side stream; clips back to back, the next "
                               "clip's first encoder chunk issued under the current clip's last steps",
                   "videos_per_gpu": B, "frames_per_clip": T, "sharding": f"videos x{world}",
                   "single_clip_ms_per_step": round(single, 4)},
    }
    if rank == 0 and T == 12:
        # the same loop on a clip of 36 frames (9 frames per encoder chunk): the first chunk is pipeline fill, longer
        # clips amortise it -- DAVIS / YouTube-VOS clips are 20 to 100 frames
        T2 = 36
        frames2 = torch.randn(B, T2, 3, H, W, device=dev)
        props2 = [[props[b][t % T] for t in range(T2)] for b in range(B)]
        loop.run(frames2, first, props2, on_labels=lambda b, t, lab: None, next_frames=frames2)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        loop.run(frames2, first, props2, on_labels=lambda b, t, lab: None, next_frames=frames2)
        torch.cuda.synchronize(dev)
        out["config"]["clip_of_36_frames_ms_per_step"] = round((time.perf_counter() - t0) / T2 * 1e3, 4)
        del frames2
    if boxlist_path and rank == 0:
        old = make(False)
        clip(old)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        clip(old)
        torch.cuda.synchronize(dev)
        out["config"]["boxlist_path_ms_per_step"] = round((time.perf_counter() - t0) / T * 1e3, 4)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: the per-GPU share of the 8-GPU training job -- ResNet-101, clips sharded per GPU, RCCL gradient mean
# ---------------------------------------------------------------------------------------------------------------------
def bench_config4(R):
    from dmm_net_amd import _lib
    from dmm_net_amd.distributed import GradBucketer
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd.encoder import FeatureEncoder
    from dmm_net_amd.proposals import SimpleBoxList
    from dmm_net_amd.roi_features import FeatureExtractor
    import torch.distributed as dist
    args, dev, rank, world = R.args, R.dev, R.rank, R.world
    _lib.load()
    own_group = False
    if not dist.is_initialized():                                # N = 1: a one-rank RCCL group, same code path
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl" if args.backend == "nccl" else args.backend, rank=0, world_size=1,
                                **({"device_id": dev} if args.backend == "nccl" else {}))
        own_group = True
    B, F, P, H, W = args.frames or 12, 5, 50, 255, 448           # 4 videos x clip 3 (train_101.sh: batch 4, clip_len 3)
    torch.manual_seed(rank)
    g = torch.Generator(device=dev).manual_seed(rank)
    enc = FeatureEncoder("resnet101").to(dev).train()
    bf16 = bool(getattr(args, "bf16", False))
    if bf16:
        # same parameters (the optimiser and the bucketer below hold the fp32 masters); no gradient reaches the decoder's skip
        # inputs in this step, like in the fp32 line (autograd skips them there)
        from dmm_net_amd.train_encoder import TrainEncoder
        run_enc = TrainEncoder(enc, skips_need_grad=False)    # (its convolution shapes ship in dmm_net_amd/miopen_db: no search)
    else:
        run_enc = enc
    # BatchNorm statistics per FRAME STEP (the 4 videos of one frame), as trainer.py:95-131's one-encoder-call-per-frame loop
    # computes them: the clip's frames are one batch for the convolutions and 3 statistics groups for BatchNorm
    bn_groups = int(getattr(args, "bn_groups", 0) or 0) or (3 if bf16 and B % 3 == 0 else 1)
    if not bf16:
        bn_groups = 1                                          # (the stock modules: one call = one statistics group)
    model = DMM_Model({"matching": {"algo": "relax"}, "relax_max_iter": 10, "relax_proj_iter": 5,
                       "relax_learning_rate": 0.1, "score_weight": 0.3}, is_test=0, feature_extractor=FeatureExtractor())
    # forward order (body, then heads): the bucketer lays its buckets out in REVERSE parameter order = the order the backward
    # produces gradients in (heads, layer4, layer3, ..., stem), so the last bucket to complete -- the only one whose all-reduce
    # nothing hides -- is the small tail (layer1 / stem), not the heads that were ready first
    params = list(enc.get_backbone_para()) + list(enc.get_skip_params())
    opt = torch.optim.Adam(params, lr=1e-4, fused=True)          # one multi-tensor kernel (foreach form: 2.3 ms per step)
    # (steady mode is opt-in: every synthetic step uses the same parameters -- a fixed graph -- so the used-mask exchange
    # may run one step late; a trainer whose videos can be skipped keeps the default per-step exchange)
    bucketer = GradBucketer(params, bucket_mb=64.0, overlap=True, steady_after=2)
    img = torch.randn(B, 3, H, W, device=dev)

    def boxes(n):
        x1 = torch.rand(n, generator=g, device=dev) * (W - 60)
        y1 = torch.rand(n, generator=g, device=dev) * (H - 60)
        return torch.stack([x1, y1, x1 + 10 + torch.rand(n, generator=g, device=dev) * 150,
                            y1 + 10 + torch.rand(n, generator=g, device=dev) * 100], 1).clamp(max=W - 1)
    props, tboxes = [], []
    for b in range(B):
        bl = SimpleBoxList(boxes(P), (W, H))
        bl.add_field("mask", torch.rand((P, 1, H, W), generator=g, device=dev))
        bl.add_field("scores", torch.rand(P, generator=g, device=dev))
        props.append(bl)
        tboxes.append(SimpleBoxList(boxes(F), (W, H)))
    mask_last = torch.rand((B, F, H, W), generator=g, device=dev)
    targets = (torch.rand((B, F, H, W), generator=g, device=dev) > 0.5).float()
    valid = torch.ones(B, F, device=dev)
    ev, losses = [], []

    def step(k):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if k is not None else None
        mark = (lambda i: e[i].record()) if e else (lambda i: None)
        mark(0)
        if args.autocast and not bf16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                feats = enc(img)
        else:
            feats = run_enc(img, bn_groups=bn_groups) if bf16 else run_enc(img)   # fp32 like the reference's trainer (train.py: no autocast),
            #                                                       or the bf16 training form (--bf16)
        mark(1)
        tplt = model.fill_template_dict(None, tboxes, feats, None, valid)
        out, _, match_loss, _ = model(None, props, feats["backbone_feature"], mask_last, tplt, valid, targets)
        soft = 1.0 - (out * targets).flatten(1).sum(1) / ((out + targets - out * targets).flatten(1).sum(1) + 1e-6)
        loss = soft.mean() + sum(match_loss) / B
        mark(2)
        opt.zero_grad()
        loss.backward()                                           # buckets are all-reduced under the backward
        mark(3)
        bucketer.finish()                                         # what is still in flight / not issued: exposed
        mark(4)
        opt.step()
        mark(5)
        if e:
            ev.append(e)
            losses.append(loss.detach())
    # settle first (not part of W or K): MIOpen picks / tunes its backward solvers on the first calls, the caching allocator
    # reaches its working set, and the bucketer leaves its per-step used-mask exchange after two identical masks -- the
    # timed steps are the steady state a long training job runs in
    settle = max(int(args.settle), 0)                            # the same count on every rank (collectives inside)
    for _ in range(settle):
        step(None)
    syncs0 = bucketer.host_syncs
    # >= 3 repeats of K timed steps each (the step follows the box: MIOpen's picks, clocks): median, min and max in the line
    reps = max(1, int(getattr(args, "repeats", 3)))
    rep_ms, rep_ev = [], []
    for r in range(reps):
        ev.clear()
        rep_ms.append(R.timed(step, args.steps, args.warmup if r == 0 else 0) / args.steps * 1e3)
        rep_ev.append(list(ev))
    order = sorted(range(reps), key=lambda k: rep_ms[k])
    mid = order[len(order) // 2]
    ev[:] = rep_ev[mid]                                          # stage times of the median repeat
    elapsed = rep_ms[mid] * args.steps / 1e3
    assert bool(torch.isfinite(torch.stack(losses)).all())
    avg = lambda i, j: float(np.mean([e[i].elapsed_time(e[j]) for e in ev]))
    exposed_all = [float(np.mean([e[3].elapsed_time(e[4]) for e in evs])) for evs in rep_ev]
    grad_bytes = sum(p.numel() * p.element_size() for p in params)
    # the collective alone (nothing to hide under): bus bandwidth of the bucketed all-reduce
    torch.cuda.synchronize(dev)
    plain = GradBucketer(params, bucket_mb=64.0, overlap=False)
    bucketer.remove_hooks()
    for _ in range(2):
        plain.all_reduce_mean()
    R.fence()
    t0 = time.perf_counter()
    for _ in range(5):
        plain.all_reduce_mean()
    R.fence()
    ar_ms = (time.perf_counter() - t0) / 5 * 1e3
    step_ms = elapsed / args.steps * 1e3
    algbw = grad_bytes / (ar_ms * 1e-3) / 1e9
    busbw = algbw * (2.0 * (world - 1) / world if world > 1 else 0.0)
    out = {
        "metric": "frames/sec (training step: ResNet-101 encoder + ROI features + matching layer forward/backward + Adam, "
                  "gradient mean over RCCL), YouTube-VOS-shaped synthetic clips (BASELINE configs[3])",
        "value": round(world * B / (step_ms * 1e-3), 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(step_ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": ("bf16 encoder (TrainEncoder: fp32 master weights, bf16 channels-last activations, "
                                       "HIP-graph replays)" if bf16 else "bf16 autocast encoder" if args.autocast
                                       else "f32 encoder (as the reference trains)") +
        ", f32 matching layer and optimiser", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3], per-GPU share: {B} frames of 255x448 (4 videos x clip 3), ResNet-101 "
                               "+ heads (" + ("train_encoder.TrainEncoder: captured segments, dmm BatchNorm / weight-gradient "
                               "kernels, hipBLASLt 1x1s, MIOpen 3x3 forward / data gradient" if bf16 else "torch.nn on MIOpen") +
                               ", random init), 50 proposals, 5 template slots, DMM_Model "
                               "training forward (10x5 solver, dual IoU with the targets) + soft-IoU + matching loss, "
                               "backward, Adam; gradient mean = distributed.GradBucketer(overlap=True, 64 MB buckets)",
                   "frames_per_gpu_per_step": B, "sharding": f"clips x{world}, one RCCL all-reduce per bucket",
                   "batchnorm_statistics": (f"{bn_groups} groups of {B // bn_groups} images = per frame step, what the reference's "
                                            "one-encoder-call-per-frame loop computes (trainer.py:95-131)" if bn_groups > 1 else
                                            f"one group: all {B} images of the call"),
                   "stage_ms": {"encoder_fwd": round(avg(0, 1), 3), "roi_layer_loss_fwd": round(avg(1, 2), 3),
                                "backward_incl_overlapped_allreduce": round(avg(2, 3), 3),
                                "allreduce_exposed_after_backward": round(avg(3, 4), 3), "adam": round(avg(4, 5), 3)},
                   "settle_steps_before_warmup": settle,
                   "repeats": {"n": reps, "steps_each": args.steps, "ms_per_step": [round(v, 3) for v in rep_ms],
                               "ms_per_step_min_median_max": [round(min(rep_ms), 3), round(rep_ms[mid], 3), round(max(rep_ms), 3)],
                               "allreduce_exposed_ms": [round(v, 3) for v in exposed_all]},
                   "note_n1": "at N = 1 the collective moves nothing between GPUs: this line shows the per-GPU share and the "
                              "bucketing / hook overhead, NOT the overlap of communication with the backward" if world == 1 else None,
                   "gradient_mean": {"bytes": grad_bytes, "buckets": bucketer.num_collectives(),
                                     "gradients_are_bucket_views": True, "divide": "in the collective (ReduceOp.AVG)"
                                     if args.backend == "nccl" else "SUM + one divide (backend has no AVG)",
                                     "used_mask_mode": bucketer.mode,
                                     "host_reads_in_timed_steps": bucketer.host_syncs - syncs0,
                                     "allreduce_alone_ms": round(ar_ms, 3),
                                     "hidden_ms": round(max(ar_ms - avg(3, 4), 0.0), 3),
                                     "algbw_GBps": round(algbw, 1), "busbw_GBps": round(busbw, 1),
                                     "xgmi_per_link_GBps": 153.0, "xgmi_links_per_gpu": 7,
                                     "note": "busbw = algbw x 2(N-1)/N (ring all-reduce); a single ring is bound by one "
                                             "xGMI link (~153 GB/s), 7 rings by 7 x 153 GB/s; N = 1 moves nothing"}},
    }
    if own_group:
        dist.destroy_process_group()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the TRAINING form of the layer (is_test = 0, targets): forward + backward, per-kernel roofline
# ---------------------------------------------------------------------------------------------------------------------
def bench_train(R, Bs=(64, 512)):
    """configs[1]'s shape through the layer as the trainer calls it (dmm_model.py:130-132: is_test=0 with targets):
    dual IoU counts (templates AND targets in one pass over the proposals, match_helper.py:30-49 + match_model.py:83-89),
    cosine, solver, the train-mode mix (every plane with R > 0.01, match_model.py:126-129,144), matching loss; backward
    through mix, taped solver and feature similarity.  ``value`` = frames/s of forward + backward through autograd
    (``autograd.match_layer_batched``) at the larger batch; every HBM-bound kernel gets HIP-event time / algorithmic bytes."""
    from dmm_net_amd import _lib, autograd, ops, synth
    args, dev, rank, world = R.args, R.dev, R.rank, R.world
    _lib.load()
    c = synth.CONFIGS[2]
    N, M, H, W, D = c["P"], c["O"], c["H"], c["W"], c["D"]
    HW = H * W
    cfg = dict(score_weight=0.3, max_iter=20, proj_iter=5, lr=0.1, is_test=0)
    g = torch.Generator(device=dev).manual_seed(synth.BASE_SEED + 7 + 1000 * rank)
    per_B = {}
    for B in ((args.frames,) if args.frames else Bs):
        pm = torch.rand((B, N, H, W), generator=g, device=dev)
        tm = torch.rand((B, M, H, W), generator=g, device=dev)
        tg = (torch.rand((B, M, H, W), generator=g, device=dev) > 0.5).float()
        pf = torch.randn((B, N, D), generator=g, device=dev, requires_grad=True)
        tf = torch.randn((B, M, D), generator=g, device=dev, requires_grad=True)
        sc = torch.rand((B, N), generator=g, device=dev)
        dfull = torch.rand((B, M, H, W), generator=g, device=dev)          # upstream gradient of full_outmask (a loss map)

        def step(k):
            pf.grad = tf.grad = None
            full, ms, ds, loss, _ = autograd.match_layer_batched(pf, pm, tf, tm, sc, tg, **cfg)
            torch.autograd.backward([full, loss], [dfull, torch.ones_like(loss)])
        steps = max(5, args.steps // (4 if B >= 256 else 1))
        elapsed = R.timed(step, steps, max(2, args.warmup // 2))
        assert bool(torch.isfinite(pf.grad).all()) and float(pf.grad.abs().sum()) > 0 and bool(torch.isfinite(tf.grad).all())
        # ---- the same kernels one by one, HIP events on the stream they run on --------------------------------------
        with torch.no_grad():
            (inter, ap, at), (gi, gat) = ops.iou_counts_dual(pm, tm, tg)
            pn, pnorm = ops.feature_normalize(pf, want_norms=True)
            tn, tnorm = ops.feature_normalize(tf, want_norms=True)
            cos = ops.cosine(tn, pn)
            r = ops.relax_match(cos, inter, ap, at, sc, **cfg)
            Rb, sim = r["Rb"], r["sim"]
            nnz = int((Rb[:, :, :N] != 0).sum())                           # (row, plane) pairs with R > 0.01
            n_union = int((Rb[:, :, :N] != 0).any(1).sum())                # distinct planes the rows of a frame select
            loss, gt, _ = autograd.matching_loss(pm, tg, cos, counts=(gi, ap, gat))
            dRb = ops.mask_mix_bwd(Rb, pm, dfull)
            dsim = ops.relax_match_bwd(sim, sc, dRb, None, None, max_iter=20, proj_iter=5, lr=0.1, is_test=0)
            reps = 20 if B <= 64 else 8
            t = {
                "iou_counts_dual": quick_ms(lambda: ops.iou_counts_dual(pm, tm, tg), reps, dev=dev),
                "feature_sim_fwd (normalise x2 + cosine)": quick_ms(lambda: ops.cosine(
                    ops.feature_normalize(tf, want_norms=True)[0], ops.feature_normalize(pf, want_norms=True)[0]), reps, dev=dev),
                "relax_match (train mode)": quick_ms(lambda: ops.relax_match(cos, inter, ap, at, sc, **cfg), reps, dev=dev),
                "mask_mix (train mode)": quick_ms(lambda: ops.mask_mix(Rb, pm, shared=True), reps, dev=dev),
                "matching_loss (greedy one-hot + mse)": quick_ms(lambda: autograd.matching_loss(pm, tg, cos, counts=(gi, ap, gat)),
                                                                 reps, dev=dev),
                "mask_mix_bwd": quick_ms(lambda: ops.mask_mix_bwd(Rb, pm, dfull), reps, dev=dev),
                "relax_match_bwd": quick_ms(lambda: ops.relax_match_bwd(sim, sc, dRb, None, None, max_iter=20, proj_iter=5,
                                                                        lr=0.1, is_test=0), reps, dev=dev),
                "feature_sim_bwd": quick_ms(lambda: ops.feature_sim_bwd(dsim, cos, gt, torch.ones_like(loss), 0.3, tf, pf,
                                                                        tn, pn, tnorm, pnorm), reps, dev=dev),
            }
        # algorithmic bytes per launch (fp32 planes): every plane the kernel must see, once
        alg = {
            "iou_counts_dual": B * ((N + 2 * M) * HW * 4 + 2 * M * N * 4),               # proposals + templates + targets
            # every selected plane ONCE (the union of the rows' supports) + M planes out / M planes of d full_outmask in
            "mask_mix (train mode)": n_union * HW * 4 + B * M * HW * 4,
            "mask_mix_bwd": n_union * HW * 4 + B * M * HW * 4,
        }
        kern = {}
        for k, ms in t.items():
            e = {"ms": round(ms, 4)}
            if k in alg:
                gbs = alg[k] / (ms * 1e-3) / 1e9
                e.update(bound="hbm", algorithmic_bytes=int(alg[k]), achieved_GBps=round(gbs, 1),
                         frac=round(gbs / HBM_PEAK_GBS, 4))
            else:
                e.update(bound="latency / VALU (no plane traffic)")
            kern[k] = e
        per_B[str(B)] = {"fwd_bwd_ms": round(elapsed / steps * 1e3, 4), "frames_per_s": round(B * steps / elapsed, 1),
                         "steps": steps, "selected_planes_per_frame": round(n_union / B, 2),
                         "selected_row_plane_pairs_per_frame": round(nnz / B, 2),
                         "kernel_sum_ms": round(sum(t.values()), 4), "kernels": kern,
                         "note": "fwd_bwd_ms = the fused training entries (the forward keeps the solver's tape, the backward "
                                 "walks it); the per-kernel rows time the granular entries one by one -- their "
                                 "relax_match_bwd re-runs the solver first (about half of its time)"}
        del pm, tm, tg, dfull
        torch.cuda.empty_cache()
    big = per_B[max(per_B, key=int)]
    Bbig = int(max(per_B, key=int))
    dual = big["kernels"]["iou_counts_dual"]
    out = {
        "metric": "frames/sec (cost+match layer, TRAINING form: is_test=0 with targets, forward + backward) at N=50 "
                  "proposals, M=10 templates, 255x255",
        "value": round(world * big["frames_per_s"], 1), "unit": "frames/s", "n_gpus": world, "steps": big["steps"],
        "warmup": max(2, args.warmup // 2), "ms_per_step": big["fwd_bwd_ms"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1] shape in the trainer's form: {N} proposals x {M} templates, 255x255 fp32 "
                               "masks, D=512, targets, is_test=0 (mix over every plane with R > 0.01), 20 x 5 relax "
                               "iterations, matching loss; forward + backward through autograd.match_layer_batched; "
                               "uniform-random masks",
                   "frames_per_gpu_per_step": Bbig, "by_batch": per_B},
        "roofline": {"bound": "hbm", "kernel": "dmm::iou_counts_tl_kernel / iou_counts_kernel (dual: templates + targets)",
                     "achieved": dual["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dual["frac"],
                     "traffic": None, "algorithmic_bytes_per_launch": dual["algorithmic_bytes"],
                     "frames_per_launch": Bbig, "avg_launch_ms": dual["ms"]},
    }
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the drop-in itself: what ONE call of the reference's classes costs, host included
# ---------------------------------------------------------------------------------------------------------------------
def bench_dropin(R):
    """north_star's deliverable is ``MatchModel`` dropped into train.py / eval.py, called once per (video, frame)
    (dmm/modules/dmm_model.py:75-77 evaluator, :130-132 trainer).  Per call, for each case: WALL time (host clock around n
    back-to-back calls + one synchronize -- what the caller's loop pays), DEVICE time (HIP events around the same calls)
    and the library's kernel launches per call (``dmm_launch_count``).  wall > device = the call is host bound."""
    from dmm_net_amd import _lib, autograd
    from dmm_net_amd.dmm_model import DMM_Model
    from dmm_net_amd.match_model import MatchModel
    from dmm_net_amd.proposals import SimpleBoxList
    args, dev, rank, world = R.args, R.dev, R.rank, R.world
    L = _lib.load()
    H, W, D = 255, 448, 512
    g = torch.Generator(device=dev).manual_seed(4321 + 1000 * rank)
    n_calls = max(50, args.steps)

    def cfgs(mi):
        return {"matching": {"algo": "relax"}, "relax_max_iter": mi, "relax_proj_iter": 5, "relax_learning_rate": 0.1,
                "score_weight": 0.3}

    def frame(P, O):
        return dict(pf=torch.randn((P, D), generator=g, device=dev), tf=torch.randn((O, D), generator=g, device=dev),
                    pm=torch.rand((P, H, W), generator=g, device=dev), tm=torch.rand((O, H, W), generator=g, device=dev),
                    sc=torch.rand((P,), generator=g, device=dev),
                    tg=(torch.rand((O, H, W), generator=g, device=dev) > 0.5).float())

    # device time of a call with the HOST OUT OF THE WAY: the calls are enqueued while the GPU is still busy with a long
    # blocker (device-to-device copies), so they run back to back from a full queue; HIP events around them
    blk_src = torch.empty((1 << 28,), dtype=torch.float32, device=dev)       # 1 GiB
    blk_dst = torch.empty_like(blk_src)

    def measure(call, n=n_calls, reps=3):
        """median (and min / max) over ``reps`` repeats of n back-to-back calls: wall us per call (host clock, one sync at
        the end), device us per call (events around the same calls queued behind a blocker) and launches per call."""
        import gc
        for _ in range(10):
            call()
        torch.cuda.synchronize(dev)
        # like timeit: no cyclic garbage collection inside the timed loops (inside the default run this case follows five
        # others in the same process -- a ResNet-101 training step among them -- and a generation-2 pass over everything
        # they left tracked cost the 0.2 ms calls here 25-40 % of wall clock: 240 us against 190-205 standalone)
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        walls, devs = [], []
        l0 = L.dmm_launch_count()
        call()
        launches = int(L.dmm_launch_count() - l0)
        torch.cuda.synchronize(dev)
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(n):
                call()
            torch.cuda.synchronize(dev)
            walls.append((time.perf_counter() - t0) / n * 1e6)
        copy_ms = quick_ms(lambda: blk_dst.copy_(blk_src), 3, warm=1, dev=dev)
        for _ in range(reps):
            n_blk = int(min(400, max(4, 1.5 * n * walls[0] * 1e-3 / copy_ms + 2)))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(n_blk):
                blk_dst.copy_(blk_src)
            a.record()
            for _ in range(n):
                call()
            b.record()
            torch.cuda.synchronize(dev)
            devs.append(a.elapsed_time(b) / n * 1e3)
        if gc_was:
            gc.enable()
        med = lambda v: sorted(v)[len(v) // 2]
        return {"wall_us": round(med(walls), 1), "wall_us_min_max": [round(min(walls), 1), round(max(walls), 1)],
                "device_us": round(med(devs), 1), "library_launches": launches,
                "wall_over_device": round(med(walls) / max(med(devs), 1e-9), 3), "calls": n, "repeats": reps}

    # one process per GPU, its host threads next to that GPU -- what a trainer / evaluator process does at start-up
    # (INTEGRATION.md); at N = 1 the mask is put back afterwards: the cpu_baseline leg uses every core
    from dmm_net_amd.distributed import bind_host_threads_to_gpu
    numa_node, unbind = bind_host_threads_to_gpu(dev.index) if world == 1 else (R.numa, lambda: None)
    cases = {}
    # (a) the evaluator's call: MatchModel(cfg, is_test=1).forward under no_grad, eval solver setting 40 x 5
    ev = MatchModel(cfgs(40), is_test=1)
    for (P, O) in ((50, 5), (50, 10)):
        f = frame(P, O)

        def call_eval(f=f):
            with torch.no_grad():
                return ev(f["pf"], f["pm"], [f["tf"]], f["tm"], f["sc"])
        cases[f"eval_forward_{P}x{O}"] = dict(measure(call_eval), what=f"MatchModel(cfgs, is_test=1).forward, {P} proposals x "
                                              f"{O} templates, {H}x{W} fp32, 40 x 5 iterations, no_grad "
                                              "(dmm_model.py:75-77)")
    # (b) the trainer's call: MatchModel(cfg, 0).forward with targets + backward, train solver setting 10 x 5
    tr = MatchModel(cfgs(10), is_test=0)
    f = frame(50, 5)
    pf = f["pf"].clone().requires_grad_(True)
    tf = f["tf"].clone().requires_grad_(True)
    dfull = torch.rand((5, H, W), generator=g, device=dev)
    one = torch.ones((), device=dev)

    def call_train():
        pf.grad = tf.grad = None
        fo, ms, ds, _, loss = tr(pf, f["pm"], [tf], f["tm"], f["sc"], f["tg"])
        torch.autograd.backward([fo, loss["cost_loss"]], [dfull, one])
    what_b = (f"MatchModel(cfgs, 0).forward with targets + backward(), 50 proposals x 5 templates, {H}x{W} fp32, 10 x 5 "
              "iterations (dmm_model.py:130-132)")
    cases["train_fwd_bwd_50x5"] = dict(measure(call_train), what=what_b)
    assert pf.grad is not None and bool(torch.isfinite(pf.grad).all()) and float(pf.grad.abs().sum()) > 0
    # the same call with autograd's backward on the CALLING thread (torch.autograd.set_multithreading_enabled(False), one line
    # in a trainer that owns one GPU per process): no hand-off to the device thread and back per backward()
    with torch.autograd.set_multithreading_enabled(False):
        cases["train_fwd_bwd_50x5_autograd_on_calling_thread"] = dict(
            measure(call_train), what=what_b + " -- under torch.autograd.set_multithreading_enabled(False)")
    old = autograd._FUSED_TRAIN
    autograd._FUSED_TRAIN = False                                # the pre-fusion chain, for the record
    try:
        cases["train_fwd_bwd_50x5_granular_chain"] = dict(measure(call_train), what=what_b + " -- through the granular entries "
                                                          "(12 library calls + tensor ops), what round 4 shipped")
    finally:
        autograd._FUSED_TRAIN = old
    # (c) DMM_Model for 4 videos (all videos of the step through one ragged launch sequence).  The valid-flags tensor is the
    # same object call after call, as the reference passes it for the frames of a clip (trainer.py:113-121): its layout is
    # read from the device once per clip, a frame step after the first has no host sync
    B, F, P = 4, 5, 50

    def boxes(n):
        x1 = torch.rand(n, generator=g, device=dev) * (W - 60)
        y1 = torch.rand(n, generator=g, device=dev) * (H - 60)
        return torch.stack([x1, y1, x1 + 10 + torch.rand(n, generator=g, device=dev) * 150,
                            y1 + 10 + torch.rand(n, generator=g, device=dev) * 100], 1).clamp(max=W - 1)
    props = []
    for b in range(B):
        bl = SimpleBoxList(boxes(P), (W, H))
        bl.add_field("mask", torch.rand((P, 1, H, W), generator=g, device=dev))
        bl.add_field("scores", torch.rand(P, generator=g, device=dev))
        props.append(bl)
    feats_p = [torch.randn((P, D), generator=g, device=dev) for _ in range(B)]
    feats_pg = [x.clone().requires_grad_(True) for x in feats_p]
    tplt = {b: {"feat": [torch.randn((F, D), generator=g, device=dev)], "refine_input_feat": [()]} for b in range(B)}
    mask_last = torch.rand((B, F, H, W), generator=g, device=dev)
    targets = (torch.rand((B, F, H, W), generator=g, device=dev) > 0.5).float()
    valid = torch.ones(B, F, device=dev)
    # the ROI feature extractor is not what is measured here: hand the rows over
    m_ev = DMM_Model(cfgs(40), is_test=1, feature_extractor=lambda feats, pr: torch.cat(feats_p, 0))
    m_tr = DMM_Model(cfgs(10), is_test=0, feature_extractor=lambda feats, pr: torch.cat(feats_pg, 0))
    infos = {"extra_frame": [False] * B, "valid": valid}

    def call_inf():
        with torch.no_grad():
            m_ev.inference(infos, props, None, mask_last, tplt)

    def call_fwd():
        for x in feats_pg:
            x.grad = None
        out, _, ml, _ = m_tr(None, props, None, mask_last, tplt, valid, targets)
        (out.sum() + sum(ml)).backward()
    # ... and for ONE video, the evaluator's batch (scripts/eval/*.sh: -batch_size=1)
    m_ev1 = DMM_Model(cfgs(40), is_test=1, feature_extractor=lambda feats, pr: feats_p[0])
    infos1 = {"extra_frame": [False], "valid": valid[:1]}

    def call_inf1():
        with torch.no_grad():
            m_ev1.inference(infos1, props[:1], None, mask_last[:1], {0: tplt[0]})
    cases["dmm_model_inference_1_video"] = dict(measure(call_inf1), what=f"DMM_Model.inference, 1 video x {P} proposals x {F} "
                                                f"templates, {H}x{W}, 40 x 5 (the evaluator's batch, scripts/eval/eval.sh:6)")
    cases["dmm_model_inference_4_videos"] = dict(measure(call_inf, n=max(20, n_calls // 4)),
                                                 what=f"DMM_Model.inference, {B} videos x {P} proposals x {F} templates, {H}x{W}, "
                                                      "40 x 5 (dmm_model.py:48-86); ROI feature rows handed over")
    with torch.autograd.set_multithreading_enabled(False):
        cases["dmm_model_forward_backward_4_videos_autograd_on_calling_thread"] = dict(
            measure(call_fwd, n=max(20, n_calls // 4)), what=f"DMM_Model.forward + backward, {B} videos, 10 x 5, under "
                                                             "torch.autograd.set_multithreading_enabled(False)")
    cases["dmm_model_forward_backward_4_videos"] = dict(measure(call_fwd, n=max(20, n_calls // 4)),
                                                        what=f"DMM_Model.forward + backward, {B} videos, 10 x 5 "
                                                             "(dmm_model.py:88-142)")
    unbind()
    head = cases["eval_forward_50x5"]
    out = {
        "metric": "calls/sec of the drop-in (MatchModel.forward as the evaluator calls it: 50 proposals x 5 templates, "
                  "255x448, one frame per call), host inclusive",
        "value": round(world * 1e6 / head["wall_us"], 1), "unit": "calls/s", "n_gpus": world, "steps": head["calls"],
        "warmup": 10, "ms_per_step": round(head["wall_us"] / 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "the drop-in classes called as the reference calls them (one frame of one video per "
                               "MatchModel call; 4 videos per DMM_Model call); wall = host clock around n back-to-back calls "
                               "+ one synchronize, device = HIP events around the same calls enqueued behind a blocker (the host "
                               "runs ahead: back-to-back device time); median of 3 repeats",
                   "host_threads": (f"pinned to the CPUs of NUMA node {numa_node} (the GPU's)" if numa_node is not None
                                    else "unpinned (GPU topology not visible)"),
                   "cases": cases},
    }
    return out


def compact(out):
    """What ``other_configs`` keeps of a workload's line."""
    c = {"metric": out["metric"], "value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"],
         "steps": out["steps"], "workload": out["config"]["workload"]}
    if "roofline" in out:
        c["roofline"] = {k: out["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac")
                         if k in out["roofline"]}
    if "roofline_layer" in out:
        c["roofline_layer_b_cost_frac"] = out["roofline_layer"]["b_cost_basis"]["frac"]
    for k in ("stage_ms", "single_clip_ms_per_step", "boxlist_path_ms_per_step", "clip_of_36_frames_ms_per_step",
              "mean_outer_iterations", "repeats", "solver_state", "f16_solver", "contiguous_planes"):
        if k in out["config"]:
            c[k] = out["config"][k]
    if "cases" in out["config"]:                                 # the drop-in: per-call wall / device / launches
        c["cases"] = {k: {kk: v[kk] for kk in ("wall_us", "device_us", "library_launches", "wall_over_device")}
                      for k, v in out["config"]["cases"].items()}
        c["host_threads"] = out["config"].get("host_threads")
    if "by_batch" in out["config"]:                              # training form: per-kernel ms / fraction of the HBM peak
        c["by_batch"] = {b: {"fwd_bwd_ms": v["fwd_bwd_ms"], "selected_planes_per_frame": v["selected_planes_per_frame"],
                             "kernels": {k: ({"ms": e["ms"], "frac": e["frac"]} if "frac" in e else {"ms": e["ms"]})
                                         for k, e in v["kernels"].items()}}
                         for b, v in out["config"]["by_batch"].items()}
    return c


_REAL_STDOUT = None


def own_stdout():
    """ONE JSON line on stdout, whatever the libraries under this process print: RCCL writes its version banner with C
    stdio to fd 1 when a communicator is created (it surfaced behind the JSON line once the default run created a one-rank
    group for config 4; every rank of an N-GPU run prints it too).  The real stdout is kept aside for the line and fd 1
    points at stderr from here on."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(text):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(text, flush=True)
    else:
        os.write(_REAL_STDOUT, (text + "\n").encode())


def main():
    args = parse()
    self_launch(args)
    own_stdout()
    R = Runner(args)
    if args.config == 3:
        out = bench_config3(R)
    elif args.config == 4:
        out = bench_config4(R)
    elif args.config == "loop":
        out = bench_frame_loop(R)
    elif args.config == "train":
        out = bench_train(R)
    elif args.config == "dropin":
        out = bench_dropin(R)
    else:
        out = bench_layer(R, args.config)
    if "per_rank" not in out:
        R.rank_spread(out, out["value"] / R.world)
    if args.config == 2 and R.rank == 0 and R.world == 1 and not args.no_extras and not args.no_others:
        # the other BASELINE configurations and the frame loop, compact, so that the driver's default run sees them
        import copy
        others = {}
        def config5_both(r):
            """BASELINE configs[4] on fp16 planes: the bit-exact fp32-state solver (default), the fp16-state figure beside it"""
            o = bench_layer(r, 5)
            a3 = copy.copy(r.args)
            a3.f16_solver = True
            keep, r.args = r.args, a3
            try:
                o32 = bench_layer(r, 5)
            finally:
                r.args = keep
            a4 = copy.copy(r.args)
            a4.contiguous_planes = True
            r.args = a4
            try:
                oc = bench_layer(r, 5)                                # the packed [B,K,H,W] layout beside the aligned one
            finally:
                r.args = keep
            o["config"]["contiguous_planes"] = {"value": oc["value"], "unit": "frames/s", "ms_per_step": oc["ms_per_step"],
                                                "roofline_frac": oc["roofline"]["frac"],
                                                "note": "packed [B,K,255,255] fp16 planes (every other plane starts 2 bytes "
                                                        "off a dword) instead of the 128-byte-aligned plane stride"}
            o["config"]["f16_solver"] = {"value": o32["value"], "unit": "frames/s", "ms_per_step": o32["ms_per_step"],
                                         "roofline_frac": o32["roofline"]["frac"],
                                         "roofline_layer_b_cost_frac": o32["roofline_layer"]["b_cost_basis"]["frac"],
                                         "note": "the fp16-state solver (--f16-solver; tolerance mode) on the same planes"}
            return o
        for name, fn, kw in (("config5", config5_both, dict(steps=30, warmup=5, frames=0)),
                             ("config4", bench_config4, dict(steps=4, warmup=1, frames=0, settle=5, repeats=3, bf16=False)),
                             ("config4_bf16", bench_config4, dict(steps=4, warmup=1, frames=0, settle=5, repeats=3, bf16=True)),
                             ("config3", bench_config3, dict(steps=100, warmup=10, frames=0)),
                             # the same encoder at a batch that fills the 256 CUs (is the 5 % of the MFMA peak
                             # at 8 frames the batch or the path?)
                             ("config3_64_frames", bench_config3, dict(steps=30, warmup=5, frames=64)),
                             ("frame_loop", bench_frame_loop, dict(frames=0)),
                             ("train", bench_train, dict(steps=40, warmup=6, frames=0)),
                             ("dropin", bench_dropin, dict(steps=100, warmup=0, frames=0))):
            a2 = copy.copy(args)
            a2.no_extras = True
            for k, v in kw.items():
                setattr(a2, k, v)
            R.args = a2
            try:
                t0 = time.perf_counter()
                others[name] = compact(fn(R))
                others[name]["wall_s"] = round(time.perf_counter() - t0, 1)
            except Exception as e:                               # a side result must not cost the headline line
                others[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        R.args = args
        out["other_configs"] = others
    R.finish(out)


if __name__ == "__main__":
    main()
