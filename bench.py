#!/usr/bin/env python
"""bench.py -- disparity throughput of the PatchMatch-stereo hot path on MI355X.

A "step" = one stereo pair through the whole timed region of the reference's "Total Time"
(main.cc:92-126): plane-cost construction (pyramid + GRD cost of both views and all levels) +
CSPatchMatch::PatchMatch(3 iterations) + PlaneToDisp for both views, with the input images already
resident in HBM.  Workload at N=1: BASELINE.json configs[2] = KITTI-size 1242x375, max_dis=128, GRD,
use_cs=true (5 levels, reg_lambda=0.3), synthetic pair (no dataset in the image).
N>1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank processes its own K pairs
(weak scaling, no data-path collective), value = all pairs' pixels / max-over-ranks time.
`python bench.py --gpus N` without a launcher spawns the N ranks itself.

Pairs are independent units: by default two of them are in flight per GPU (two contexts, two HIP streams; with the folded sweep of round 6
two measure 135.7 ms per pair, three 136.8, four 137.7; four for the small configurations C1 / C2), so the CUs the raster sweep of one pair leaves idle evaluate the other pair's
planes; all K pairs complete inside the timed region.  --in-flight 1 gives the one-pair-at-a-time number.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_TAP = 19  # 3 B guide pixel + 2 x 8 B cost cells per window tap (f64 volume)
OPS_PER_TAP = 14    # algorithmic f64 lane-operations per window tap (an FMA counts once)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# The binding resource is VALU issue (the tap stream never leaves the chip: measured HBM traffic is ~2 % of the algorithmic
# bytes).  Peak = the guide's f64 vector rate: 78.6 TFLOP/s = 3.93e13 lane-operations/s (an FMA = 2 flop = ONE operation)
# = 1024 SIMDs x 16 lanes x 2.4 GHz = one wave64 instruction per SIMD every 4.0 cycles (measured 4.1: profiles/*valu_issue*).
F64_LANE_OPS_PEAK = 78.6e12 / 2.0
VALU_CYCLES_PER_WINSTR = 4.0
N_SIMD = 1024
PMC_FILE = os.path.join(ROOT, "profiles", "refine_pmc.json")  # SQ_INSTS_VALU etc. of k_refine, stamped with the kernel sources' hash


def kernel_source_hash():
    """sha256 over the device sources with comments and white space removed: a committed PMC file is only quoted while it
    describes the CODE of the kernels that are being timed (editing a comment does not invalidate a measurement)"""
    import hashlib
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "crossscalepatchmatch_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            src = open(os.path.join(d, f), encoding="utf-8", errors="replace").read()
            src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)     # block comments
            src = re.sub(r"//[^\n]*", " ", src)                   # line comments (no string literal in csrc/ contains "//")
            h.update(f.encode())
            h.update(" ".join(src.split()).encode())
    return h.hexdigest()[:16]


def cpu_baseline_on(name, device_index=0, crop=None):
    """The oracle (kind "port": the reference itself needs OpenCV/gflags and cannot be built here) in REFERENCE ORDER
    (serial raster sweep, serial window sum, OpenMP over the rows of init/refinement as the reference) on the whole of one
    BASELINE.json config on this box's host cores, next to the GPU on the same pair, seed and schedule -- which also gives
    the north-star parity figure (disparities within 0.5 px of the reference-order CPU result)."""
    from oracle import pyoracle as po
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import synth
    cfg, l, r, _, _ = synth.make_config(name)
    full_w = cfg["w"]
    if crop:  # a centred column band of the pair (both images, same columns): a bounded sample of the same workload
        x0 = (cfg["w"] - crop) // 2
        l, r = np.ascontiguousarray(l[:, x0:x0 + crop]), np.ascontiguousarray(r[:, x0:x0 + crop])
        cfg = dict(cfg, w=crop)
    threads = max(1, min(po.effective_cpus(), cfg["h"]))  # the cores this process may use (cgroup quota), not the visible count
    t0 = time.perf_counter()
    pc = po.PlaneCost(l, r, cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
    pm.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL, threads=threads)
    dt = time.perf_counter() - t0
    g = cs.StereoContext(device_index)
    g.set_images(l, r)
    g.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    g.patchmatch(3, seed=12345, schedule=cs.SCHED_RASTER)  # warm-up
    g.synchronize()
    t1 = time.perf_counter()
    g.set_images(l, r)
    g.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    g.patchmatch(3, seed=12345, schedule=cs.SCHED_RASTER)
    g.synchronize()
    gdt = time.perf_counter() - t1
    taps = g.taps_per_view_pass() * 2 * (pm.evals() // (2 * cfg["w"] * cfg["h"]))
    diff = [np.abs(g.disparity_f64(v) - pm.disp_f64(v)) for v in (0, 1)]
    extra = {}
    if crop:  # what the whole pair would take at this tap rate -- EXTRAPOLATED by the exact in-image tap counts, labelled as such
        lf, rf = synth.make_config(name)[1:3]
        g.set_images(lf, rf)
        g.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
        full_taps = g.taps_per_view_pass() * 2 * (pm.evals() // (2 * cfg["w"] * cfg["h"]))
        extra["whole_pair_extrapolated"] = {"seconds": dt * full_taps / taps, "value": full_w * cfg["h"] / 1e6 / (dt * full_taps / taps), "unit": "Mpix/s",
                                            "note": f"extrapolated, not measured: crop time x (taps of the whole {full_w}x{cfg['h']} pair / taps of the crop) "
                                                    f"= x {full_taps / taps:.3f}"}
    g.close()
    mpix = cfg["w"] * cfg["h"] / 1e6
    return {
        "value": mpix / dt, "unit": "Mpix/s", "cores": threads, "kind": "port",
        "sample": (f"a centred {crop}-column crop of {name} ({full_w}x{cfg['h']})" if crop else f"the whole of {name}") +
                  f": {cfg['w']}x{cfg['h']} max_dis={cfg['max_dis']} GRD "
                  f"{'cross-scale (5 levels, lambda 0.3)' if cfg['scale_num'] else 'single scale'}, 3 iterations, "
                  f"reference order (serial raster sweep, serial window sum), OpenMP over the rows of init/refinement as the reference; "
                  f"{dt:.1f} s, {taps / dt / 1e9:.3f} Gtap/s",
        "seconds": dt,
        "gpu_same_workload": {"value": mpix / gdt, "unit": "Mpix/s", "seconds": gdt, "note": "host buffers in, PCIe included"},
        "gpu_vs_cpu_within0.5px": float(np.mean([np.mean(d <= 0.5) for d in diff])),
        "gpu_vs_cpu_bad0.5": float(np.mean([np.mean(d > 0.5) for d in diff])),
        "gpu_vs_cpu_bad2.0": float(np.mean([np.mean(d > 2.0) for d in diff])),
        "gpu_vs_cpu_max_abs_px": float(max(d.max() for d in diff)),
        **extra,
    }


def cpu_baseline(device_index=0, crop=0, extra_legs=True):
    """The headline configuration, WHOLE: C3 (configs[2], 1242x375, D=128, 5 levels) -- all 2 x 465 750 pixels in the reference's
    order (its raster sweep is serial): three to four minutes on 16 host threads, no extrapolation (BASELINE.md section 4,
    main.cc:92-126).  `crop` > 0 (flag --cpu-crop) times a centred column band of that width instead and labels the extrapolated
    whole pair -- the bounded sample of earlier rounds, kept for quick runs.  Then, as before, the whole of C2 (configs[1], 450x375,
    D=60, 5 levels: the reference's cross-scale path pre_cs_pc.cc:133-188) and the whole of C1 (configs[0], single scale, the
    reference's own CPU-runnable case).  Every leg also runs the GPU on the same inputs: the north-star parity figure (disparities
    within 0.5 px of the reference-order CPU result)."""
    out = cpu_baseline_on("C3", device_index, crop=crop or None)
    out["sample"] += " (BASELINE.json configs[2], the configuration `value` is measured on)"
    if extra_legs:
        out["c2_cross_scale"] = cpu_baseline_on("C2", device_index)
        out["c2_cross_scale"]["sample"] += " (BASELINE.json configs[1])"
        out["c1_single_scale"] = cpu_baseline_on("C1", device_index)
    return out


def real_pair_accuracy(device_index=0):
    """bad-2.0 against REAL ground truth (the synthetic pairs have none worth the name): the Middlebury-2014 Motorcycle pair shipped with
    scikit-image in this image, left view, pixels with known ground truth -- the whole 741x500 pair (max_dis 64) and its half-size
    version (370x250, max_dis 32: the size SURVEY.md 8(c) quotes the unmodified reference on) where that directory exists, and always
    the 200x128 half-size crop committed under tests/data/.  GRD, 5 levels, lambda 0.3, 3 iterations, seed 12345, raw planes and
    after post-processing.  Outside the timed region."""
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import realdata as rd
    out = {"reference_probe_half_size_raw": "0.109-0.111 (SURVEY.md 8(c): the unmodified reference, clock-seeded, 370x250 max_dis 32)",
           "metric": "fraction of left-view pixels with known ground truth whose disparity is off by more than 2 px"}
    g = cs.StereoContext(device_index)
    try:
        for name, loader in (("full_741x500_D64", rd.load_full), ("half_370x250_D32", rd.load_half), ("crop_200x128_D32", rd.load_crop)):
            item = loader()
            if item is None:
                out[name] = None
                continue
            cfg, l, r, gt = item
            g.set_images(l, r)
            g.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
            g.patchmatch(3, seed=12345, schedule=cs.SCHED_RASTER)
            raw = rd.bad_fraction(g.disparity_f64(0), gt, 2.0)
            lo, _ = g.postprocess(cfg["dis_scale"])
            out[name] = {"raw": raw, "post_processed": rd.bad_fraction(lo.astype(np.float64) / cfg["dis_scale"], gt, 2.0)}
    finally:
        g.close()
    return out


def main():
    # ONE JSON line on stdout, nothing else: native libraries write to file descriptor 1 behind Python's back (RCCL prints a
    # five-line version banner there when a communicator is created).  Keep the real stdout for the result line and point fd 1 at
    # stderr for everything else, in this process and in whatever it loads.
    # HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; two pair streams on one queue run one after the other
    # (seen with streams from torch's pool).  The contexts own their streams and did not collide, but leave more queues than streams.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", help="C1 | C2 | C3 | C4 | C5 (crossscalepatchmatch_amd/synth.py); C4 = the batch of C3-shaped pairs "
                                                   "held by rank 0 and dispatched through crossscalepatchmatch_amd.batch.run_batch")
    ap.add_argument("--same-pair", action="store_true", help="time one pair K times instead of K distinct pairs (seeds base + k)")
    ap.add_argument("--in-flight", type=int, default=0, help="stereo pairs in flight per GPU (contexts / HIP streams); 0 (default) = by image size: 2 for KITTI-size pairs and larger "
                                                            "(measured 135.7 ms per C3 pair against 136.8 with 3), 4 below 400 000 pixels (C1 / C2: 19.1 / 57.6 ms against "
                                                            "20.7 / 60.1 with 2 -- small kernels leave more of the GPU idle).  With two or more the sweep runs "
                                                            "four-wave workgroups (CSPM_OPT_SWEEP_FOLD: they leave the other pairs' kernels their registers)")
    ap.add_argument("--schedule", default="raster", choices=["raster", "redblack"])
    ap.add_argument("--rb-rounds", type=int, default=1)
    ap.add_argument("--no-early-exit", action="store_true")
    ap.add_argument("--volumes", action="store_true", help="materialise f64 cost volumes (reference data flow) instead of fused cells")
    ap.add_argument("--sweep-pairs", action="store_true", help="raster sweep on paired-cell volumes (CSPM_OPT_SWEEP_PAIRS) instead of the fused cells")
    ap.add_argument("--raster-launches", action="store_true", help="raster sweep as one launch per anti-diagonal instead of the persistent kernel")
    ap.add_argument("--cc", default="GRD", choices=["GRD", "CEN", "IMG"],
                    help="cost function (BASELINE.json's metric is GRD; CEN = census; IMG = the volume-free GrdPC / CSPC plane costs, for comparison)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, default=0, help="CPU baseline on a centred crop of this many columns of C3 (with the whole pair extrapolated and "
                                                            "labelled so) instead of the whole pair, which takes three to four minutes of host time")
    ap.add_argument("--cpu-c3-only", action="store_true", help="CPU baseline: skip the C2 and C1 legs")
    ap.add_argument("--no-real-pair", action="store_true", help="skip the accuracy figure on the real Middlebury pair (real_pair_bad2)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket launches with hipEvents")
    args = ap.parse_args()

    import torch
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import synth

    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: libcspm_hip has no CPU fallback")
    backend = os.environ.get("CSPM_BENCH_BACKEND", "nccl")  # "gloo": only to exercise the N>1 control flow on a 1-GPU box
    if args.gpus < 1 or args.in_flight < 0:
        raise SystemExit("--gpus must be >= 1 and --in-flight >= 0")
    if args.in_flight == 0:
        args.in_flight = 2 if synth.CONFIGS[args.config]["w"] * synth.CONFIGS[args.config]["h"] >= 400000 else 4
    if backend == "nccl" and ndev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes only {ndev} GPU(s); one rank per GPU is required "
                         f"(set CSPM_BENCH_BACKEND=gloo to exercise the N>1 control flow with ranks sharing a GPU)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher the driver would have used -- one rank per GPU
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, stdout=result_out))  # the ranks inherit the real stdout; rank 0 writes the line
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: they must agree")
    dist = None
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    force_dist = os.environ.get("CSPM_BENCH_FORCE_DIST", "0") == "1"  # a process group (and C4's collectives) even at world size 1: RCCL on a one-GPU box
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    # K distinct pairs per rank (seeds base + i), all resident in HBM before the timed region starts.
    # C4 (BASELINE.json configs[3]): rank 0 holds the WHOLE batch (K pairs per rank x world) and batch.run_batch dispatches it
    # (parameter broadcast, chunked scatter, per-rank contexts in flight, gather of the maps) inside the timed region.
    batch_mode = args.config == "C4"
    npairs = 1 if args.same_pair else args.steps
    first = rank * npairs
    if batch_mode and rank == 0 and not args.same_pair:
        first, npairs = 0, args.steps * world
    host_pairs = [synth.make_config(args.config, index=first + k) for k in range(npairs)]
    cfg, _, _, gl, gr = host_pairs[(args.steps - 1) % npairs]  # ground truth of the LAST timed pair (checked below)
    w, h = cfg["w"], cfg["h"]
    use_pp = bool(cfg.get("use_pp"))
    d_pairs = [(torch.from_numpy(p[1]).to(dev), torch.from_numpy(p[2]).to(dev)) for p in host_pairs]
    torch.cuda.synchronize()
    from crossscalepatchmatch_amd import capi
    nfl = max(1, min(args.in_flight, args.steps))
    pair_fn = None
    if batch_mode:
        from crossscalepatchmatch_amd import batch
        pair_fn = batch.HipPairFn(dev_index, in_flight=nfl)
        ctxs = pair_fn.ctxs
        batch_dev = dev if backend == "nccl" else torch.device("cpu")  # the collectives' side: RCCL moves device tensors, gloo host tensors
        batch_pairs = torch.stack([torch.stack(p) for p in d_pairs]).to(batch_dev) if rank == 0 else None  # [n, 2, h, w, 3], resident before the timed region
        batch_params = dict(w=w, h=h, max_dis=cfg["max_dis"], dis_scale=cfg["dis_scale"], scale_num=cfg["scale_num"], reg_lambda=cfg["reg_lambda"],
                            iters=3, seed=12345, schedule=0 if args.schedule == "raster" else 1, use_pp=int(use_pp), cc=batch.CC_CODES[args.cc])

        class _ViaHost:  # gloo control plane (CSPM_BENCH_BACKEND=gloo, ranks sharing a GPU): blocks arrive as host tensors
            def __call__(self, l, r, p):
                dl, dr = pair_fn(l.to(dev).contiguous(), r.to(dev).contiguous(), p)
                pair_fn.order_after_pairs()
                return dl.cpu(), dr.cpu()

            def finalize(self):
                pair_fn.finalize()
    else:
        ctxs = [cs.StereoContext(dev_index) for _ in range(nfl)]  # each context owns a HIP stream
        if nfl >= 2 and not os.environ.get("CSPM_SWEEP_FOLD"):
            for ctx in ctxs:  # the pairs share the GPU: four-wave sweep workgroups leave room for two of another pair's refinement workgroups (include/cspm.h)
                ctx.set_option(capi.OPT_SWEEP_FOLD, 1)
    d_out = [[torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(2)] for _ in range(nfl)]
    for ctx in ctxs:
        ctx.set_option(capi.OPT_RASTER_LAUNCHES, int(args.raster_launches))
    sched = cs.SCHED_RASTER if args.schedule == "raster" else cs.SCHED_REDBLACK
    pm_kw = dict(seed=12345, schedule=sched, rb_rounds=args.rb_rounds, rb_neighbours=4, early_exit=0 if args.no_early_exit else 1)

    def step(k, pair=None):
        ctx, out = ctxs[k % nfl], d_out[k % nfl]  # everything below is enqueued on the context's stream, nothing waits
        d_l, d_r = d_pairs[(k if pair is None else pair) % npairs]
        ctx.set_images_device(d_l.data_ptr(), d_r.data_ptr(), w, h, w * 3)
        if args.cc == "IMG":
            ctx.build_cost_img(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
        else:
            if args.cc == "GRD":
                ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"], volumes=args.volumes, sweep_pairs=True if args.sweep_pairs else None)
            else:
                ctx.build_cost_cen(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"], volumes=args.volumes)
        ctx.patchmatch(3, **pm_kw)
        if use_pp:  # PatchMatch(iter_num, plane_cost, use_pp = true): post-processing is inside the reference's timed region (main.cc:92-126)
            ctx.postprocess_device(cfg["dis_scale"], out[0].data_ptr(), out[1].data_ptr())
        else:
            for v in (0, 1):
                ctx.disparity_u8_device(v, cfg["dis_scale"], out[v].data_ptr())

    def sync_all():
        for ctx in ctxs:
            ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def run_batch_once(count_per_rank):
        sub = batch_pairs[: count_per_rank * world] if rank == 0 else None
        fn = pair_fn if batch_dev.type == "cuda" else _ViaHost()
        return batch.run_batch(sub, batch_params if rank == 0 else None, fn, device=str(batch_dev), dist=dist, force_collectives=force_dist)

    if batch_mode:
        if args.warmup:
            run_batch_once(min(args.warmup, args.steps))
    else:
        for k in range(args.warmup):
            step(k, pair=npairs - 1 - k % npairs)  # warm-up on pairs from the END of the list: the timed region starts on inputs not seen yet
    for ctx in ctxs:
        ctx.synchronize()
        ctx.enable_timing(not args.no_kernel_timing)
        ctx.reset_timing()
    fallbacks_before = (sum(ctx.get_option(capi.OPT_SWEEP_FALLBACKS) for ctx in ctxs), sum(ctx.get_option(capi.OPT_VOLUME_FALLBACKS) for ctx in ctxs))
    sync_all()
    t0 = time.perf_counter()
    if batch_mode:
        batch_maps = run_batch_once(args.steps)  # dispatch + compute + gather: the whole of configs[3]
    else:
        for k in range(args.steps):
            step(k)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # what happened inside the timed region, beyond its duration: a raster sweep that timed out and was repeated with per-diagonal
    # launches, or an optional volume that could not be allocated, means `value` timed a different code path than the one it names
    sweep_fallbacks = sum(ctx.get_option(capi.OPT_SWEEP_FALLBACKS) for ctx in ctxs) - fallbacks_before[0]
    volume_fallbacks = sum(ctx.get_option(capi.OPT_VOLUME_FALLBACKS) for ctx in ctxs) - fallbacks_before[1]
    table_volumes_active = [int(ctx.get_option(capi.OPT_TABLE_VOLUMES_ACTIVE)) for ctx in ctxs]
    if dist is not None:  # every rank's contexts
        fb = torch.tensor([sweep_fallbacks, volume_fallbacks], dtype=torch.float64, device=red_dev)
        dist.all_reduce(fb, op=dist.ReduceOp.SUM)
        sweep_fallbacks, volume_fallbacks = int(fb[0].item()), int(fb[1].item())
    timing = {}
    for ctx in ctxs:
        for k, v in ctx.timing().items():
            acc = timing.setdefault(k, {"launches": 0, "ms": 0.0, "evals": 0})
            for f in acc:
                acc[f] += v[f]
        ctx.enable_timing(False)
    # Kernel durations for the roofline: with several pairs in flight the kernels of different pairs share the GPU and a
    # launch's hipEvent bracket measures the mixture, so the dominant kernel is timed once more with ONE pair on the GPU
    # (same inputs, same code path; outside the timed region, which `value` comes from).
    solo = timing
    if nfl > 1:
        fold_keep = ctxs[0].get_option(capi.OPT_SWEEP_FOLD)
        ctxs[0].set_option(capi.OPT_SWEEP_FOLD, 0)  # a pair that is alone on the GPU runs the sweep with the library's default (one wave per level)
        ctxs[0].enable_timing(not args.no_kernel_timing)
        ctxs[0].reset_timing()
        step(0, pair=args.steps - 1)
        ctxs[0].synchronize()
        solo = ctxs[0].timing()
        ctxs[0].enable_timing(False)
        ctxs[0].set_option(capi.OPT_SWEEP_FOLD, fold_keep)

    bad_run = False
    if rank == 0:
        ctx = ctxs[0]
        mpix = w * h * args.steps * world / dt / 1e6
        alg_taps = 2 * ctx.taps_per_view_pass()              # one evaluation of every pixel of both views (in-image window taps)
        exe_taps = 2 * ctx.row_engine_taps_per_view_pass()   # lane-taps the row engine executes for it
        out = {
            "metric": "Mpix/s disparity (%s, use_cs=%s)" % (args.cc, "true" if cfg["scale_num"] else "false"), "value": mpix, "unit": "Mpix/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {w}x{h} max_dis={cfg['max_dis']} GRD scale_num={cfg['scale_num']} "
                                   f"reg_lambda={cfg['reg_lambda']} wnd=35 iters=3 (BASELINE.json configs[2] when C3)",
                       "cc_name": args.cc, "cost_source": "volumes" if args.volumes else "fused", "schedule": args.schedule,
                       "sweep_cells": "paired-cell volumes" if ctxs[0].get_option(capi.OPT_SWEEP_PAIRS_ACTIVE) else "as cost_source",
                       "raster_sweep": "per-diagonal launches" if args.raster_launches else "persistent", "rb_rounds": args.rb_rounds,
                       "sweep_workgroups_per_cu": ctxs[0].get_option(capi.OPT_SWEEP_WG) or 2,
                       "sweep_workgroup_waves": ("levels - 1 (coarsest level folded, CSPM_OPT_SWEEP_FOLD)" if ctxs[0].get_option(capi.OPT_SWEEP_FOLD) else "one per level"),
                       "early_exit": not args.no_early_exit, "pairs_per_gpu": args.steps, "distinct_pairs_per_gpu": npairs, "pairs_in_flight_per_gpu": nfl, "use_pp": use_pp,
                       "parallelism": f"{world} rank(s), one per GPU, {nfl} independent pair stream(s) each",
                       "dispatch": ("batch.run_batch: rank 0 holds the %d pairs; broadcast of the parameters, chunked scatter of the pair blocks, "
                                    "gather of the 8-bit maps -- all inside the timed region" % (args.steps * world)) if batch_mode else
                                   "every rank generates its own pairs (no collective in the timed region)"},
        }
        ref = solo["refine"]
        if ref["launches"]:
            # the dominant kernel: k_refine = all halving steps of one PlaneRefinement iteration (cs_patchmatch.cc:292-345)
            avg_s = ref["ms"] / ref["launches"] / 1e3
            steps_per_launch = ref["evals"] / ref["launches"] / (2 * w * h)
            taps_launch = alg_taps * steps_per_launch
            alg_bytes = taps_launch * BYTES_PER_TAP
            alg_ops = taps_launch * OPS_PER_TAP
            roof = {
                "kernel": "k_refine (row engine; one launch = the %d halving steps of one PlaneRefinement iteration)" % round(steps_per_launch),
                "avg_launch_ms": avg_s * 1e3, "launches": ref["launches"],
                "measured": "hipEvents on the context stream; " + ("one extra pair alone on the GPU after the timed region (kernels of overlapping "
                                                                     "pairs are not separable)" if nfl > 1 else "the timed region"),
                # the roofline of SURVEY.md 8(d): ALGORITHMIC operations (14 per in-image window tap, as the reference performs them)
                # per launch / the launch duration measured here, against the chip's f64 vector rate
                "bound": "valu_issue", "achieved": alg_ops / avg_s / 1e12, "peak": F64_LANE_OPS_PEAK / 1e12, "unit": "T f64 lane-op/s",
                "frac": alg_ops / avg_s / F64_LANE_OPS_PEAK,
                "algorithmic_ops_per_tap": OPS_PER_TAP,
                "algorithmic_taps_per_launch": taps_launch, "executed_lane_taps_per_launch": exe_taps * steps_per_launch,
                "executed_vs_algorithmic_taps": exe_taps / alg_taps, "taps_per_s": taps_launch / avg_s,
                # the HBM yardstick next to it: > 1 x peak only says that the tap stream is served on chip
                "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBs": alg_bytes / avg_s / 1e9,
                "algorithmic_hbm_frac": alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
                "traffic": None,
                "note": "bound = VALU issue: the tap stream never leaves the chip (see traffic vs algorithmic_bytes_per_launch), HBM at %.1f x its "
                        "peak by the algorithmic-byte yardstick.  frac = algorithmic f64 operations (14 per tap x in-image taps of one launch) / "
                        "launch time / 3.93e13 lane-op/s (78.6 TFLOP/s f64 vector, an FMA counted once).  frac is a yardstick of the REFERENCE's "
                        "operation count, not a distance to an attainable 1.0: the kernel's table rows execute fewer VALU instructions per tap (7.9-9.9) "
                        "than the 14 operations they are credited with, its general rows more (21-25); see valu_winstr_per_64_algorithmic_taps for "
                        "what was actually issued." % (alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS),
            }
            # instruction-level view of the same launch from the committed rocprofv3 --pmc passes -- quoted only while the file
            # describes the kernels being timed (source hash) and the headline workload
            pmc = json.load(open(PMC_FILE)) if os.path.exists(PMC_FILE) else None
            if pmc and pmc.get("kernel_source_hash") == kernel_source_hash() and args.config in ("C3", "C4") and args.cc == "GRD" and not args.volumes:
                winstr = pmc["valu_winstr_per_launch"]
                peak = N_SIMD * pmc["shader_clock_ghz"] * 1e9 / VALU_CYCLES_PER_WINSTR
                roof.update({
                    "traffic": pmc["hbm_bytes_per_launch"], "hbm_frac_of_peak": pmc["hbm_bytes_per_launch"] / avg_s / 1e9 / HBM_PEAK_GBS,
                    "valu_busy_frac": winstr / avg_s / peak, "valu_winstr_per_launch_pmc": winstr,
                    "valu_winstr_per_64_algorithmic_taps": winstr / (taps_launch / 64),
                    "lds_busy_frac_pmc": pmc["lds_busy_frac"], "pmc_file": os.path.relpath(PMC_FILE, ROOT),
                    "pmc_note": "valu_busy_frac = SQ_INSTS_VALU per launch (rocprofv3 --pmc of this command, committed file, same kernel-source "
                                "hash) x 4.0 cycles / 1024 SIMDs / the shader cycles of the launch timed here",
                })
            else:
                roof["pmc_note"] = ("no committed PMC file for these kernels (source hash mismatch or another workload): instruction-level "
                                    "figures omitted rather than quoted stale")
            # every plane-evaluating kernel against the same yardstick (one pair alone on the GPU), and the whole pair as timed
            pass_taps = alg_taps / 2.0  # one evaluation of every pixel of ONE view
            per_view_evals = {
                # InitRandomPlane: every pixel of both views once (cs_patchmatch.cc:115-148)
                "init": ("k_init", 2.0 * w * h),
                # SpatialPropagation: two evaluations per pixel as the reference performs them -- one in the first sweep row and
                # column, none at the first pixel (cs_patchmatch.cc:178-213) -- both views per launch.  The kernel itself skips
                # evaluations that cannot change the outcome (bitwise equal candidates): the yardstick does not
                "spatial": ("k_spatial_sweep" if not args.raster_launches else "k_spatial_diag (all launches of a sweep)", 2.0 * (2.0 * w * h - w - h)),
                # ViewPropagation: one candidate per source pixel, one target view per launch (cs_patchmatch.cc:229-277); candidates are
                # evaluated at their target column -- the tap count of a uniform pass over the view is used
                "view": ("k_view_eval", 1.0 * w * h),
                "refine": ("k_refine", 2.0 * w * h * steps_per_launch),
            }
            kern = {}
            sweeps_per_timed = 1 if not args.raster_launches else (w + h - 2)
            for key, (kname, evals) in per_view_evals.items():
                t = solo.get(key)
                if not t or not t["launches"]:
                    continue
                launches = t["launches"] / (sweeps_per_timed if key == "spatial" else 1)
                k_s = t["ms"] / launches / 1e3
                k_ops = evals / (w * h) * pass_taps * OPS_PER_TAP
                kern[key] = {"kernel": kname, "avg_launch_ms": k_s * 1e3, "launches_per_pair": launches, "algorithmic_evaluations_per_launch": evals,
                             "algorithmic_ops_per_launch": k_ops, "frac": k_ops / k_s / F64_LANE_OPS_PEAK}
            roof["kernels"] = kern
            evals_per_pixel_view = sum(per_view_evals[k][1] / (2.0 * w * h) * (solo[k]["launches"] / (sweeps_per_timed if k == "spatial" else 1))
                                       for k in per_view_evals if solo.get(k) and solo[k]["launches"])
            pair_ops = evals_per_pixel_view * alg_taps * OPS_PER_TAP
            roof["pair_frac"] = pair_ops / (dt / args.steps) / F64_LANE_OPS_PEAK
            roof["pair_note"] = ("pair_frac = algorithmic operations of a whole pair (%.1f evaluations per pixel and view x in-image taps x 14) / ms_per_step "
                                 "(the timed region, %d pair(s) in flight) / 3.93e13" % (evals_per_pixel_view, nfl))
            spmc_file = os.path.join(ROOT, "profiles", "sweep_pmc.json")
            spmc = json.load(open(spmc_file)) if os.path.exists(spmc_file) else None
            if spmc and spmc.get("kernel_source_hash") == kernel_source_hash() and args.config in ("C3", "C4") and args.cc == "GRD" and not args.volumes and "spatial" in kern:
                kern["spatial"].update({"pmc_file": os.path.relpath(spmc_file, ROOT), "td_busy_frac_pmc": spmc.get("td_busy_frac"), "ta_busy_frac_pmc": spmc.get("ta_busy_frac"),
                                        "valu_winstr_per_launch_pmc": spmc.get("valu_winstr_per_launch"), "vmem_rd_instr_per_pixel_pmc": spmc.get("vmem_rd_instr_per_launch", 0) / (2.0 * w * h),
                                        "traffic": spmc.get("hbm_bytes_per_launch"), "bound": "L1 return path (TD) + dependency chain of the anti-diagonals"})
            out["roofline"] = roof
        # hipEvent brackets of the launches, summed per kernel class and divided by the pairs.  With several pairs in flight the brackets
        # of different pairs OVERLAP in time (a launch's bracket also contains whatever the other streams ran meanwhile), so these do not
        # add up to ms_per_step and are not kernel durations: the key says so.  Durations: roofline.kernels / kernel_ms_one_pair_alone.
        out["kernel_bracket_ms_per_pair_overlapping" if nfl > 1 else "kernel_ms_per_step"] = {k: v["ms"] / args.steps for k, v in timing.items()}
        out["kernel_launches_per_step"] = {k: v["launches"] / args.steps for k, v in timing.items()}
        out["sweep_fallbacks"] = int(sweep_fallbacks)
        out["volume_fallbacks"] = int(volume_fallbacks)
        out["table_volumes_active"] = bool(all(table_volumes_active)) if args.cc == "GRD" and not args.volumes else None
        out["timed_region_note"] = ("sweep_fallbacks / volume_fallbacks: raster sweeps repeated with per-diagonal launches after a hand-over timeout, and "
                                    "optional volumes given up, INSIDE the timed region (all contexts of all ranks); both must be 0 or the run exits non-zero")
        if nfl > 1:
            out["kernel_ms_one_pair_alone"] = {k: v["ms"] for k, v in solo.items()}
        if world == 1 and not args.no_cpu_baseline and args.cc == "GRD":
            try:
                out["cpu_baseline"] = cpu_baseline(dev_index, crop=args.cpu_crop, extra_legs=not args.cpu_c3_only)
            except Exception as e:  # the timed result must not be lost to a problem of the reporting leg: say so in its place
                out["cpu_baseline"] = {"error": repr(e)}
                print("bench.py: the cpu_baseline leg failed: %r" % (e,), file=sys.stderr)
        # sanity of the LAST pair that was timed (not part of the timed region)
        if batch_mode:  # rank 0's last own pair: index steps-1 of the batch; its 8-bit map came back through the gather
            gl = host_pairs[min(args.steps, npairs) - 1][3]
            out["bad2_vs_gt_left"] = synth.bad_fraction(batch_maps[min(args.steps, npairs) - 1, 0].cpu().numpy().astype(np.float64) / cfg["dis_scale"], gl, 2.0)
        else:
            last = ctxs[(args.steps - 1) % nfl]
            out["bad2_vs_gt_left"] = synth.bad_fraction(last.disparity_f64(0), gl, 2.0)
        out["distinct_pairs"] = npairs
        if world == 1 and not args.no_real_pair:
            try:
                out["real_pair_bad2"] = real_pair_accuracy(dev_index)
            except Exception as e:
                out["real_pair_bad2"] = {"error": repr(e)}
                print("bench.py: the real-pair leg failed: %r" % (e,), file=sys.stderr)
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
        if sweep_fallbacks or volume_fallbacks:
            bad_run = True
    if pair_fn is not None:
        pair_fn.close()
    else:
        for ctx in ctxs:
            ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if bad_run:
        raise SystemExit("bench.py: a fallback happened inside the timed region (sweep_fallbacks / volume_fallbacks in the JSON line): "
                         "`value` timed another code path than the one the line names")


if __name__ == "__main__":
    main()
