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

Pairs are independent units: by default three of them are in flight per GPU (three contexts, three HIP streams), so the
CUs the raster sweep of one pair leaves idle (short anti-diagonals) evaluate the other pair's planes; all K pairs
complete inside the timed region.  --in-flight 1 gives the one-pair-at-a-time number.

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
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# VALU issue: one wave64 instruction occupies its SIMD for 4.1 shader cycles, f64 and 32-bit alike (measured:
# tools/ubench/valu_issue.hip, profiles/r02_valu_issue.txt; = the guide's 78.6 TFLOP/s f64 vector peak, 2 flop per lane
# per FMA); 256 CUs x 4 SIMDs.  The shader clock under this kernel is taken from the same PMC run as the instruction count.
VALU_CYCLES_PER_WINSTR = 4.1
N_SIMD = 1024
PMC_FILE = os.path.join(ROOT, "profiles", "r02_refine_pmc.json")


def cpu_baseline(device_index=0):
    """The oracle (kind "port": the reference itself needs OpenCV/gflags and cannot be built here) in REFERENCE ORDER
    (serial raster sweep, serial window sum, OpenMP over the rows of init/refinement as the reference) on the whole of
    BASELINE.json configs[0] (C1: 450x375, max_dis=60, GRD, single scale) on this box's host cores, next to the GPU on the
    same pair, seed and schedule -- which also gives the north-star parity figure (disparities within 0.5 px)."""
    from oracle import pyoracle as po
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import synth
    cfg, l, r, _, _ = synth.make_config("C1")
    threads = max(1, min(os.cpu_count() or 1, cfg["h"]))
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
    g.close()
    mpix = cfg["w"] * cfg["h"] / 1e6
    return {
        "value": mpix / dt, "unit": "Mpix/s", "cores": threads, "kind": "port",
        "sample": f"the whole of C1 (BASELINE.json configs[0]): {cfg['w']}x{cfg['h']} max_dis={cfg['max_dis']} GRD single scale, 3 iterations, "
                  f"reference order (serial raster sweep, serial window sum), OpenMP over the rows of init/refinement as the reference; "
                  f"{dt:.1f} s, {taps / dt / 1e9:.3f} Gtap/s",
        "seconds": dt,
        "gpu_same_workload": {"value": mpix / gdt, "unit": "Mpix/s", "seconds": gdt, "note": "host buffers in, PCIe included"},
        "gpu_vs_cpu_bad0.5": float(np.mean([np.mean(d > 0.5) for d in diff])),
        "gpu_vs_cpu_bad2.0": float(np.mean([np.mean(d > 2.0) for d in diff])),
        "gpu_vs_cpu_max_abs_px": float(max(d.max() for d in diff)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", help="C1 | C2 | C3 | C5 (crossscalepatchmatch_amd/synth.py)")
    ap.add_argument("--in-flight", type=int, default=3, help="stereo pairs in flight per GPU (contexts / HIP streams)")
    ap.add_argument("--schedule", default="raster", choices=["raster", "redblack"])
    ap.add_argument("--rb-rounds", type=int, default=1)
    ap.add_argument("--no-early-exit", action="store_true")
    ap.add_argument("--volumes", action="store_true", help="materialise f64 cost volumes (reference data flow) instead of fused cells")
    ap.add_argument("--raster-launches", action="store_true", help="raster sweep as one launch per anti-diagonal instead of the persistent kernel")
    ap.add_argument("--cc", default="GRD", choices=["GRD", "CEN", "IMG"],
                    help="cost function (BASELINE.json's metric is GRD; CEN = census; IMG = the volume-free GrdPC / CSPC plane costs, for comparison)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket launches with hipEvents")
    args = ap.parse_args()

    import torch
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import synth

    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: libcspm_hip has no CPU fallback")
    backend = os.environ.get("CSPM_BENCH_BACKEND", "nccl")  # "gloo": only to exercise the N>1 control flow on a 1-GPU box
    if args.gpus < 1 or args.in_flight < 1:
        raise SystemExit("--gpus and --in-flight must be >= 1")
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
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: they must agree")
    dist = None
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    cfg, l, r, gl, gr = synth.make_config(args.config, index=rank)
    w, h = cfg["w"], cfg["h"]
    d_l = torch.from_numpy(l).to(dev)
    d_r = torch.from_numpy(r).to(dev)
    torch.cuda.synchronize()
    from crossscalepatchmatch_amd import capi
    nfl = max(1, min(args.in_flight, args.steps))
    ctxs = [cs.StereoContext(dev_index) for _ in range(nfl)]  # each context owns a HIP stream
    d_out = [[torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(2)] for _ in range(nfl)]
    for ctx in ctxs:
        ctx.set_option(capi.OPT_RASTER_LAUNCHES, int(args.raster_launches))
    sched = cs.SCHED_RASTER if args.schedule == "raster" else cs.SCHED_REDBLACK
    pm_kw = dict(seed=12345, schedule=sched, rb_rounds=args.rb_rounds, rb_neighbours=4, early_exit=0 if args.no_early_exit else 1)

    def step(k):
        ctx, out = ctxs[k % nfl], d_out[k % nfl]  # everything below is enqueued on the context's stream, nothing waits
        ctx.set_images_device(d_l.data_ptr(), d_r.data_ptr(), w, h, w * 3)
        if args.cc == "IMG":
            ctx.build_cost_img(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
        else:
            (ctx.build_cost_grd if args.cc == "GRD" else ctx.build_cost_cen)(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"], volumes=args.volumes)
        ctx.patchmatch(3, **pm_kw)
        for v in (0, 1):
            ctx.disparity_u8_device(v, cfg["dis_scale"], out[v].data_ptr())

    def sync_all():
        for ctx in ctxs:
            ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    for ctx in ctxs:
        ctx.synchronize()
        ctx.enable_timing(not args.no_kernel_timing)
        ctx.reset_timing()
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
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
        ctxs[0].enable_timing(not args.no_kernel_timing)
        ctxs[0].reset_timing()
        step(0)
        ctxs[0].synchronize()
        solo = ctxs[0].timing()
        ctxs[0].enable_timing(False)

    if rank == 0:
        ctx = ctxs[0]
        mpix = w * h * args.steps * world / dt / 1e6
        alg_taps = 2 * ctx.taps_per_view_pass()              # one evaluation of every pixel of both views (in-image window taps)
        exe_taps = 2 * ctx.row_engine_taps_per_view_pass()   # lane-taps the row engine executes for it
        out = {
            "metric": "Mpix/s disparity (%s, use_cs=true)" % args.cc, "value": mpix, "unit": "Mpix/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {w}x{h} max_dis={cfg['max_dis']} GRD scale_num={cfg['scale_num']} "
                                   f"reg_lambda={cfg['reg_lambda']} wnd=35 iters=3 (BASELINE.json configs[2] when C3)",
                       "cc_name": args.cc, "cost_source": "volumes" if args.volumes else "fused", "schedule": args.schedule,
                       "raster_sweep": "per-diagonal launches" if args.raster_launches else "persistent", "rb_rounds": args.rb_rounds,
                       "early_exit": not args.no_early_exit, "pairs_per_gpu": args.steps, "pairs_in_flight_per_gpu": nfl,
                       "parallelism": f"{world} rank(s), one per GPU, {nfl} independent pair stream(s) each"},
        }
        ref = solo["refine"]
        if ref["launches"]:
            # the dominant kernel: k_refine = all halving steps of one PlaneRefinement iteration (cs_patchmatch.cc:292-345)
            avg_s = ref["ms"] / ref["launches"] / 1e3
            steps_per_launch = ref["evals"] / ref["launches"] / (2 * w * h)
            alg_bytes = alg_taps * steps_per_launch * BYTES_PER_TAP
            roof = {
                "kernel": "k_refine (row engine; one launch = the %d halving steps of one PlaneRefinement iteration)" % round(steps_per_launch),
                "avg_launch_ms": avg_s * 1e3, "launches": ref["launches"],
                "measured": "hipEvents on the context stream; " + ("one extra pair alone on the GPU after the timed region (kernels of overlapping "
                                                                     "pairs are not separable)" if nfl > 1 else "the timed region"),
                "algorithmic_taps_per_launch": alg_taps * steps_per_launch, "executed_lane_taps_per_launch": exe_taps * steps_per_launch,
                "executed_vs_algorithmic_taps": exe_taps / alg_taps,
                "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBs": alg_bytes / avg_s / 1e9,
                "taps_per_s": alg_taps * steps_per_launch / avg_s,
            }
            pmc = json.load(open(PMC_FILE)) if os.path.exists(PMC_FILE) else None
            if pmc and args.config == "C3" and args.cc == "GRD" and not args.volumes:
                winstr = pmc["valu_winstr_per_launch"]
                peak = N_SIMD * pmc["shader_clock_ghz"] * 1e9 / VALU_CYCLES_PER_WINSTR
                roof.update({
                    "bound": "valu_issue", "achieved": winstr / avg_s / 1e9, "peak": peak / 1e9, "unit": "G wave-instr/s",
                    "frac": winstr / avg_s / peak,
                    "traffic": pmc["hbm_bytes_per_launch"], "hbm_frac_of_peak": pmc["hbm_bytes_per_launch"] / avg_s / 1e9 / HBM_PEAK_GBS,
                    "lds_busy_frac_pmc": pmc["lds_busy_frac"], "valu_winstr_per_64_algorithmic_taps": winstr / (alg_taps * steps_per_launch / 64),
                    "note": "The tap stream is served on chip (measured HBM traffic is %.1f %% of the algorithmic bytes), so the "
                            "bound is not HBM.  bound = VALU issue: achieved = VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU of "
                            "the same command, profiles/r02_refine_pmc.json) / the launch time measured here; peak = 1024 SIMDs x shader "
                            "clock / 4.1 cycles per wave-instruction (tools/ubench/valu_issue.hip).  The LDS (strips + tables) is the "
                            "second resource, lds_busy_frac_pmc." % (100.0 * pmc["hbm_bytes_per_launch"] / alg_bytes),
                })
            else:
                roof.update({"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G wave-instr/s", "frac": None, "traffic": None,
                             "note": "instruction counts are committed for the headline workload only (C3, GRD, fused)"})
            out["roofline"] = roof
        out["kernel_ms_per_step"] = {k: v["ms"] / args.steps for k, v in timing.items()}  # with pairs in flight: overlapping brackets
        out["kernel_launches_per_step"] = {k: v["launches"] / args.steps for k, v in timing.items()}
        if nfl > 1:
            out["kernel_ms_one_pair_alone"] = {k: v["ms"] for k, v in solo.items()}
        if world == 1 and not args.no_cpu_baseline and args.cc == "GRD":
            out["cpu_baseline"] = cpu_baseline(dev_index)
        # sanity of the result that was timed (not part of the timed region)
        dl = ctx.disparity_f64(0)
        out["bad2_vs_gt_left"] = synth.bad_fraction(dl, gl, 2.0)
        print(json.dumps(out))
    for ctx in ctxs:
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
