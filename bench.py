#!/usr/bin/env python
"""bench.py -- disparity throughput of the PatchMatch-stereo hot path on MI355X.

A "step" = one stereo pair through the whole timed region of the reference's "Total Time"
(main.cc:92-126): plane-cost construction (pyramid + GRD cost volumes of both views and all levels) +
CSPatchMatch::PatchMatch(3 iterations) + PlaneToDisp for both views, with the input images already
resident in HBM.  Workload at N=1: BASELINE.json configs[2] = KITTI-size 1242x375, max_dis=128, GRD,
use_cs=true (5 levels, reg_lambda=0.3), synthetic pair (no dataset in the image).
N>1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank processes its own K pairs
(weak scaling, no data-path collective), value = all pairs' pixels / max-over-ranks time.

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


def cpu_baseline(cfg, l, r, device_index=0):
    """The oracle (kind "port": the reference itself needs OpenCV/gflags and cannot be built here) in
    reference order on a bounded centred crop of the same pair, on this box's host cores."""
    from oracle import pyoracle as po
    cw, ch = min(cfg["w"], 288), min(cfg["h"], 160)
    x0, y0 = (cfg["w"] - cw) // 2, (cfg["h"] - ch) // 2
    lc = np.ascontiguousarray(l[y0:y0 + ch, x0:x0 + cw])
    rc = np.ascontiguousarray(r[y0:y0 + ch, x0:x0 + cw])
    threads = max(1, min(os.cpu_count() or 1, ch))
    t0 = time.perf_counter()
    pc = po.PlaneCost(lc, rc, cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    pm = po.PatchMatch(lc, rc, cfg["max_dis"], cfg["dis_scale"])
    pm.run(3, pc, False, seed=12345, schedule=po.SCHED_RASTER, sum_order=po.SUM_SERIAL, threads=threads)
    dt = time.perf_counter() - t0
    taps = sum(pc.taps(x, y) for y in range(ch) for x in range(cw)) * (pm.evals() // (cw * ch))
    # the north-star parity figure on the same crop: HIP path vs this CPU run (identical inputs, seeds, schedule)
    import crossscalepatchmatch_amd as cs
    g = cs.StereoContext(device_index)
    g.set_images(lc, rc)
    g.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    g.patchmatch(3, seed=12345, schedule=cs.SCHED_RASTER)
    diff = [np.abs(g.disparity_f64(v) - pm.disp_f64(v)) for v in (0, 1)]
    g.close()
    return {
        "gpu_vs_cpu_bad0.5": float(np.mean([np.mean(d > 0.5) for d in diff])),
        "gpu_vs_cpu_bad2.0": float(np.mean([np.mean(d > 2.0) for d in diff])),
        "gpu_vs_cpu_max_abs_px": float(max(d.max() for d in diff)),
        "value": cw * ch / dt / 1e6, "unit": "Mpix/s", "cores": threads, "kind": "port",
        "sample": f"centred {cw}x{ch} crop of the same pair, max_dis={cfg['max_dis']}, {cfg['scale_num']} levels, "
                  f"reference order (raster sweep, serial sum), OpenMP over rows of init/refinement as the reference; "
                  f"{dt:.1f} s, {taps / dt / 1e9:.3f} Gtap/s (crop windows are border-clipped: optimistic for the CPU)",
        "seconds": dt,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", help="C1 | C2 | C3 | C5 (crossscalepatchmatch_amd/synth.py)")
    ap.add_argument("--schedule", default="raster", choices=["raster", "redblack"])
    ap.add_argument("--rb-rounds", type=int, default=1)
    ap.add_argument("--no-early-exit", action="store_true")
    ap.add_argument("--volumes", action="store_true", help="materialise f64 cost volumes (reference data flow) instead of fused cells")
    ap.add_argument("--raster-launches", action="store_true", help="raster sweep as one launch per anti-diagonal instead of the persistent kernel")
    ap.add_argument("--cc", default="GRD", choices=["GRD", "CEN"], help="cost function (BASELINE.json's metric is GRD; CEN for comparison)")
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
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
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
    d_out = [torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(2)]
    ctx = cs.StereoContext(dev_index)
    from crossscalepatchmatch_amd import capi
    ctx.set_option(capi.OPT_RASTER_LAUNCHES, int(args.raster_launches))
    sched = cs.SCHED_RASTER if args.schedule == "raster" else cs.SCHED_REDBLACK
    pm_kw = dict(seed=12345, schedule=sched, rb_rounds=args.rb_rounds, rb_neighbours=4, early_exit=0 if args.no_early_exit else 1)

    def step():
        ctx.set_images_device(d_l.data_ptr(), d_r.data_ptr(), w, h, w * 3)
        (ctx.build_cost_grd if args.cc == "GRD" else ctx.build_cost_cen)(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"], volumes=args.volumes)
        ctx.patchmatch(3, **pm_kw)
        for v in (0, 1):
            ctx.disparity_u8_device(v, cfg["dis_scale"], d_out[v].data_ptr())

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    ctx.enable_timing(not args.no_kernel_timing)
    ctx.reset_timing()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timing = ctx.timing()
    ctx.enable_timing(False)

    if rank == 0:
        mpix = w * h * args.steps * world / dt / 1e6
        taps_launch = 2 * ctx.taps_per_view_pass()  # one refinement launch evaluates every pixel of both views once
        out = {
            "metric": "Mpix/s disparity (%s, use_cs=true)" % args.cc, "value": mpix, "unit": "Mpix/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {w}x{h} max_dis={cfg['max_dis']} GRD scale_num={cfg['scale_num']} "
                                   f"reg_lambda={cfg['reg_lambda']} wnd=35 iters=3 (BASELINE.json configs[2] when C3)",
                       "cc_name": args.cc, "cost_source": "volumes" if args.volumes else "fused", "schedule": args.schedule, "raster_sweep": "per-diagonal launches" if args.raster_launches else "persistent", "rb_rounds": args.rb_rounds, "early_exit": not args.no_early_exit,
                       "pairs_per_gpu": args.steps, "parallelism": f"{world} independent pair stream(s), one per GPU"},
        }
        ref = timing["refine"]
        traffic = None  # HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE), committed
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if args.config == "C3" and args.cc == "GRD" and not args.volumes and os.path.exists(tpath):
            traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
        if ref["launches"]:
            avg_s = ref["ms"] / ref["launches"] / 1e3
            achieved = taps_launch * BYTES_PER_TAP / avg_s / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": "k_refine (plane cost evaluation, one PlaneRefinement halving step)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": taps_launch * BYTES_PER_TAP, "avg_launch_ms": avg_s * 1e3,
                "launches": ref["launches"],
                "note": "algorithmic bytes = in-image window taps x 19 B (SURVEY.md 8(d)); the tap stream is served on chip "
                        "(measured HBM traffic is ~0.2 % of it), so frac vs the HBM peak exceeds 1: the binding resources are the "
                        "CU L1 return path (TD ~97 % busy) and VALU issue, see DESIGN.md section 5; early exit skips taps but not "
                        "algorithmic bytes",
            }
        out["kernel_ms_per_step"] = {k: v["ms"] / args.steps for k, v in timing.items()}
        out["kernel_launches_per_step"] = {k: v["launches"] / args.steps for k, v in timing.items()}
        if world == 1 and not args.no_cpu_baseline and args.cc == "GRD":
            out["cpu_baseline"] = cpu_baseline(cfg, l, r, dev_index)
        # sanity of the result that was timed (not part of the timed region)
        dl = ctx.disparity_f64(0)
        out["bad2_vs_gt_left"] = synth.bad_fraction(dl, gl, 2.0)
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
