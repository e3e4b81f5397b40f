"""Batch dispatch of independent stereo pairs over the GPUs of one node (BASELINE.json configs[3]: 200 KITTI-size
pairs on 8 x MI355X).  One process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo"
on CPU is used by the tests for the plumbing.

Stereo pairs are independent units: nothing in the hot path crosses pairs, so the only
collectives are for dispatch -- rank 0 broadcasts the run parameters, scatters each rank's contiguous block of
input pairs and gathers the 8-bit disparity maps (a few MB per pair; bandwidth-trivial against 7 x ~153 GB/s of
xGMI per GPU).  No collective inside the timed PatchMatch loop.

The per-pair compute function is injectable; the default one drives libcspm_hip.so and raises without a GPU
(there is no CPU fallback).
"""
import os

import numpy as np

PARAM_KEYS = ("w", "h", "max_dis", "dis_scale", "scale_num", "reg_lambda", "iters", "seed", "schedule", "use_pp")


def partition(n_items, world, rank):
    """contiguous block of rank `rank`: sizes differ by at most one, earlier ranks take the remainder."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def block_sizes(n_items, world):
    return [partition(n_items, world, r)[1] - partition(n_items, world, r)[0] for r in range(world)]


class HipPairFn:
    """(l_bgr, r_bgr) torch uint8 tensors on the rank's GPU -> (l_dis, r_dis) torch uint8 tensors, via the C ABI.

    Stream ordering: libcspm runs on a stream of its own (a torch side stream handed to cspm_set_stream).  Before a pair
    is enqueued that stream waits (device-side) for torch's current stream -- the one the NCCL/RCCL scatter was ordered
    into by ProcessGroupNCCL.wait() -- so k_pack_bgr never reads a half-received input block; after the pair, torch's
    current stream waits for the libcspm stream, so the gather that follows sees finished maps.  No host synchronisation
    per pair."""

    def __init__(self, device_index):
        import torch
        from .capi import StereoContext
        self.device = torch.device("cuda", device_index)
        self.ctx = StereoContext(device_index)  # raises CspmError without libcspm_hip.so / a gfx950 device
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx.set_stream(self.stream.cuda_stream)

    def __call__(self, l, r, p):
        import torch
        h, w = int(p["h"]), int(p["w"])
        assert l.is_cuda and l.device == self.device and l.is_contiguous() and r.is_contiguous()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self.ctx.set_images_device(l.data_ptr(), r.data_ptr(), w, h, w * 3)
        self.ctx.build_cost_grd(int(p["max_dis"]), 35, int(p["scale_num"]), float(p["reg_lambda"]))
        self.ctx.patchmatch(int(p["iters"]), seed=int(p["seed"]), schedule=int(p["schedule"]))
        if int(p["use_pp"]):
            lo, ro = self.ctx.postprocess(int(p["dis_scale"]))  # synchronises (host buffers)
            return torch.from_numpy(lo).to(l.device), torch.from_numpy(ro).to(l.device)
        out = [torch.empty((h, w), dtype=torch.uint8, device=l.device) for _ in range(2)]
        for v in (0, 1):
            out[v].record_stream(self.stream)
            self.ctx.disparity_u8_device(v, int(p["dis_scale"]), out[v].data_ptr())
        l.record_stream(self.stream)
        r.record_stream(self.stream)
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return out[0], out[1]

    def close(self):
        self.ctx.synchronize()  # also reports a raster sweep that timed out
        self.ctx.close()


def run_batch(pairs, params, pair_fn, device="cpu", dist=None):
    """pairs: on rank 0 a uint8 array/tensor [n, 2, h, w, 3] (ignored elsewhere); params: dict with PARAM_KEYS on
    rank 0.  Returns on rank 0 a uint8 tensor [n, 2, h, w] (disparity maps in input order), None on other ranks."""
    import torch
    if dist is None or not dist.is_initialized():
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device(device)
    # 1. run parameters: one small broadcast from rank 0
    meta = torch.zeros(len(PARAM_KEYS) + 1, dtype=torch.float64, device=dev)
    if rank == 0:
        n = int(pairs.shape[0])
        meta = torch.tensor([float(params[k]) for k in PARAM_KEYS] + [float(n)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.broadcast(meta, src=0)
    p = {k: meta[i].item() for i, k in enumerate(PARAM_KEYS)}
    n = int(meta[-1].item())
    h, w = int(p["h"]), int(p["w"])
    sizes = block_sizes(n, world)
    cap = max(sizes) if sizes else 0
    # 2. scatter the input blocks (padded to the largest block so every rank receives the same shape)
    mine = torch.zeros((cap, 2, h, w, 3), dtype=torch.uint8, device=dev)
    if world > 1:
        chunks = None
        if rank == 0:
            src = torch.as_tensor(pairs, dtype=torch.uint8).to(dev)
            chunks = []
            for r in range(world):
                a, b = partition(n, world, r)
                c = torch.zeros((cap, 2, h, w, 3), dtype=torch.uint8, device=dev)
                c[: b - a] = src[a:b]
                chunks.append(c)
        dist.scatter(mine, chunks, src=0)
    else:
        mine = torch.as_tensor(pairs, dtype=torch.uint8).to(dev)
    # 3. the hot path, pair by pair, no communication
    out = torch.zeros((cap, 2, h, w), dtype=torch.uint8, device=dev)
    for i in range(sizes[rank]):
        q = dict(p)
        q["seed"] = int(p["seed"]) + partition(n, world, rank)[0] + i  # per-pair seed = global pair index
        dl, dr = pair_fn(mine[i, 0].contiguous(), mine[i, 1].contiguous(), q)
        out[i, 0], out[i, 1] = dl, dr
    # 4. gather the 8-bit maps on rank 0
    if world > 1:
        got = [torch.zeros_like(out) for _ in range(world)] if rank == 0 else None
        dist.gather(out, got, dst=0)
        if rank != 0:
            return None
        return torch.cat([got[r][: sizes[r]] for r in range(world)], 0)
    return out[:n]


def main():
    """python -m torch.distributed.run --nproc-per-node N -m crossscalepatchmatch_amd.batch --pairs 200"""
    import argparse
    import time
    import torch
    import torch.distributed as dist
    from . import synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--schedule", type=int, default=0)
    ap.add_argument("--use_pp", type=int, default=0)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank = dist.get_rank() if world > 1 else 0
    cfg = dict(synth.CONFIGS[args.config])
    pairs, params = None, None
    if rank == 0:
        pairs = np.stack([np.stack(synth.make_pair(cfg["w"], cfg["h"], cfg["max_dis"], cfg["regions"], cfg["seed"] + i)[:2])
                          for i in range(args.pairs)])
        params = dict(w=cfg["w"], h=cfg["h"], max_dis=cfg["max_dis"], dis_scale=cfg["dis_scale"], scale_num=cfg["scale_num"],
                      reg_lambda=cfg["reg_lambda"], iters=3, seed=12345, schedule=args.schedule, use_pp=args.use_pp)
    fn = HipPairFn(local_rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_batch(pairs, params, fn, device=f"cuda:{local_rank}", dist=dist if world > 1 else None)
    fn.ctx.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"{args.pairs} pairs of {cfg['w']}x{cfg['h']} on {world} GPU(s): {dt:.2f} s end to end incl. dispatch, "
              f"{args.pairs * cfg['w'] * cfg['h'] / dt / 1e6:.3f} Mpix/s, result {tuple(out.shape)}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
