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

PARAM_KEYS = ("w", "h", "max_dis", "dis_scale", "scale_num", "reg_lambda", "iters", "seed", "schedule", "use_pp", "cc")
CC_CODES = {"GRD": 0, "CEN": 1, "IMG": 2}  # params["cc"]: the cost family (cc/grd_cc, cc/cen_cc; IMG = GrdPC / CSPC)


def partition(n_items, world, rank):
    """contiguous block of rank `rank`: sizes differ by at most one, earlier ranks take the remainder."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def block_sizes(n_items, world):
    return [partition(n_items, world, r)[1] - partition(n_items, world, r)[0] for r in range(world)]


class HipPairFn:
    """(l_bgr, r_bgr) torch uint8 tensors on the rank's GPU -> (l_dis, r_dis) torch uint8 tensors, via the C ABI.

    `in_flight` contexts, each with a HIP stream of its own, take the pairs round-robin: a pair's latency-bound raster sweep and
    the tail of every launch leave CUs idle that the next pair's kernels use (the same arrangement as bench.py).  Nothing
    waits on the host per pair -- finalize() synchronises every context once, which is also where an error inside an
    asynchronous run (a sweep that timed out with more than one pair enqueued) is raised.

    Stream ordering: before a pair is enqueued its stream waits (device-side) for the event `after` -- recorded by run_batch on
    torch's current stream right after Work.wait() ordered the NCCL/RCCL scatter of the pair's block into it -- or, without one,
    for torch's current stream as it is now; so k_pack_bgr never reads a half-received input block.  The pair writes its maps
    straight into the caller's output views (`out=`); torch's current stream is NOT made to wait per pair (that would chain the
    pairs of different contexts one after the other): run_batch calls order_after_pairs() before it lets a collective overwrite
    a receive buffer, and finalize() before it reads the maps.

    Lifetime of the tensors a pair reads and writes (they are used on streams torch's allocator does not own, and the library may
    REWRITE a pair's maps inside a later synchronising call when its sweep timed out, include/cspm.h): each pair records an event on
    its stream and is held per context; when a context takes its next pair, the library has dropped the earlier pairs' replay
    requests, so those whose event has completed are released -- at most the pairs actually in flight stay pinned, however long
    the loop runs (a 200-pair block held 800 tensors until round 4).  finalize() is REQUIRED before the maps are consumed and
    before the last pairs' tensors are freed: it is the synchronising call that raises asynchronous errors and makes the maps final."""
    writes_out = True

    def __init__(self, device_index, in_flight=2):
        import torch
        from .capi import StereoContext
        self.device = torch.device("cuda", device_index)
        self.ctxs = [StereoContext(device_index) for _ in range(max(1, int(in_flight)))]  # raises CspmError without libcspm_hip.so / a device
        if len(self.ctxs) >= 2 and not os.environ.get("CSPM_SWEEP_FOLD"):
            # the pairs share the GPU: four-wave sweep workgroups (the coarsest level folded) leave room for two of another pair's
            # refinement workgroups per CU instead of one (include/cspm.h CSPM_OPT_SWEEP_FOLD); the environment variable, if set, decides
            from .capi import OPT_SWEEP_FOLD
            for c in self.ctxs:
                c.set_option(OPT_SWEEP_FOLD, 1)
        # every context runs on the non-blocking HIP stream it owns; torch sees it as an ExternalStream for the waits below.  (Streams
        # from torch's pool may share a hardware queue -- with GPU_MAX_HW_QUEUES at its default, two of them did: the pairs of
        # two contexts then ran strictly one after the other, 192 instead of 172 ms per KITTI-size pair.)
        self.streams = [torch.cuda.ExternalStream(c.stream_ptr(), device=self.device) for c in self.ctxs]
        self.calls = 0
        self.ctx = self.ctxs[0]
        import collections
        self._held = [collections.deque() for _ in self.ctxs]  # per context: (event, tensors) of pairs enqueued and possibly still running

    def __call__(self, l, r, p, out=None, after=None):
        import torch
        h, w = int(p["h"]), int(p["w"])
        assert l.is_cuda and l.device == self.device and l.is_contiguous() and r.is_contiguous()
        k = self.calls % len(self.ctxs)
        self.calls += 1
        ctx, stream = self.ctxs[k], self.streams[k]
        if after is not None:
            stream.wait_event(after)
        else:
            stream.wait_stream(torch.cuda.current_stream(self.device))
        ctx.set_images_device(l.data_ptr(), r.data_ptr(), w, h, w * 3)
        cc = int(p.get("cc", 0))
        args = (int(p["max_dis"]), 35, int(p["scale_num"]), float(p["reg_lambda"]))
        if cc == CC_CODES["IMG"]:
            ctx.build_cost_img(*args)
        elif cc == CC_CODES["CEN"]:
            ctx.build_cost_cen(*args)
        else:
            ctx.build_cost_grd(*args)
        ctx.patchmatch(int(p["iters"]), seed=int(p["seed"]), schedule=int(p["schedule"]))
        if out is None:
            out = [torch.empty((h, w), dtype=torch.uint8, device=l.device) for _ in range(2)]
        assert all(o.is_contiguous() and o.device == self.device and o.dtype == torch.uint8 for o in out)
        if int(p["use_pp"]):
            ctx.postprocess_device(int(p["dis_scale"]), out[0].data_ptr(), out[1].data_ptr())
        else:
            for v in (0, 1):
                ctx.disparity_u8_device(v, int(p["dis_scale"]), out[v].data_ptr())
        held = self._held[k]
        while held and held[0][0].query():  # earlier pairs of this context: superseded (no replay) and finished
            held.popleft()
        done = torch.cuda.Event()
        done.record(stream)
        held.append((done, out[0], out[1], l, r))
        self.max_held = max(getattr(self, "max_held", 0), sum(len(q) for q in self._held))
        return out[0], out[1]

    def order_after_pairs(self):
        """torch's current stream waits (device-side) for everything enqueued on the pair streams so far"""
        import torch
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def finalize(self):
        """host-synchronise every context: raises if anything inside the asynchronous runs failed"""
        try:
            for c in self.ctxs:
                c.synchronize()
        finally:
            for q in self._held:
                q.clear()

    def close(self):
        try:
            self.finalize()
        finally:
            for c in self.ctxs:
                c.close()


def run_batch(pairs, params, pair_fn, device="cpu", dist=None, chunk_pairs=4, force_collectives=False):
    """pairs: on rank 0 a uint8 array/tensor [n, 2, h, w, 3] (ignored elsewhere); params: dict with PARAM_KEYS on
    rank 0 ("cc" optional).  Returns on rank 0 a uint8 tensor [n, 2, h, w] (disparity maps in input order), None on other ranks.

    Dispatch: every rank owns a contiguous block of pairs.  The blocks travel in rounds of `chunk_pairs` pairs per rank
    (one scatter per round, the next round's scatter in flight while this round's pairs are being enqueued), straight out of
    rank 0's copy of the batch: no per-rank padded staging copies, receive memory bounded by two chunks.

    A single rank needs no collective and skips them -- unless `force_collectives` (with an initialised process group): then the
    broadcast, the chunked asynchronous scatter and the gather run with world size 1, which is how the RCCL code path (backend
    "nccl": device tensors, views of the batch as scatter inputs, pair streams ordered against the collectives' stream) is
    exercised on a one-GPU box."""
    import torch
    if dist is None or not dist.is_initialized():
        world, rank = 1, 0
        force_collectives = False
    else:
        world, rank = dist.get_world_size(), dist.get_rank()
    coll = world > 1 or bool(force_collectives)
    dev = torch.device(device)
    # 1. run parameters: one small broadcast from rank 0
    meta = torch.zeros(len(PARAM_KEYS) + 1, dtype=torch.float64, device=dev)
    if rank == 0:
        n = int(pairs.shape[0])
        meta = torch.tensor([float(params.get(k, 0)) for k in PARAM_KEYS] + [float(n)], dtype=torch.float64, device=dev)
    if coll:
        dist.broadcast(meta, src=0)
    p = {k: meta[i].item() for i, k in enumerate(PARAM_KEYS)}
    n = int(meta[-1].item())
    h, w = int(p["h"]), int(p["w"])
    sizes = block_sizes(n, world)
    cap = max(sizes) if sizes else 0
    first = partition(n, world, rank)[0]
    out = torch.zeros((cap, 2, h, w), dtype=torch.uint8, device=dev)

    def compute(block, base, count, after=None):  # the hot path, pair by pair, no communication
        for i in range(count):
            q = dict(p)
            q["seed"] = int(p["seed"]) + first + base + i  # per-pair seed = global pair index
            if getattr(pair_fn, "writes_out", False):  # the maps land in `out` directly, nothing is ordered into torch's stream
                pair_fn(block[i, 0].contiguous(), block[i, 1].contiguous(), q, out=(out[base + i, 0], out[base + i, 1]), after=after)
            else:
                dl, dr = pair_fn(block[i, 0].contiguous(), block[i, 1].contiguous(), q)
                out[base + i, 0], out[base + i, 1] = dl, dr

    if coll:
        chunk = max(1, min(int(chunk_pairs), cap)) if cap else 1
        rounds = (cap + chunk - 1) // chunk
        src = torch.as_tensor(pairs, dtype=torch.uint8).to(dev) if rank == 0 else None
        filler = torch.zeros((chunk, 2, h, w, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
        recv = [torch.zeros((chunk, 2, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        staged = {}

        def issue(j):
            lst = None
            if rank == 0:
                lst = []
                for r in range(world):
                    a = partition(n, world, r)[0] + j * chunk
                    cnt = max(0, min(chunk, sizes[r] - j * chunk))
                    if cnt == chunk:
                        lst.append(src[a:a + chunk])  # a view of the batch: nothing is copied on rank 0
                    else:                             # the ragged tail of a block: the one staged piece per round
                        t = staged.setdefault((j, r), filler.clone() if cnt else filler)
                        if cnt:
                            t[:cnt] = src[a:a + cnt]
                        lst.append(t)
            return dist.scatter(recv[j % 2], lst, src=0, async_op=True)

        work = issue(0) if rounds else None
        for j in range(rounds):
            work.wait()
            # this round's pairs wait for exactly this: the scatter of their block.  (Waiting for the current stream itself would,
            # from the second round on, also wait for everything order_after_pairs() puts into it: a barrier between rounds.)
            ready = None
            if dev.type == "cuda" and getattr(pair_fn, "writes_out", False):
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(dev))
            if j >= 1 and hasattr(pair_fn, "order_after_pairs"):
                pair_fn.order_after_pairs()  # round j+1 lands in the buffer round j-1's pairs read
            nxt = issue(j + 1) if j + 1 < rounds else None
            compute(recv[j % 2], j * chunk, max(0, min(chunk, sizes[rank] - j * chunk)), after=ready)
            work = nxt
    else:
        compute(torch.as_tensor(pairs, dtype=torch.uint8).to(dev), 0, sizes[rank])
    # errors inside the asynchronous runs surface here, before any map is handed on
    fin = getattr(pair_fn, "finalize", None)
    if fin is not None:
        fin()
    # 2. gather the 8-bit maps on rank 0
    if coll:
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
    ap.add_argument("--cc", default="GRD", choices=sorted(CC_CODES))
    ap.add_argument("--in-flight", type=int, default=2)
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
                      reg_lambda=cfg["reg_lambda"], iters=3, seed=12345, schedule=args.schedule, use_pp=args.use_pp, cc=CC_CODES[args.cc])
    fn = HipPairFn(local_rank, in_flight=args.in_flight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_batch(pairs, params, fn, device=f"cuda:{local_rank}", dist=dist if world > 1 else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fn.close()
    if rank == 0:
        print(f"{args.pairs} pairs of {cfg['w']}x{cfg['h']} on {world} GPU(s): {dt:.2f} s end to end incl. dispatch, "
              f"{args.pairs * cfg['w'] * cfg['h'] / dt / 1e6:.3f} Mpix/s, result {tuple(out.shape)}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
