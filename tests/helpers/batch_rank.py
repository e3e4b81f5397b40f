"""One rank of tests/test_gpu_batch.py.
  default (CSPM_BATCH_BACKEND unset / "gloo"): test_run_batch_two_ranks_one_gpu -- gloo process group, compute on cuda:0;
  CSPM_BATCH_BACKEND=nccl: test_run_batch_over_rccl_world1 -- an RCCL process group on cuda:0 (world size 1 on a one-GPU box),
  device tensors end to end, run_batch(force_collectives=True)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from crossscalepatchmatch_amd import batch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_batch import PARAMS  # noqa: E402


def main():
    out_dir = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("CSPM_BATCH_BACKEND", "gloo")
    if backend == "nccl":
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)  # RCCL
        rank = dist.get_rank()
        fn = batch.HipPairFn(dev.index, in_flight=2)
        pairs = torch.from_numpy(np.load(os.path.join(out_dir, "pairs.npy"))).to(dev) if rank == 0 else None  # the batch is resident in HBM
        got = batch.run_batch(pairs, PARAMS if rank == 0 else None, fn, device=str(dev), dist=dist, chunk_pairs=2, force_collectives=True)
        torch.cuda.synchronize()
        fn.close()
        if rank == 0:
            np.save(os.path.join(out_dir, "out.npy"), got.cpu().numpy())
        dist.destroy_process_group()
        return
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.init()
    fn = batch.HipPairFn(0, in_flight=2)

    class OnGpu:
        """gloo moves host tensors; the per-pair function wants its inputs in HBM and hands device tensors back (.cpu() waits
        for torch's current stream, which order_after_pairs() puts behind the pair's stream)"""
        def __call__(self, l, r, p):
            dl, dr = fn(l.cuda(0).contiguous(), r.cuda(0).contiguous(), p)
            fn.order_after_pairs()
            return dl.cpu(), dr.cpu()

        def finalize(self):
            fn.finalize()

    pairs = np.load(os.path.join(out_dir, "pairs.npy")) if rank == 0 else None
    got = batch.run_batch(pairs, PARAMS if rank == 0 else None, OnGpu(), device="cpu", dist=dist, chunk_pairs=2)
    fn.close()
    if rank == 0:
        np.save(os.path.join(out_dir, "out.npy"), got.numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
