"""One rank of tests/test_gpu_batch.py::test_run_batch_two_ranks_one_gpu: gloo process group, compute on cuda:0."""
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
