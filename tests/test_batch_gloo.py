"""Multi-rank dispatch on CPU: world_size 2 and 3 over gloo.  Exercises the plumbing the
8-GPU run uses over RCCL -- parameter broadcast, scatter of contiguous pair blocks (uneven sizes), per-pair
seeds, gather in input order -- with a stand-in per-pair function (the HIP function needs a GPU and is covered
by the -m gpu tests; it raises here, there is no CPU fallback)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from crossscalepatchmatch_amd import batch


def test_partition_covers_everything_once():
    for n in (0, 1, 7, 8, 200):
        for world in (1, 2, 3, 8):
            spans = [batch.partition(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = batch.block_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    assert batch.block_sizes(200, 8) == [25] * 8


def _fake_pair_fn(l, r, p):
    """deterministic stand-in: depends on the pixels, the per-pair seed and the parameters it was sent"""
    k = (int(p["seed"]) * 7 + int(p["max_dis"])) % 251
    dl = (l.to(torch.int32).sum(-1) + k) % 256
    dr = (r.to(torch.int32).sum(-1) * 3 + k) % 256
    return dl.to(torch.uint8), dr.to(torch.uint8)


class _CountingFn:
    """the stand-in with a finalize hook: run_batch must call it exactly once, after the last pair and before the gather"""
    def __init__(self):
        self.pairs = self.finalized = 0

    def __call__(self, l, r, p):
        assert not self.finalized
        self.pairs += 1
        return _fake_pair_fn(l, r, p)

    def finalize(self):
        self.finalized += 1


def _worker(rank, world, port, n_pairs, ret, chunk=4):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    pairs = rng.integers(0, 256, (n_pairs, 2, 6, 9, 3)).astype(np.uint8) if rank == 0 else None
    params = dict(w=9, h=6, max_dis=16, dis_scale=4, scale_num=5, reg_lambda=0.3, iters=3, seed=100, schedule=0, use_pp=0) if rank == 0 else None
    fn = _CountingFn()
    out = batch.run_batch(pairs, params, fn, device="cpu", dist=dist, chunk_pairs=chunk)
    assert fn.finalized == 1 and fn.pairs == batch.block_sizes(n_pairs, world)[rank]
    if rank == 0:
        ret["out"] = out.numpy()
        ret["pairs"] = pairs
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_pairs,chunk", [(2, 5, 4), (3, 7, 1), (2, 1, 4), (2, 9, 2), (3, 2, 4)])
def test_scatter_compute_gather(world, n_pairs, chunk):
    """rounds of `chunk` pairs per rank: blocks longer than a round, ragged last rounds, ranks with nothing to do"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_pairs, ret, chunk), nprocs=world, join=True)
    out, pairs = ret["out"], ret["pairs"]
    assert out.shape == (n_pairs, 2, 6, 9)
    p = dict(max_dis=16)
    for i in range(n_pairs):
        p["seed"] = 100 + i
        dl, dr = _fake_pair_fn(torch.from_numpy(pairs[i, 0]), torch.from_numpy(pairs[i, 1]), p)
        np.testing.assert_array_equal(out[i, 0], dl.numpy())
        np.testing.assert_array_equal(out[i, 1], dr.numpy())


def _worker_c4(rank, world, port, n_pairs, ret, chunk):
    """BASELINE configs[3] in shape: 200 pairs over 8 ranks.  Besides the maps, every rank reports which global indices (= seeds) it
    computed and in which order, and how many scatter rounds it saw."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(15)
    pairs = rng.integers(0, 256, (n_pairs, 2, 3, 5, 3)).astype(np.uint8) if rank == 0 else None
    params = dict(w=5, h=3, max_dis=16, dis_scale=4, scale_num=5, reg_lambda=0.3, iters=3, seed=1000, schedule=0, use_pp=0) if rank == 0 else None
    seeds = []

    class Fn(_CountingFn):
        def __call__(self, l, r, p):
            seeds.append(int(p["seed"]))
            return super().__call__(l, r, p)

    fn = Fn()
    scatters = []
    real_scatter = dist.scatter

    def counting_scatter(*a, **kw):
        scatters.append(1)
        return real_scatter(*a, **kw)

    dist.scatter = counting_scatter
    try:
        out = batch.run_batch(pairs, params, fn, device="cpu", dist=dist, chunk_pairs=chunk)
    finally:
        dist.scatter = real_scatter
    ret[f"seeds{rank}"] = seeds
    ret[f"rounds{rank}"] = len(scatters)
    assert fn.finalized == 1
    if rank == 0:
        ret["out"] = out.numpy()
        ret["pairs"] = pairs
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pairs,chunk", [(8, 200, 4), (8, 5, 4)])
def test_c4_shape_200_pairs_over_8_ranks(world, n_pairs, chunk):
    """configs[3]: 200 pairs sharded over 8 ranks in rounds of 4 -- 25 pairs per rank = 7 scatter rounds with a ragged last one
    (1 pair); and 5 pairs over 8 ranks (three ranks idle: they still take part in every collective).  Input order, seed = global
    pair index, one finalize per rank, every rank sees the same number of rounds."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_c4, args=(world, _free_port(), n_pairs, ret, chunk), nprocs=world, join=True)
    out, pairs = ret["out"], ret["pairs"]
    assert out.shape == (n_pairs, 2, 3, 5)
    sizes = batch.block_sizes(n_pairs, world)
    cap = max(sizes)
    rounds = -(-cap // min(chunk, cap))
    for r in range(world):
        a, b = batch.partition(n_pairs, world, r)
        assert list(ret[f"seeds{r}"]) == [1000 + i for i in range(a, b)], f"rank {r}: seeds are global pair indices, in order"
        assert ret[f"rounds{r}"] == rounds
    if n_pairs == 200:
        assert sizes == [25] * 8 and rounds == 7
    p = dict(max_dis=16)
    for i in range(n_pairs):
        p["seed"] = 1000 + i
        dl, dr = _fake_pair_fn(torch.from_numpy(pairs[i, 0]), torch.from_numpy(pairs[i, 1]), p)
        np.testing.assert_array_equal(out[i, 0], dl.numpy())
        np.testing.assert_array_equal(out[i, 1], dr.numpy())


def test_single_process_path_and_no_cpu_fallback():
    rng = np.random.default_rng(6)
    pairs = rng.integers(0, 256, (3, 2, 4, 5, 3)).astype(np.uint8)
    params = dict(w=5, h=4, max_dis=8, dis_scale=1, scale_num=0, reg_lambda=0.0, iters=1, seed=1, schedule=0, use_pp=0)
    out = batch.run_batch(pairs, params, _fake_pair_fn)
    assert tuple(out.shape) == (3, 2, 4, 5)
    if not torch.cuda.is_available():
        import crossscalepatchmatch_amd as cs
        with pytest.raises(cs.CspmError):
            batch.HipPairFn(0)
