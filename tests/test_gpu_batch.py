"""batch.run_batch with the real per-pair function (HipPairFn -> libcspm_hip.so) on a device: world 1 in-process, and two
ranks sharing GPU 0 with a gloo control plane (the only way to run world > 1 on a 1-GPU box).  Every map must equal the
direct C-ABI path on the same pair and seed."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from crossscalepatchmatch_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, D = 72, 40, 12
PARAMS = dict(w=W, h=H, max_dis=D, dis_scale=4, scale_num=3, reg_lambda=0.3, iters=2, seed=500, schedule=0, use_pp=0)


def _pairs(n):
    return np.stack([np.stack(synth.make_pair(W, H, D, 3, 40 + i)[:2]) for i in range(n)])


def _direct(ctx, pairs, use_pp=0):
    out = []
    for i, (l, r) in enumerate(pairs):
        ctx.set_images(l, r)
        ctx.build_cost_grd(D, 35, PARAMS["scale_num"], PARAMS["reg_lambda"])
        ctx.patchmatch(PARAMS["iters"], seed=PARAMS["seed"] + i, schedule=0)
        out.append(ctx.postprocess(4) if use_pp else (ctx.disparity_u8(0, 4), ctx.disparity_u8(1, 4)))
    return np.array(out)


@pytest.mark.parametrize("in_flight", [1, 2, 3])
@pytest.mark.parametrize("use_pp", [0, 1])
def test_run_batch_world1_on_device(gpu_ctx, use_pp, in_flight):
    """1, 2 or 3 contexts in flight per rank (pairs round-robin over them, nothing waits per pair; run_batch synchronises once
    through HipPairFn.finalize before it hands the maps on); with and without post-processing on the device."""
    import torch
    from crossscalepatchmatch_amd import batch
    pairs = _pairs(5)
    fn = batch.HipPairFn(0, in_flight=in_flight)
    got = batch.run_batch(pairs, dict(PARAMS, use_pp=use_pp), fn, device="cuda:0", dist=None)
    torch.cuda.synchronize()
    fn.close()
    np.testing.assert_array_equal(got.cpu().numpy(), _direct(gpu_ctx, pairs, use_pp))


def test_run_batch_cost_family(gpu_ctx):
    """params["cc"] selects the cost family on every rank: census here"""
    import torch
    from crossscalepatchmatch_amd import batch
    pairs = _pairs(2)
    fn = batch.HipPairFn(0, in_flight=2)
    got = batch.run_batch(pairs, dict(PARAMS, cc=batch.CC_CODES["CEN"]), fn, device="cuda:0", dist=None).cpu().numpy()
    fn.close()
    for i, (l, r) in enumerate(pairs):
        gpu_ctx.set_images(l, r)
        gpu_ctx.build_cost_cen(D, 35, PARAMS["scale_num"], PARAMS["reg_lambda"])
        gpu_ctx.patchmatch(PARAMS["iters"], seed=PARAMS["seed"] + i, schedule=0)
        for v in (0, 1):
            np.testing.assert_array_equal(got[i, v], gpu_ctx.disparity_u8(v, 4))


def test_run_batch_two_ranks_one_gpu(gpu_ctx, tmp_path):
    """2 processes, gloo rendezvous on 127.0.0.1, both computing on cuda:0 with HipPairFn (2 contexts in flight each); 7 pairs ->
    blocks of 4 and 3, dispatched in rounds of 2 pairs per rank (the ragged last round is the staged one)."""
    pairs = _pairs(7)
    np.save(tmp_path / "pairs.npy", pairs)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "helpers", "batch_rank.py"), str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=600)
    got = np.load(tmp_path / "out.npy")
    np.testing.assert_array_equal(got, _direct(gpu_ctx, pairs))


def test_run_batch_over_rccl_world1(gpu_ctx, tmp_path):
    """The RCCL code path itself (torch.distributed backend "nccl"), on the one GPU a test box has: a process group of world
    size 1 on cuda:0 and run_batch(force_collectives=True) -- the parameter broadcast, the chunked asynchronous scatter of views
    of the device-resident batch (rounds of 2 pairs, a ragged last round), HipPairFn with 2 contexts in flight ordered against
    the collectives' stream, and the gather of the device maps all execute as RCCL kernels.  A separate process, so that the
    suite's own process never owns a process group.  Maps == the direct C-ABI path."""
    pairs = _pairs(5)
    np.save(tmp_path / "pairs.npy", pairs)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", CSPM_BATCH_BACKEND="nccl")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "batch_rank.py"), str(tmp_path)], check=True, env=env, timeout=600)
    got = np.load(tmp_path / "out.npy")
    np.testing.assert_array_equal(got, _direct(gpu_ctx, pairs))


def test_run_batch_200_pairs_bounded_pinning(gpu_ctx):
    """BASELINE configs[3] is a batch of 200 pairs: run_batch over 200 (small) pairs on one rank -- maps 0, 103 and 199 equal the
    direct C-ABI path with seed = global pair index -- and HipPairFn pins only the pairs in flight, not the whole block
    (round-4 review: `_held` grew to 200 x 4 tensors until finalize())."""
    import torch
    from crossscalepatchmatch_amd import batch
    base = _pairs(8)
    pairs = np.stack([base[i % 8] if (i // 8) % 2 == 0 else base[i % 8][:, :, ::-1].copy() for i in range(200)])  # 16 distinct inputs, 200 seeds
    fn = batch.HipPairFn(0, in_flight=2)
    got = batch.run_batch(pairs, dict(PARAMS), fn, device="cuda:0", dist=None)
    torch.cuda.synchronize()
    assert fn.calls == 200
    assert sum(len(q) for q in fn._held) == 0  # finalize() released everything
    # What stays pinned is what has not FINISHED (the host enqueues faster than the GPU computes, so that can be many pairs), never
    # what has: 30 pairs, wait for the device, then one more pair per context -- each context drops its finished pairs as it takes
    # the next one, so only the two new pairs (and nothing of the 30) remain.
    dev = torch.device("cuda", 0)
    d_pairs = torch.from_numpy(pairs[:32]).to(dev)
    outs = torch.zeros((32, 2, H, W), dtype=torch.uint8, device=dev)
    q = dict(PARAMS)
    for i in range(30):
        fn(d_pairs[i, 0], d_pairs[i, 1], dict(q, seed=PARAMS["seed"] + i), out=(outs[i, 0], outs[i, 1]))
    assert sum(len(h) for h in fn._held) >= 2
    for st in fn.streams:
        st.synchronize()
    for i in (30, 31):
        fn(d_pairs[i, 0], d_pairs[i, 1], dict(q, seed=PARAMS["seed"] + i), out=(outs[i, 0], outs[i, 1]))
    assert sum(len(h) for h in fn._held) == 2, [len(h) for h in fn._held]
    fn.finalize()
    np.testing.assert_array_equal(outs[:32].cpu().numpy(), got[:32].cpu().numpy())  # the same pairs and seeds as the batch's first 32
    fn.close()
    got = got.cpu().numpy()
    assert got.shape == (200, 2, H, W)
    for i in (0, 103, 199):
        l, r = pairs[i]
        gpu_ctx.set_images(l, r)
        gpu_ctx.build_cost_grd(D, 35, PARAMS["scale_num"], PARAMS["reg_lambda"])
        gpu_ctx.patchmatch(PARAMS["iters"], seed=PARAMS["seed"] + i, schedule=0)
        for v in (0, 1):
            np.testing.assert_array_equal(got[i, v], gpu_ctx.disparity_u8(v, 4), err_msg=f"pair {i} view {v}")


def test_table_volumes_are_optional_memory(gpu_ctx, tmp_path):
    """The device-cell volumes are an accelerator, not a requirement (advisor, round 4): a context whose share of the free memory
    does not cover them, and one whose hipMalloc for them FAILS (fault injection: CSPM_OPT_FAULT_VOLUME_ALLOC, a set_option test hook
    that nothing in the environment reaches), both go on with computed tables and produce the same planes; the environment knob
    CSPM_TABLE_VOLUMES=0 is not overridden by the Python wrapper's defaults.  A cost object that runs without the volumes it wanted
    asks again after CSPM_OPT_VOLUME_RETRY_PAIRS reuses (advisor, round 5: one transient shortage must not last for ever)."""
    import crossscalepatchmatch_amd as cs
    from crossscalepatchmatch_amd import capi
    l, r = _pairs(1)[0]
    gpu_ctx.set_images(l, r)
    gpu_ctx.build_cost_grd(D, 35, 5, 0.3)
    assert gpu_ctx.get_option(capi.OPT_TABLE_VOLUMES_ACTIVE) == 1
    gpu_ctx.patchmatch(2, seed=4, schedule=0)
    want = [gpu_ctx.get_planes(v) for v in (0, 1)]

    def same_planes(ctx, tag):
        ctx.patchmatch(2, seed=4, schedule=0)
        for v in (0, 1):
            npar, cost = ctx.get_planes(v)
            np.testing.assert_array_equal(npar, want[v][0], err_msg=tag)
            np.testing.assert_array_equal(cost, want[v][1], err_msg=tag)

    for env, fault, fallbacks, comes_back in (({"CSPM_VOLUMES_MEM_FRACTION": "0"}, 0, 0, False), ({}, 3, 1, True), ({"CSPM_TABLE_VOLUMES": "0"}, 0, 0, False)):
        os.environ.update(env)
        try:
            ctx = cs.StereoContext(0)
        finally:
            for k in env:
                del os.environ[k]
        try:
            assert ctx.get_option(capi.OPT_VOLUME_RETRY_PAIRS) == 16
            ctx.set_option(capi.OPT_VOLUME_RETRY_PAIRS, 3)
            if fault:
                ctx.set_option(capi.OPT_FAULT_VOLUME_ALLOC, fault)
            ctx.set_images(l, r)
            ctx.build_cost_grd(D, 35, 5, 0.3)
            assert ctx.get_option(capi.OPT_TABLE_VOLUMES_ACTIVE) == 0, (env, fault)
            assert ctx.get_option(capi.OPT_VOLUME_FALLBACKS) == fallbacks, (env, fault)
            same_planes(ctx, str((env, fault)))
            for _ in range(2):  # the next pairs of the same shape reuse the buffers as they are: no retry yet, no error
                ctx.build_cost_grd(D, 35, 5, 0.3)
                assert ctx.get_option(capi.OPT_TABLE_VOLUMES_ACTIVE) == 0
            ctx.build_cost_grd(D, 35, 5, 0.3)  # third reuse: asks again -- the injected failure was transient, a veto of the memory share is not
            assert ctx.get_option(capi.OPT_TABLE_VOLUMES_ACTIVE) == (1 if comes_back else 0), (env, fault)
            assert ctx.get_option(capi.OPT_VOLUME_FALLBACKS) == fallbacks
            same_planes(ctx, str((env, fault, "after the retry")))
        finally:
            ctx.close()
