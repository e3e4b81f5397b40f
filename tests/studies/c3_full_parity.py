"""One whole C3 pair (BASELINE.json configs[2]: 1242x375, max_dis 128, GRD, 5 levels, lambda 0.3, 3 iterations, raster sweeps) through
the HIP path and through the CPU oracle in the same device order: every plane, every stored cost and both 8-bit maps must be
identical.  The GPU suite checks C3 on samples and properties (the oracle needs minutes for the pair); this is the full check,
run once per round on the GPU box:  python tests/studies/c3_full_parity.py [pair index] > profiles/rNN_c3_full_parity.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crossscalepatchmatch_amd import capi, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

# argument: a pair index (synthetic KITTI-size pair, seed 2000 + index) or the name of an adversarial kind of synth.make_adversarial at
# the C3 size (tie-heavy / saturated / flat inputs through the DMA-filled tables, range tests and clusters of the full-size row kernels)
arg = sys.argv[1] if len(sys.argv) > 1 else "0"
if arg in synth.ADVERSARIAL_KINDS:
    cfg = dict(synth.CONFIGS["C3"])
    l, r = synth.make_adversarial(arg, cfg["w"], cfg["h"], cfg["max_dis"], seed=77)
    index = arg
else:
    index = int(arg)
    cfg, l, r, gl, gr = synth.make_config("C3", index=index)
seed = 12345
ctx = capi.StereoContext(0)
ctx.set_images(l, r)
t = time.time()
ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
ctx.patchmatch(3, seed=seed, schedule=0)
got = [ctx.get_planes(v) for v in (0, 1)]
maps = [ctx.disparity_u8(v, cfg["dis_scale"]) for v in (0, 1)]
t_gpu = time.time() - t
print(f"C3 pair {index} ({'adversarial kind' if isinstance(index, str) else 'synthetic seed 2000+' + str(index)}), PatchMatch seed {seed}: GPU {t_gpu:.2f} s host to host", flush=True)
t = time.time()
pc = po.PlaneCost(l, r, cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
pm = po.PatchMatch(l, r, cfg["max_dis"], cfg["dis_scale"])
pm.run(3, pc, False, seed=seed, schedule=po.SCHED_RASTER, sum_order=po.SUM_DEVICE, wavefront=True)  # the oracle's sweep as a wavefront: identical to its serial loop (tests/test_oracle_primitives.py)
t_cpu = time.time() - t
print(f"oracle, device order, {po.effective_cpus()} threads: {t_cpu:.1f} s", flush=True)
ok = True
for v in (0, 1):
    npar, cost = got[v]
    P = pm.planes(v)
    checks = {"normals": np.array_equal(npar[..., :3], P[..., 0:3]), "plane parameters": np.array_equal(npar[..., 3:6], P[..., 6:9]),
              "stored costs": np.array_equal(cost, pm.min_cost(v)), "8-bit map": np.array_equal(maps[v], pm.dis(v))}
    for k, e in checks.items():
        print(f"view {v}: {k:18s} {'identical' if e else 'DIFFER'}  ({npar.shape[0] * npar.shape[1]} pixels)")
        ok &= e
print("RESULT:", "bit-identical" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
