#!/usr/bin/env python
"""The C++ product path end to end, with files: `cspm_main --batch_list` on N KITTI-size pairs (BASELINE.json configs[2] / [3]:
1242x375, max_dis 128, GRD, 5 levels, lambda 0.3), PNG in / PNG out, one pair at a time against K pairs in flight (worker
threads, one device context each) and against two worker sets on the same GPU (`--devices 0,0`, the multi-GPU rehearsal a
1-GPU box allows).  Checks that every run wrote the same maps, and that pair 0's maps equal the C-ABI path.

    python tools/cli_batch_bench.py [--pairs 24] [--out gpurun_out/r06_cli_batch.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLI = os.path.join(ROOT, "crossscalepatchmatch_amd", "cspm_main")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=24)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_cli_batch.json"))
    ap.add_argument("--runs", default="1:0,2:0,3:0,4:0,2:0+0", help="comma-separated in_flight:devices (devices joined with +)")
    args = ap.parse_args()
    from PIL import Image
    from crossscalepatchmatch_amd import synth
    tmp = tempfile.mkdtemp(prefix="cspm_cli_batch_")
    cfg = None
    t0 = time.perf_counter()
    for i in range(args.pairs):
        cfg, l, r, _, _ = synth.make_config(args.config, index=i)
        Image.fromarray(l[..., ::-1]).save(os.path.join(tmp, f"l{i}.png"), compress_level=1)
        Image.fromarray(r[..., ::-1]).save(os.path.join(tmp, f"r{i}.png"), compress_level=1)
    print(f"wrote {args.pairs} pairs to {tmp} in {time.perf_counter() - t0:.1f} s", file=sys.stderr)
    flags = [f"--max_dis={cfg['max_dis']}", f"--dis_scale={cfg['dis_scale']}", "--cc_name=GRD", f"--use_cs={'true' if cfg['scale_num'] else 'false'}",
             f"--reg_lambda={cfg['reg_lambda']}", f"--use_pp={'true' if cfg.get('use_pp') else 'false'}", "--seed=12345"]
    results, first = [], None
    for run in args.runs.split(","):
        k, devs = run.split(":")
        devs = devs.replace("+", ",")
        tag = f"k{k}_d{devs.replace(',', '')}"
        with open(os.path.join(tmp, f"list_{tag}.txt"), "w") as f:
            for i in range(args.pairs):
                f.write(f"{tmp}/l{i}.png {tmp}/r{i}.png {tmp}/{tag}_ld{i}.png {tmp}/{tag}_rd{i}.png\n")
        cmd = [CLI, f"--batch_list={tmp}/list_{tag}.txt", f"--in_flight={k}", f"--devices={devs}"] + flags
        t = time.perf_counter()
        out = subprocess.check_output(cmd).decode()
        wall = time.perf_counter() - t
        m = re.search(r"Batch: (\d+) pairs in ([0-9.eE+-]+) s, (\d+) failed", out)
        fb = re.search(r"Batch fallbacks: (\d+) raster sweeps.*?, (\d+) optional", out)
        assert m and int(m.group(1)) == args.pairs and int(m.group(3)) == 0, out[-2000:]
        per_pair = [float(x) for x in re.findall(r"Total Time: ([0-9.eE+-]+)", out)]
        maps = [np.asarray(Image.open(f"{tmp}/{tag}_{s}d{i}.png")) for i in range(args.pairs) for s in "lr"]
        if first is None:
            first = maps
        same = all(np.array_equal(a, b) for a, b in zip(first, maps))
        results.append({"in_flight": int(k), "devices": devs, "pairs": args.pairs, "batch_seconds": float(m.group(2)), "process_wall_seconds": wall,
                        "ms_per_pair_end_to_end": float(m.group(2)) / args.pairs * 1e3, "mpix_per_s": cfg["w"] * cfg["h"] * args.pairs / float(m.group(2)) / 1e6,
                        "mean_total_time_one_pair_ms": float(np.mean(per_pair)) * 1e3 if per_pair else None,
                        "sweep_fallbacks": int(fb.group(1)) if fb else None, "volume_fallbacks": int(fb.group(2)) if fb else None,
                        "maps_identical_to_first_run": bool(same)})
        print(json.dumps(results[-1]), file=sys.stderr)
        assert same, tag
    # pair 0 through the C ABI
    import crossscalepatchmatch_amd as cs
    cfg, l, r, _, _ = synth.make_config(args.config, index=0)
    ctx = cs.StereoContext(0)
    ctx.set_images(l, r)
    ctx.build_cost_grd(cfg["max_dis"], 35, cfg["scale_num"], cfg["reg_lambda"])
    ctx.patchmatch(3, seed=12345, schedule=0)
    want = ctx.postprocess(cfg["dis_scale"]) if cfg.get("use_pp") else (ctx.disparity_u8(0, cfg["dis_scale"]), ctx.disparity_u8(1, cfg["dis_scale"]))
    abi_equal = bool(np.array_equal(first[0], want[0]) and np.array_equal(first[1], want[1]))
    ctx.close()
    out = {"what": "cspm_main --batch_list, PNG in / PNG out, wall clock of the whole batch (decode + upload + compute + download + encode)",
           "workload": f"{args.config}: {cfg['w']}x{cfg['h']} max_dis={cfg['max_dis']} GRD scale_num={cfg['scale_num']} reg_lambda={cfg['reg_lambda']}, {args.pairs} distinct pairs",
           "runs": results, "pair0_equals_c_abi": abi_equal}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))
    assert abi_equal


if __name__ == "__main__":
    main()
