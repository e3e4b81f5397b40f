#!/usr/bin/env python
"""rocprofv3 --pmc output of tools/pmc_run.sh -> the per-launch figures bench.py quotes for one kernel.
usage: pmc_to_json.py <gpurun_out/dir> <kernel substring> <out.json> [description]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # the file is quoted by bench.py only while this hash matches the sources it times

root, kern, out = sys.argv[1], sys.argv[2], sys.argv[3]
desc = sys.argv[4] if len(sys.argv) > 4 else ""
sums = collections.defaultdict(float)
disp = collections.defaultdict(set)
dur_ns = {}
for f in sorted(glob.glob(root + "/pmc*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]:
            continue
        c = r["Counter_Name"]
        sums[c] += float(r["Counter_Value"])
        disp[c].add(r["Dispatch_Id"])
        if c == "GRBM_GUI_ACTIVE":
            dur_ns[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
n = len(disp["SQ_INSTS_VALU"])
per = {c: v / max(1, len(disp[c])) for c, v in sums.items()}
gui = per["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
dur = sum(dur_ns.values()) / len(dur_ns)
res = {
    "kernel": kern, "description": desc, "launches_profiled": n, "kernel_source_hash": kernel_source_hash(),
    "valu_winstr_per_launch": per["SQ_INSTS_VALU"],
    "shader_cycles_per_launch_pmc": gui, "launch_ns_pmc": dur, "shader_clock_ghz": gui / dur,
    "valu_busy_frac_pmc": per["SQ_INSTS_VALU"] * 4.0 / 1024 / gui,
    "lds_busy_frac": per["SQ_LDS_IDX_ACTIVE"] / 256 / gui,
    "lds_bank_conflict_share": per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"],
    "lds_instr_per_launch": per["SQ_INSTS_LDS"], "vmem_rd_instr_per_launch": per["SQ_INSTS_VMEM_RD"],
    "ta_busy_frac": per["TA_TA_BUSY_sum"] / 256 / gui, "td_busy_frac": per["TD_TD_BUSY_sum"] / 256 / gui,
    "waves_per_launch": per["SQ_WAVES"], "mean_waves_per_simd": per["SQ_WAVE_CYCLES"] * 4 / 1024 / gui,
    "fetch_size_kb": per.get("FETCH_SIZE"), "write_size_kb": per.get("WRITE_SIZE"),
    "l2_hit_rate": (per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])) if per.get("TCC_HIT_sum") is not None and per.get("TCC_MISS_sum") is not None else None,
    # HBM bytes: 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md for wide reads; an upper bound otherwise) + WRITE_SIZE
    "hbm_bytes_per_launch": (2 * per.get("FETCH_SIZE", 0) + per.get("WRITE_SIZE", 0)) * 1024,
    "raw_per_launch": per,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "raw_per_launch"}, indent=1))
