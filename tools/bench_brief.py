#!/usr/bin/env python
"""stdin: one bench.py JSON line -> one short line (value, ms per pair, kernel ms per pair)"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", "in-flight", d["config"]["pairs_in_flight_per_gpu"], f"{d['value']:.3f} Mpix/s {d['ms_per_step']:.1f} ms/pair",
      {k: round(v, 1) for k, v in d.get("kernel_ms_per_step", d.get("kernel_bracket_ms_per_pair_overlapping", {})).items()})
