#!/usr/bin/env python
"""Instruction census of the hot loop of a kernel: compiles csrc/cspm_api.hip with -save-temps (gfx950), finds the innermost
loops of the kernel and counts instructions per loop body.  Usage: tools/isa_census.py [mangled-name-substring] > profiles/..."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kern = sys.argv[1] if len(sys.argv) > 1 else "k_refineILb1ELi1E"
tmp = tempfile.mkdtemp()
src = os.path.join(ROOT, "crossscalepatchmatch_amd", "csrc", "cspm_api.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                       "-save-temps", "-c", "-o", os.path.join(tmp, "a.o"), src], cwd=tmp, stderr=subprocess.DEVNULL)
asm = open(os.path.join(tmp, "cspm_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
start = next(i for i, l in enumerate(asm) if l.startswith("_ZN4cspm") and kern in l and ":" in l and l.split(":")[0].endswith(l.split(":")[0]) and not l.startswith("\t") and l.rstrip().split()[0].endswith(":"))
end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
body = asm[start:end]
INSTR = re.compile(r"\s+[vs]_|\s+ds_|\s+global_|\s+scratch_")
total = sum(1 for l in body if INSTR.match(l))
print(f"kernel {asm[start][:-1]}: {total} instructions in total")
# innermost loops: from an "Inner Loop Header" label to the first backward branch to it
heads = [i for i, l in enumerate(body) if "Inner Loop Header" in l]
for h in heads:
    lab = None
    for k in range(h, max(h - 10, -1), -1):
        m = re.match(r"(\.LBB\d+_\d+):", body[k])
        if m:
            lab = m.group(1)
            break
    if lab is None:
        continue
    e = next((i for i in range(h, len(body)) if re.search(r"s_cbranch\w*\s+" + re.escape(lab) + r"\b", body[i])), None)
    if e is None:
        continue
    ins = [l.split()[0] for l in body[h:e + 1] if re.match(r"\s+[a-z]", l) and not l.strip().startswith(";")]
    cnt = collections.Counter(ins)
    valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
    lds = sum(v for k, v in cnt.items() if k.startswith("ds_"))
    vmem = sum(v for k, v in cnt.items() if k.startswith(("global_", "scratch_", "flat_", "buffer_")))
    if valu < 60:
        continue
    print(f"\nloop {lab} (lines {h}-{e}): {len(ins)} instructions: {valu} VALU, {lds} LDS, {vmem} VMEM, {sum(v for k, v in cnt.items() if k.startswith('s_'))} SALU/waitcnt")
    print("  " + ", ".join(f"{k} {v}" for k, v in cnt.most_common() if k.startswith(("v_", "ds_", "global_", "scratch_"))))
