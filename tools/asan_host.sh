#!/bin/bash
# Sanitizer pass over the C++ host layer: the host sources are rebuilt with AddressSanitizer and
# UndefinedBehaviorSanitizer and exercised
#   (1) without a GPU: the image codecs (PNG / PNM in, PNG / PNM / PFM out) and the gflags-compatible parser through
#       tests/helpers/host_io_check.cc, incl. truncated and corrupt files;
#   (2) on a GPU box (skipped when no device answers): the reference-style command line end to end on a small synthetic pair,
#       single pair and --batch_list with a failing line.
# Leak checking is off for (2) only: the HIP runtime keeps process-lifetime allocations that LeakSanitizer reports.
#   usage: tools/asan_host.sh          (from the repo root; needs crossscalepatchmatch_amd/libcspm_hip.so for step 2)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
H=$R/crossscalepatchmatch_amd/host
B=$R/tests/_build/asan
mkdir -p "$B"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1 -std=c++14"
g++ $SAN -I "$H" -o "$B/host_io_check" "$R/tests/helpers/host_io_check.cc" "$H/image_io.cc" -lz
cd "$B"
python3 - "$R" <<'PY'
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import pngio
rng = np.random.default_rng(1)
rgb = rng.integers(0, 256, (19, 23, 3)).astype(np.uint8)
for filt in (0, 1, 2):
    pngio.write_png("in.png", rgb, filter_type=filt)
    out = subprocess.check_output(["./host_io_check", "in.png", "c.png", "g.pgm", "--max_dis=60", "--use_cs"]).decode()
    assert np.array_equal(pngio.read_png("c.png"), rgb), filt
raw = open("in.png", "rb").read()
for cut in (0, 7, 8, 20, 33, len(raw) // 2, len(raw) - 1):      # truncated files must fail cleanly, not crash
    open("cut.png", "wb").write(raw[:cut])
    p = subprocess.run(["./host_io_check", "cut.png", "c.png", "g.pgm"], capture_output=True)
    assert p.returncode in (0, 3) and b"Sanitizer" not in p.stderr and b"runtime error" not in p.stderr, (cut, p.returncode, p.stderr[-300:])
    assert p.returncode == 3 or cut >= len(raw) - 12  # only a cut inside the trailing IEND chunk may still decode
for k in range(40):                                               # flipped bytes: any exit code but a sanitizer abort
    b = bytearray(raw)
    b[int(rng.integers(8, len(b)))] ^= 1 << int(rng.integers(0, 8))
    open("bad.png", "wb").write(bytes(b))
    p = subprocess.run(["./host_io_check", "bad.png", "c.png", "g.pgm"], capture_output=True)
    assert b"Sanitizer" not in p.stderr and b"runtime error" not in p.stderr, p.stderr.decode()[-400:]
print("asan/ubsan: image codecs + flag parser clean")
PY
if [ -f "$R/crossscalepatchmatch_amd/libcspm_hip.so" ] && python3 -c "
import sys; sys.path.insert(0, '$R')
import crossscalepatchmatch_amd as cs
sys.exit(0 if cs.load_library().cspm_device_count() > 0 else 1)" 2>/dev/null; then
  g++ $SAN -I "$H" -o "$B/cspm_main_asan" "$H/main.cc" "$H/host_impl.cc" "$H/image_io.cc" -L"$R/crossscalepatchmatch_amd" -lcspm_hip -lz \
      -Wl,-rpath,"$R/crossscalepatchmatch_amd"
  python3 - "$R" <<'PY'
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import pngio
from crossscalepatchmatch_amd import synth
l, r, _, _ = synth.make_pair(96, 64, 16, regions=3, seed=5)
pngio.write_png("l.png", l[..., ::-1]); pngio.write_png("r.png", r[..., ::-1])
open("list.txt", "w").write("l.png r.png ld1.png rd1.png l1.pfm r1.pfm\nnope_l.png nope_r.png x.png y.png\nl.png r.png ld2.png rd2.png\n")
PY
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:use_sigaltstack=0  # sigaltstack: ASan's per-thread alternate stack trips its own CHECK when a worker thread exits under the HIP runtime
  ./cspm_main_asan --l_img_file=l.png --r_img_file=r.png --l_dis_file=ld.png --r_dis_file=rd.png --max_dis=16 --dis_scale=4 --cc_name=GRD \
      --use_cs=true --use_pp=true --reg_lambda=0.3 --quiet
  set +e
  ./cspm_main_asan --batch_list=list.txt --in_flight=2 --max_dis=16 --dis_scale=4 --cc_name=GRD --use_cs=true --reg_lambda=0.3 --quiet > batch.out 2> batch.err
  rc=$?
  set -e
  # exactly the "one pair failed" exit, the summary line, and nothing from a sanitizer (a crash must not pass for the expected failure)
  [ "$rc" = 1 ] && grep -q "Batch: 3 pairs" batch.out && grep -q "1 failed" batch.out && ! grep -qE "Sanitizer|runtime error" batch.err || { cat batch.out batch.err | tail -40; exit 1; }
  cmp ld1.png ld2.png
  echo "asan/ubsan: command line on the GPU clean (single pair + batch list with a failing line)"
  # (3) ThreadSanitizer over the worker-thread batch (round 6): host sources rebuilt with -fsanitize=thread, 8 pairs of two sizes
  #     through --devices 0,0 --in_flight 2 (four workers: DeviceSlot, parked contexts, the live registry, the shared output stream).
  #     The HIP runtime is not instrumented; a report counts only when one of its frames is in the host layer's own sources.
  g++ -fsanitize=thread -fno-omit-frame-pointer -g -O1 -std=c++14 -pthread -I "$H" -o "$B/cspm_main_tsan" "$H/main.cc" "$H/host_impl.cc" "$H/image_io.cc" \
      -L"$R/crossscalepatchmatch_amd" -lcspm_hip -lz -Wl,-rpath,"$R/crossscalepatchmatch_amd"
  python3 - "$R" <<'PY'
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import pngio
from crossscalepatchmatch_amd import synth
lines = []
for i in range(8):
    w, h = ((96, 64), (64, 48))[i % 2]
    l, r, _, _ = synth.make_pair(w, h, 16, regions=3, seed=40 + i)
    pngio.write_png(f"tl{i}.png", l[..., ::-1]); pngio.write_png(f"tr{i}.png", r[..., ::-1])
    lines.append(f"tl{i}.png tr{i}.png tld{i}.png trd{i}.png")
open("tlist.txt", "w").write("\n".join(lines) + "\n")
PY
  # setarch -R: no address-space randomisation -- TSan refuses the high-entropy mmap layout of this kernel ("unexpected memory mapping")
  export TSAN_OPTIONS="halt_on_error=0 exitcode=0 report_signal_unsafe=0 log_path=$B/tsan_report"
  rm -f "$B"/tsan_report.*
  setarch "$(uname -m)" -R ./cspm_main_tsan --batch_list=tlist.txt --devices=0,0 --in_flight=2 --max_dis=16 --dis_scale=4 --cc_name=GRD --use_cs=true --use_pp=true \
      --reg_lambda=0.3 --quiet || { echo "tsan: the batch itself failed"; exit 1; }
  mkdir -p "$R/gpurun_out" && cat "$B"/tsan_report.* > "$R/gpurun_out/tsan_report.txt" 2>/dev/null || true
  # A report counts when one of its two RACING ACCESSES is in the host layer's own code: the first frame below libtsan's interceptor of
  # the "Write / Read of size" or "Previous write / read" stack.  The uninstrumented HSA / HIP runtime synchronises its own allocations
  # in ways TSan cannot see (operator new in one runtime thread, delete in another); such reports carry host-layer frames only further
  # down, as the caller that entered the runtime.
  python3 - "$R/gpurun_out/tsan_report.txt" <<'PY'
import re, sys
text = open(sys.argv[1]).read() if len(sys.argv) > 1 else ""
reports = text.split("WARNING: ThreadSanitizer")[1:]
own = re.compile(r"host/(main\.cc|host_impl\.cc|image_io\.cc|[a-z_/]+\.h)")
bad = []
for r in reports:
    for m in re.finditer(r"^  (?:Write|Read|Previous write|Previous read|Atomic write|Atomic read|Previous atomic write|Previous atomic read) of size.*?\n((?:    #\d+ .*\n)+)", r, re.M):
        frames = [f for f in m.group(1).splitlines() if "libtsan" not in f]
        if frames and own.search(frames[0]):
            bad.append(r)
            break
print(f"tsan: {len(reports)} reports, {len(bad)} with a racing access in the host layer")
if bad:
    print("WARNING: ThreadSanitizer" + bad[0][:4000])
    sys.exit(1)
PY
  echo "tsan: worker-thread batch clean (no racing access in the host layer; the rest is inside the uninstrumented HSA / HIP runtime)"
else
  echo "asan/ubsan: no GPU (or libcspm_hip.so missing): the command-line step is skipped"
fi
