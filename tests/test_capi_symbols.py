"""-m "not gpu": libcspm_hip.so builds (hipcc cross-compiles gfx950 without a GPU), loads, exports every
symbol include/cspm.h declares, and refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "cspm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cspm_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _declared() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(cs.library_path()):
        cs.build_library()
    lib = C.CDLL(cs.library_path())
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = cs.load_library()
    assert L.cspm_device_count() == 0
    with pytest.raises(cs.CspmError, match="no HIP device"):
        cs.StereoContext(0)
    p = capi.PmParams()
    assert L.cspm_pm_default_params(C.byref(p)) == 0 and p.early_exit == 1  # pure host logic still answers


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under crossscalepatchmatch_amd/ (the product) or tools/ (profiling and build
    helpers) may import, link or load it -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do."""
    for top in ("crossscalepatchmatch_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".sh", ".h", ".hip", ".cc", ".cpp", ".c", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "pyoracle" not in txt and "cspm_oracle" not in txt and "libcspm_oracle" not in txt, os.path.join(dirpath, f)


def test_bench_fails_loudly_without_a_gpu():
    """bench.py has no CPU path either: without a visible GPU it must exit non-zero with a message, for any --gpus."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([], ["--gpus", "2"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"] + extra, capture_output=True, timeout=600)
        assert p.returncode != 0
        assert b"needs a GPU" in p.stderr + p.stdout


def test_kernel_source_hash_ignores_comments_but_not_code(tmp_path, monkeypatch):
    """bench.py quotes the committed rocprofv3 counters (profiles/refine_pmc.json) only while kernel_source_hash() equals the hash
    the file was stamped with: editing a comment must not invalidate a measurement, editing code must."""
    import bench
    d = tmp_path / "crossscalepatchmatch_amd" / "csrc"
    d.mkdir(parents=True)
    (d / "a.h").write_text("// header\nint f(int x) { return x + 1; }  /* plus one */\n")
    (d / "b.hip").write_text("__global__ void k() {}\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    h0 = bench.kernel_source_hash()
    (d / "a.h").write_text("// another header comment\n\nint f(int x) {\n  return x + 1;   /* still plus one */\n}\n")
    assert bench.kernel_source_hash() == h0
    (d / "a.h").write_text("// header\nint f(int x) { return x + 2; }\n")
    assert bench.kernel_source_hash() != h0


def test_effective_cpus_is_bounded_by_what_the_container_may_use():
    from oracle import pyoracle as po
    n = po.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        assert n <= len(os.sched_getaffinity(0))
    except AttributeError:
        pass


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/cspm.h is the boundary a cgo / JNI / ctypes binding is written against: it must compile as pedantic C99 (no C++
    in the signatures) and a C program must link against the shared library and get a status code -- not an exception -- back."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "crossscalepatchmatch_amd")
    src = tmp_path / "bind.c"
    src.write_text('#include <stdio.h>\n#include "cspm.h"\n'
                   "int main(void) {\n"
                   "  cspm_ctx *ctx = NULL;\n"
                   "  int rc = cspm_create(&ctx, 1 << 20);  /* no such device anywhere: an error code either way */\n"
                   '  printf("%d|%s\\n", rc, cspm_last_error(NULL));\n'
                   "  if (ctx) cspm_destroy(ctx);\n"
                   "  return rc == CSPM_OK;\n"
                   "}\n")
    exe = tmp_path / "bind"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lcspm_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    rc, msg = out.stdout.strip().split("|", 1)
    assert int(rc) < 0 and msg  # a negative status and a message


def test_python_constants_mirror_the_header():
    """capi.py repeats the option keys, kernel classes and phase codes of include/cspm.h by hand: they must not drift apart."""
    import re
    from crossscalepatchmatch_amd import capi
    hdr = open(os.path.join(ROOT, "include", "cspm.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define\s+(CSPM_[A-Z0-9_]+)\s+(-?\d+)\s*(?:/\*.*)?$", hdr, re.M)}
    opts = {k: v for k, v in defs.items() if k.startswith("CSPM_OPT_")}
    assert len(set(opts.values())) == len(opts), "two options share a key"
    for name, value in opts.items():
        py = name[len("CSPM_"):]
        assert getattr(capi, py) == value, name
    for k, name in enumerate(capi.K_NAMES):
        assert defs["CSPM_K_" + {"grd": "GRD", "init": "INIT", "spatial": "SPATIAL", "view": "VIEW", "refine": "REFINE", "misc": "MISC", "post": "POST"}[name]] == k
    assert defs["CSPM_K_COUNT"] == len(capi.K_NAMES)
