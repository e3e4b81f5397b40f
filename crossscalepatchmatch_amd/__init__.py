"""crossscalepatchmatch_amd -- MI355X (gfx950) implementation of the PatchMatch-stereo hot path of
rookiepig/CrossScalePatchMatch behind a C ABI (include/cspm.h, libcspm_hip.so).

The package holds the HIP sources (csrc/), the thin ctypes view of the C ABI used by tests, bench and
the batch driver (capi.py), the multi-GPU batch dispatcher (batch.py) and the synthetic stereo-pair
generator used for measurement (synth.py).  There is no CPU fallback: importing works anywhere, but
every compute entry point raises when the HIP library or a gfx950 device is missing.
"""
from .capi import (CspmError, StereoContext, PmParams, library_path, load_library, build_library,  # noqa: F401
                   SCHED_RASTER, SCHED_REDBLACK, RNG_PER_PIXEL, RNG_ROW_SHARED, K_NAMES)
