// test-only stand-in: the reference's commfunc.h includes <gflags/gflags.h>; the sources compiled by tests/test_reference_loops.py
// (cs_patchmatch.cc, pre_*_pc.cc, grd_cc.cpp) use none of it
#pragma once
