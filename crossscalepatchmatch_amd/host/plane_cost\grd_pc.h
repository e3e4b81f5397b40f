// CSPM/main.cc:13-14 includes "plane_cost\\grd_pc.h" (Windows path separator).  Forwarding header so that the reference's
// main.cc compiles unchanged; the class GrdPC lives in the real header.
#pragma once
#include "plane_cost/grd_pc.h"
