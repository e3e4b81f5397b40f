// CSPM/main.cc:13-14 includes "plane_cost\\cspc.h" (Windows path separator).  Forwarding header so that the reference's
// main.cc compiles unchanged; the class CSPC lives in the real header.
#pragma once
#include "plane_cost/cspc.h"
