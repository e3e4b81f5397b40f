// CSPM/main.cc:13-14 includes "plane_cost\grd_pc.h" but never instantiates GrdPC (main.cc:106-107 are commented out).
// Forwarding stub so that the reference's main.cc compiles unchanged; see the real header for what is offered.
#pragma once
#include "plane_cost/grd_pc.h"
