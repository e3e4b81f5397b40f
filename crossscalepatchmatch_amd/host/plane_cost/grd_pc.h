// grd_pc.h -- placeholder for the reference's on-the-fly single-scale plane cost GrdPC (CSPM/plane_cost/grd_pc.h).
// main.cc includes this header but never constructs a GrdPC (main.cc:106-107 are commented out); SURVEY.md 8(f4).
// The class is not offered by this build: PreSSPC is the single-scale cost the CLI reaches.
#pragma once
#include "i_plane_cost.h"
