// cspc.h -- placeholder for the reference's on-the-fly cross-scale plane cost CSPC (CSPM/plane_cost/cspc.h).
// main.cc includes this header but never constructs a CSPC (main.cc:106-107 are commented out); SURVEY.md 8(f4).
// The class is not offered by this build: PreCSPC is the cross-scale cost the CLI reaches.
#pragma once
#include "i_plane_cost.h"
