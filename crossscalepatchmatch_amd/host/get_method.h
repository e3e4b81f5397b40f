// get_method.h -- name -> method factories with the reference's contract (CSPM/get_method.h:21-67,
// main.cc:39-55): unknown or unimplemented names yield NULL.
#pragma once
#include "ca_method.h"
#include "cc/cen_cc.h"
#include "cc/grd_cc.h"
#include "cc_method.h"

inline CCMethod *getCCType(const string &name) {
  if (name == "GRD") return new GrdCC();
  if (name == "CEN") return new CenCC();
  return NULL;  // "BSM", "CG": NULL in the reference too (main.cc:47-54)
}
inline CCMethod *GetCCType(const string &name) { return getCCType(name); }  // spelling used by main.cc:39
inline CAMethod *getCAType(const string &) { return NULL; }                 // "GF", "BF", "BOX", "NL", "ST"
// getPPType (CSPM/get_method.h:57-67): the reference declares it over a `PPMethod` it never defines (PPWM/WMPP.h and PPSG/SGPP.h are
// not in its tree; post-processing lives in CSPatchMatch::PostProcessing, cs_patchmatch.cc:508-588).  The name contract is kept:
// "NP" is NULL there, "SG" / "WM" have no class to construct -- NULL here, like every unimplemented name of the other factories.
class PPMethod;
inline PPMethod *getPPType(const string &) { return NULL; }
