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
