// The reference spells its includes the Windows way (CSPM/main.cc:13-18: #include"cc\cen_cc.h").  GCC takes the
// backslash literally, so this file -- whose NAME contains the backslash -- forwards to the real header and lets
// the reference's main.cc compile against this host layer unchanged.
#pragma once
#include "cc/cen_cc.h"
