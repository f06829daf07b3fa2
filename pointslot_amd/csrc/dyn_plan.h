// Device-side descriptor of one DynamicStaticDiscrimination problem (dyn_kernels.hip / opt_host.hip).
#pragma once
#include <stdint.h>
#include "se3.h"
#define PS_DYN_MAX 2048       // object points per detection
struct DynProb {
  int32_t off, n;
  Se3 tco;                    // Tco of the object in the last frame
  Se3 trel;                   // Tcw_cur * Tcw_last^-1
  double fx, fy, cx, cy;
  float mbf;
};
