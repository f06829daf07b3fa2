// Device-side problem descriptors of the optimiser kernels.
#pragma once
#include <stdint.h>
#define PS_PO_MAX_K 16     // SE3 vertices (objects) per pose-only problem
#define PS_PO_TRACE 64     // LM iterations recorded per problem when tracing
struct PoProb { int32_t v_off, k, mode; float fx, fy, cx, cy, bf; };   // mode 0 PoseOptimization, 1 CFSE3
struct PoVertex { int32_t e_begin, e_end; };
