// Device-side problem descriptors of the matcher kernels.
#pragma once
#include <stdint.h>
#define PS_BF_MAX_TRAIN 4096   // train descriptors per brute-force problem (the reference caps object ORB at 1000)
#define PS_BF_TOPK 8
#define PS_BF_QPB 32           // queries per bf_topk block
struct BfProb { int32_t q_off, nq, t_off, nt; };
struct BfBlock { int32_t prob, q_first, q_count, pad; };
