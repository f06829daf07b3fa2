// Device-side problem descriptors of the matcher kernels.
#pragma once
#include <stdint.h>
#define PS_BF_MAX_TRAIN 4096   // train descriptors per brute-force problem (the reference caps object ORB at 1000)
#define PS_BF_TOPK 8
#define PS_BF_QPB 32           // queries per bf_topk block
#define PS_BF_SMALL_NT 512     // problems with at most this many train descriptors run in bf_topk_small (keys in registers)
struct BfProb { int32_t q_off, nq, t_off, nt; };
struct BfBlock { int32_t prob, q_first, q_count, pad; };

// ---- windowed (projection) matching ----
#define PS_PJ_CAP 256          // candidates kept per query in the everyday key format
#define PS_PJ_CAP_WIDE 1024    // ... in the wide format that re-runs problems whose windows overflowed (frames of <= PS_PJ_WIDE_MAX_N features)
#define PS_PJ_WIDE_MAX_N 8191
#define PS_GRID_COLS 64        // FRAME_GRID_COLS / ROWS, /root/reference/include/Frame.h:40-41
#define PS_GRID_ROWS 48
struct PjProb {
  int32_t t_off, nt;           // train features
  int32_t q_off, nq;           // queries
  int32_t c_off;               // first query's slot in the candidate store (= q_off, except in the compact store of a wide re-run)
  int32_t grid_off;            // into cell_off (COLS*ROWS+1 entries per problem); cell_idx shares t_off
  float min_x, min_y, gw_inv, gh_inv;
  int32_t th_dist;             // TH_HIGH (100) or TH_HIGH_FORDYNAMIC (130)
  int32_t ratio_test;          // 1: best/second-best test with nn_ratio when both are on the same level
  float nn_ratio;
  int32_t check_ori;           // rotation histogram (frame-to-frame variant)
  int32_t use_bbox;
  int32_t frame_mode;          // 1: queries are produced by pj_project (SearchByProjection(cur, last))
  // frame_mode inputs
  float tcw[16], tlw[16];
  float fx, fy, cx, cy, mbf, mb;
  float bounds[4];             // mnMinX, mnMaxX, mnMinY, mnMaxY
  float scale[8];
  float th;
  int32_t mono;
};

// device pointers of one windowed-matching batch (passed by value to the kernels)
struct PjArrays {
  const PjProb* prob;
  // train side
  const float* tx; const float* ty; const int32_t* toct; const float* tang; const float* tur; const uint8_t* tdesc;
  const uint8_t* tocc; const uint8_t* tbbox; const int32_t* cell_off; const int32_t* cell_idx;
  // query side
  uint8_t* qvalid; float* qu; float* qv; float* qur; float* qrad; float* qrer; int32_t* qminl; int32_t* qmaxl;
  const uint8_t* qdesc; const uint8_t* qobs; const float* qang;
  const float* qxw; const int32_t* qoct;     // frame mode
  // work / outputs
  uint32_t* cand; int32_t* ncand; int32_t* match; int32_t* nmatch;
  int32_t* overflow;                         // per problem: queries whose window held more candidates than the key format stores
  int32_t* qbest;                            // per query: matched train (or -1), for the rotation pass
  uint4* ttop;                               // per query: the four smallest candidate keys among the trains free at entry (pj_gather)
  uint8_t* qbin;
};

// ---- ORBmatcher::Fuse search (projection + gates + chi-square-gated best match per candidate point) ----
struct FuProb {
  int32_t t_off, nt, q_off, nq, grid_off;
  float min_x, min_y, gw_inv, gh_inv;
  float R[9], t[3], ow[3];
  float fx, fy, cx, cy, bf;
  double bounds[4];
  float scale[8], inv_sigma2[8];
  float log_scale;
  int32_t n_levels;
  float th;
};
struct FuArrays {
  const FuProb* prob;
  const float* tx; const float* ty; const int32_t* toct; const float* tur; const uint8_t* tdesc;
  const int32_t* cell_off; const int32_t* cell_idx;
  const uint8_t* qvalid; const float* qpos; const float* qnormal; const float* qmin; const float* qmax; const uint8_t* qdesc;
  int32_t* best_idx; int32_t* best_dist;
};
