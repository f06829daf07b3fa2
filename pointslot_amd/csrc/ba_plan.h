// Device-side descriptors of the object bundle adjustment (Optimizer::ObjectLocalBundleAdjustment,
// /root/reference/src/Optimizer.cc:755-1075).  One BaProb per object graph; all problems of a batch
// advance through the same kernel sequence, each according to its own BaState.
#pragma once
#include <stdint.h>

#define PS_BA_TILE 8          // poses per Schur tile edge (48 x 48 scalars)
#define PS_BA_TRACE 40        // LM iterations recorded per problem
#define PS_BA_MAX_POSES 128   // FREE poses per problem (the LDL^T panel of the reduced system lives in LDS)

enum { BA_PH_BEGIN = 0, BA_PH_LINEARIZE = 1, BA_PH_TRIAL = 2, BA_PH_DONE = 3,
       BA_PH_IDLE = 4 };   // (unused since the adaptive-depth experiment of r06 was dropped: a member parked for a round; every kernel passes it by)

struct BaProb {
  int32_t np, nl, ne;               // poses (free + fixed), points, edges
  int32_t pose_base, point_base, edge_base;   // element offsets into the batch-wide arrays
  int32_t csr_pose_base;            // into csr_off[]: np + 1 entries; edge ids (problem-local) in csr_pose_edges
  int32_t csr_point_base;           // nl + 1 entries
  int32_t csr_pose_edges_base, csr_point_edges_base;   // into the int32 edge-id lists (ne entries each)
  int64_t W_base;                   // doubles: [np][nl][18]  (6x3 blocks, row-major)
  int64_t S_base;                   // doubles: [6 np][6 np]  (lower triangle used), lda = 6 * np
  int32_t part_base;                // doubles: per-block partial sums, `part_cap` entries
  int32_t part_cap;
  float fx, fy, cx, cy, bf;
  // where the problem READS its linearisation (H_pp, b_p / H_ll, b_l / W / the chi2 partials): its own arrays - or, for a speculative twin, the
  // primary's: both twins stand at the same estimate when an iteration starts, so the twin does not linearise at all
  int32_t lin_pose_base, lin_point_base, lin_part_base;
  int64_t lin_W_base;
};

struct BaState {
  int32_t stage;        // 0: optimize(5) on all edges, 1: optimize(10) on the inliers, 2: finished
  int32_t phase;
  int32_t iter, max_iter, trial, n_bad;
  int32_t npa, nla;     // active poses / points of the current stage
  int32_t robust;       // Huber on (stage 0) / off
  int32_t ok2;
  int32_t ntrace;
  int32_t iters_done;   // LM iterations executed over both stages (for ms/iter reporting)
  int32_t trials_done;
  int32_t spec;         // s >= 1: the s-th speculative member of the problem in front of it (ba_decide): its damping trial runs with the lambda the
                        // primary's s-th next trial would use if the ones before are rejected
  int32_t depth;        // (primary) speculative members that run beside it in the current round: 1 .. group size - 1
  double lambda, ni, current_chi, ini_chi, rho;
};

// device pointers of one BA batch (passed by value to the kernels)
struct BaArrays {
  const BaProb* prob;
  BaState* state;
  double* poses; double* poses_bak; const uint8_t* pose_flags; int32_t* pidx; int32_t* pact;
  double* points; double* points_bak; uint8_t* lact;
  const int32_t* e_pose; const int32_t* e_point; const float* e_obs; const float* e_is2;
  uint8_t* e_state; double* chi2c; uint8_t* erase;
  const int32_t* csr_off; const int32_t* csr_edges;
  double* Hpp; double* bp; double* Hll; double* bl; double* Dinv; double* bs; double* xp; double* xl;
  double* W; double* S; double* part; double* trace;
  double* Wd;                  // W D^-1 of the current damping trial (the blocks of inactive points: zero), same layout as W: ba_prep writes it, ba_schur's A side reads it
  int32_t* ndone;
};
// ba_update: points per 256-thread block (16 lanes per point share the sum over the free poses)
#define PS_BA_UPD_PPB 16
