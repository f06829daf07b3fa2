// Device-side state of the lockstep tracker (track_kernels.hip / track_host.hip): S independent stereo sequences whose
// per-frame chain  Frame::Frame -> TrackWithMotionModel -> TrackLocalMap  (/root/reference/src/Tracking.cc:2840-3160)
// runs on the device without a host round trip.  Every array is [S][cap] (element (s, i) at s * cap + i) unless noted.
#pragma once
#include <stdint.h>
#include "match_plan.h"
#include "opt_plan.h"

#define PS_TRK_NCELL (PS_GRID_COLS * PS_GRID_ROWS)

// what the tracking thread keeps of a Frame (Frame.h: mvKeysUn, mvuRight, mvDepth, mDescriptors, mGrid, mvpMapPoints, mvbOutlier, mTcw)
struct TrkFrame {
  float* x; float* y; float* angle; float* uright; float* depth;
  float* xw;              // [S][cap][3] world position of mvpMapPoints[i]
  int32_t* octave;
  int32_t* mp_id;         // index into the sequence's local map, -1 for temporal points
  int32_t* cell_off;      // [S][NCELL + 1] mGrid as CSR, cell = ix * 48 + iy
  int32_t* cell_idx;
  uint8_t* desc;          // [S][cap][32]
  uint8_t* mp_valid; uint8_t* mp_observed; uint8_t* outlier;
  int32_t* n;             // [S]
  float* tcw;             // [S][16]
};

enum { TRK_NOT_INITIALIZED = 0, TRK_OK = 1, TRK_LOST = 2 };
enum { TRK_PH_IDLE = 0, TRK_PH_MM, TRK_PH_POSE1, TRK_PH_LM, TRK_PH_POSE2, TRK_PH_FINISH };

struct TrkSeq {
  int32_t state, phase;
  int32_t have_velocity, retried, nvalid_pose, lm_n, lm_searched, pad;
  float velocity[16];
};

// per step and sequence (ps_tracker_fetch hands these out as ps_track_stat)
struct TrkStat {
  int32_t state;          // state after the frame
  int32_t tracked;        // the frame has a pose
  int32_t n;              // keypoints of the left image
  int32_t mm_matches;     // SearchByProjection(cur, last) of the attempt that was used
  int32_t retried;        // the 2 * th retry ran
  int32_t matches;        // after the first PoseOptimization and the outlier discard
  int32_t map_matches;
  int32_t lm_candidates;  // local-map points that passed isInFrustum
  int32_t lm_inliers;     // mnMatchesInliers of TrackLocalMap
  int32_t pad[3];
};

struct TrkCam {
  float fx, fy, cx, cy, mbf, mb, th_depth, gw_inv, gh_inv, log_sf, inv_fx, inv_fy;
  int32_t w, h, nlevels, pad;
  float sf[8], inv_sigma2[8];
};

struct TrkArrays {
  TrkCam cam;
  int32_t S, cap, kp_cap, max_steps;
  // extractor / stereo matcher outputs of the step (pointslot_hip.h: ps_orb_batch_device_outputs, ps_orb_stereo_device_outputs)
  const void* orb_kps; const uint8_t* orb_desc; const int32_t* orb_counts; const float* orb_uright; const float* orb_depth;
  // SLOT.MODE 4: the left 8-bit instance-id masks of the step (nullptr: none).  Frame::AssignFeatures keeps the keypoints on
  // background pixels (mask 0) as static features (Frame.cc:811-822)
  const uint8_t* idmask; int32_t mask_stride; size_t mask_pitch;
  TrkFrame cur, last;
  TrkSeq* seq;
  // the initial keyframe's map points = the local map of this slice
  float* lm_xw; float* lm_normal; float* lm_maxd; float* lm_mind; uint8_t* lm_desc;
  // SearchByProjection problems: motion model at th, its retry at 2 * th, local map
  PjProb* prob_mm1; PjProb* prob_mm2; PjProb* prob_lm;
  int32_t* nmatch_mm1; int32_t* nmatch_mm2; int32_t* nmatch_lm;
  uint8_t* qvalid; uint8_t* occupied; int32_t* match;
  float* qu; float* qv; float* qur; float* qrad; int32_t* qminl; int32_t* qmaxl;
  // PoseOptimization problems
  PoProb* po_prob; PoVertex* po_vert; float* po_obs; float* po_is2; double* po_pose; int32_t* po_result;
  // results
  float* traj;            // [max_steps][S][16] Tcw
  TrkStat* stats;         // [max_steps][S]
};
