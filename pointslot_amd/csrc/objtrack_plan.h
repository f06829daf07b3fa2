// Device-side state of the object half of the lockstep tracker (objtrack_kernels.hip / track_host.hip): per sequence the
// detections of the current and the last frame with their object features, and the sequence's MapObjects - the data the chain
//   ExtractObjORB -> ComputeObjStereoMatches -> AssignFeatures -> TrackMapObject -> TrackLastFrameObjectPoint -> TrackObjectLocalMap
// (/root/reference/src/Frame.cc:690-733,762-977; src/Tracking.cc:1224-1233,1443-1478,1533-2031,2288-2712) works on.
// pointslot_amd/object_tracker.py is the per-call twin and documents the slice.
#pragma once
#include <stdint.h>
#include "match_plan.h"
#include "opt_plan.h"

#define OB_MAXK 16       // detections per frame and sequence (PS_PO_MAX_K: one CFSE3 graph holds them all)
#define OB_MAXM 64       // MapObjects per sequence (AllObjects): upper bound of ps_tracker_config.max_map_objects (default 8)
#define OB_NCELL (PS_GRID_COLS * PS_GRID_ROWS)

// = ps_detection of pointslot_hip.h
struct ObDet { int32_t id; int32_t bbox[4]; int32_t pad[3]; double scale[3]; double pose7[7]; };

// what a Frame keeps per detection; the features of all detections of a sequence share one [S][OC] array, sorted by detection
struct ObFrame {
  float* x; float* y; float* angle; float* uright; float* depth; int32_t* octave;
  uint8_t* desc;                                    // [S][OC][32] mvObjPointsDescriptors
  uint8_t* mp_valid; uint8_t* mp_observed; uint8_t* outlier; int32_t* mp_id; float* mp_po;   // mvpMapObjectPoints (po: [S][OC][3])
  int32_t* off;                                     // [S][K + 1] feature range of detection j
  ObDet* det;                                       // [S][K] mvDetectionObjects
  int32_t* mo;                                      // [S][K] slot of mvMapObjects[j] in the sequence's table, -1 = NULL
  double* tco;                                      // [S][K][7] GetCFInFrameObjState(frame).pose when the frame was finished
  int32_t* cell_off;                                // [S][K][OB_NCELL + 1] mvObjKeysGrid as CSR
  int32_t* cell_idx;                                // [S][OC], indices relative to the detection's first feature
  int32_t* ndet;                                    // [S]
};

struct ObMapObject {
  int32_t id;                 // mnTruthID, -1 = free slot
  int32_t first_frame;        // mnFirstObservationFrameId
  int32_t kf_frame;           // mnLastKeyFrameId (the frame of the object keyframe that is its local map)
  int32_t tco_frame;          // latest frame with a camera-frame state
  int32_t local_valid;        // mvLocalObjectKeyFrames holds the keyframe
  int32_t npts;               // points of the keyframe (local map)
  int32_t dyn;                // bit 0 mbDynamicFlag, bit 1 mbFirstObserved, bit 2 mbDynamicChanged, bits 4-7 mqbHistoricalDynafFlag
                              // (oldest entry in bit 4), bits 8-10 its length (MapObject.cc:414-448)
  int32_t pad;
  double scale[3];
  double tco[7];
};

// per step, sequence and detection (ps_tracker_fetch_objects hands these out as ps_object_stat)
struct ObStat {
  int32_t id, n, stereo, tracked, is_new, track_ok, inliers, bf_matches, lm_candidates, lm_matches, map_points, reinit;
  int32_t dynamic, mo_dynamic, dyn_n_mono, dyn_n_stereo;   // DynamicStaticDiscrimination: the detection's / its MapObject's flag (-1: none), points averaged
  double tco[7];
  double dyn_mono, dyn_stereo;                             // DetectionObject::mdMonoDynaVal / mdStereoDynaVal
};

struct ObCam { float fx, fy, cx, cy, mbf, mb, th_depth, gw_inv, gh_inv, log_sf, inv_fx, inv_fy; int32_t w, h, nlevels, pad; float sf[8], inv_sigma2[8]; };

struct ObArrays {
  ObCam cam;
  int32_t S, K, M, OC, LC, max_steps;
  // inputs of the step
  const uint8_t* idmask; int32_t mask_stride; size_t mask_pitch;     // [S] left 8-bit id masks
  const ObDet* dets_in;                                              // [S][K]
  const void* cv_kps; const uint8_t* cv_desc; const int32_t* cv_count; int32_t cv_cap;   // cv::ORB of image 2s (left)
  const float* st_uright; const float* st_depth;                     // [S][OC] ComputeObjStereoMatches of the temp keys
  const float* cam_traj; const int32_t* cam_stats; int32_t cam_stat_words;   // the camera tracker's results [max_steps][S]
  ObFrame cur, last;
  ObMapObject* mobj;                                                 // [S][M]
  float* lm_po; float* lm_normal; float* lm_maxd; float* lm_mind; uint8_t* lm_desc;   // [S][M][LC] the keyframe's points
  // per-frame work
  int8_t* owner;                                                     // [S][cv_cap]
  int32_t* in_last; int32_t* tracked; int32_t* need; int32_t* track_ok; int32_t* inl_flag;   // [S][K]; inl_flag [S][OC]
  double* cam_pts;                                                   // [S][OC][3] camera-frame points of a detection (RANSAC)
  double* last_tco;                                                  // [S][K][7] the last frame's Tco of the detection's track (slot in_last), copied by ob_track:
                                                                     // ob_finish's blocks overwrite last.tco while others still run the discrimination test
  // brute-force matcher
  BfProb* bf_prob; uint32_t* bf_topk; uint8_t* bf_qvalid; int32_t* bf_qot; int32_t* bf_nmatch;
  // windowed matcher (queries: the local map of the detection's MapObject, [S][M][LC])
  PjProb* pj_prob; uint8_t* pj_qvalid; float* pj_qu; float* pj_qv; float* pj_qur; float* pj_qrad; float* pj_qrer; int32_t* pj_qminl; int32_t* pj_qmaxl;
  uint8_t* occupied; uint8_t* inbbox; int32_t* pj_match; int32_t* pj_nmatch;
  // CFSE3
  PoProb* po_prob; PoVertex* po_vert; float* po_obs; float* po_is2; double* po_pose; int32_t* po_result; int32_t* po_vmap;
  // results
  ObStat* stats;                                                     // [max_steps][S][K]
  int32_t* dropped;                                                  // [S] detections ignored because the MapObject table was full
  // capacity events, sticky per sequence until ps_tracker_reset (the sources are cleared / reused by every step):
  const int32_t* cv_overflow;                                        // [2 S] the batched cv::ORB's flags of THIS step (images 2s, 2s + 1)
  int32_t* pj_overflow;                                              // [S][K] overflowed windows of this step's object searches
  int32_t* det_overflow;                                             // [S] steps in which the object detector hit a keypoint capacity
  int32_t* search_overflow;                                          // [S] object search windows beyond the candidate store, all steps
};
