/* pointslot_hip.h — C-ABI of libpointslot_hip.so: the MI355X (gfx950) implementation of the SLOT
 * front-end / back-end hot path of pkzhou/PointSLOT.
 *
 * The reference has no FFI layer: callers use the C++ classes ORB_SLAM2::ORBextractor, ORBmatcher
 * and Optimizer directly.  This header is the boundary a maintainer would
 * bind from those classes (the shim classes are in pointslot_amd/host/, the recipe in
 * INTEGRATION.md).  Every entry point cites the reference function it replaces.
 *
 * Conventions: plain C, POD structs, caller-owned buffers, no exceptions.  Every function returns
 * PS_OK (0) or a negative ps_status.  Handles are not re-entrant (like an ORBextractor instance,
 * which owns mvImagePyramid); distinct handles may be used from distinct threads concurrently.
 * Pointers named d_* are device (HBM) pointers, everything else is host memory.
 */
#ifndef POINTSLOT_HIP_H
#define POINTSLOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ps_status {
  PS_OK = 0,
  PS_ERR_INVALID = -1,   /* bad argument (null pointer, non-positive size, unsupported shape) */
  PS_ERR_HIP = -2,       /* a HIP runtime call failed; ps_last_error() has the text          */
  PS_ERR_CAPACITY = -3,  /* caller buffer or plan capacity too small                          */
  PS_ERR_NO_DEVICE = -4  /* no gfx950 device visible                                          */
} ps_status;

/* Binary-compatible with cv::KeyPoint (OpenCV 3.x): pt.x, pt.y, size, angle, response, octave,
 * class_id — 28 bytes.  Filled exactly as ORBextractor::operator() fills it
 * (/root/reference/src/ORBextractor.cc:837-852,1095-1101). */
typedef struct ps_keypoint {
  float x, y;
  float size;
  float angle;
  float response;
  int32_t octave;
  int32_t class_id;
} ps_keypoint;

const char* ps_last_error(void);
int ps_device_count(int* count);
const char* ps_version(void);
/* Test / diagnostic: fills the LDS of every CU of `device` with the 32-bit `pattern` (e.g. 0xFFFFFFFF: NaN when read as floating
 * point) and leaves it there.  LDS is not cleared between kernels, so a kernel that reads LDS it never wrote - or multiplies such
 * a slot by zero - then misbehaves deterministically instead of once in a while; the GPU tests call this before the kernels. */
int ps_debug_poison_lds(int device, uint32_t pattern);
/* Diagnostic: measured FP64 matrix-core rate of `device` in TFLOP/s (a loop of independent v_mfma_f64_16x16x4_f64 on every CU).
 * The local hardware guide has no FP64 MFMA peak; this is the denominator of the object-BA roofline. */
int ps_debug_mfma_f64_peak(int device, double* tflops);
/* Diagnostic: `repeats` launches of a streaming kernel that touches exactly `bytes` of a fresh buffer once per launch - mode 0 reads
 * 16 B per lane, 1 reads 4 B per lane, 2 writes 4 B per lane, 3 writes 16 B per lane (kernel names traffic_read / traffic_write).
 * Run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` it calibrates those counters for the access widths the kernels of this
 * library use (the hardware guide only calibrates 16-byte reads); tools/pmc_traffic.sh applies the factors. */
int ps_debug_traffic_kernel(int device, int mode, size_t bytes, int repeats);
/* Page-locked host memory for image and result buffers: copies from/to it are asynchronous and run at full PCIe rate
 * (ps_orb_extract_batch reads the caller's image buffers directly).  NULL on failure (ps_last_error has the text). */
void* ps_pinned_alloc(size_t bytes);
void ps_pinned_free(void* p);

/* ------------------------------------------------------------------------------------------------
 * ORB extractor — replaces ORB_SLAM2::ORBextractor (/root/reference/include/ORBextractor.h:51-85,
 * src/ORBextractor.cc:410-470 ctor, :1043-1105 operator()).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ps_orb ps_orb;

typedef struct ps_orb_config {
  int32_t nfeatures;      /* ORBextractor.nFeatures   (yaml: 1000; BASELINE config 2: 2000) */
  float scale_factor;     /* ORBextractor.scaleFactor (1.2)                                  */
  int32_t nlevels;        /* ORBextractor.nLevels     (8; 1..8 supported)                    */
  int32_t ini_th_fast;    /* ORBextractor.iniThFAST   (20)                                   */
  int32_t min_th_fast;    /* ORBextractor.minThFAST   (5)                                    */
  int32_t max_batch;      /* images processed per ps_orb_extract_batch_device call (>= 1)    */
  int32_t device;         /* HIP device ordinal                                              */
} ps_orb_config;

int ps_orb_create(const ps_orb_config* cfg, ps_orb** out);
void ps_orb_destroy(ps_orb* h);

/* Getters of ORBextractor.h:61-83.  Each array has nlevels entries; any pointer may be NULL. */
int ps_orb_get_tables(const ps_orb* h, float* scale_factors, float* inv_scale_factors,
                      float* level_sigma2, float* inv_level_sigma2, int32_t* features_per_level);

/* Level geometry for an image of w x h: level sizes (ORBextractor.cc:1111-1112).  The padded plane
 * of a level is (w_l + 38) x (h_l + 38) (EDGE_THRESHOLD = 19 on every side, :1113). */
int ps_orb_level_size(const ps_orb* h, int w, int hgt, int level, int32_t* w_l, int32_t* h_l);

/* ORBextractor::operator()(image, mask (ignored), keypoints, descriptors) for ONE host image.
 *   img/stride : CV_8UC1 image, `stride` bytes per row.  w <= 0 || hgt <= 0 || !img => *n = 0, PS_OK
 *                (the reference returns silently on an empty image, ORBextractor.cc:1046-1047).
 *   kps, desc  : caller buffers with room for `cap` keypoints / cap x 32 descriptor bytes.  The
 *                extractor may return up to nfeatures + 3 * nlevels keypoints (the quadtree stops
 *                at >= quota leaves per level, ORBextractor.cc:669-737).
 *   pyramid_out: NULL, or nlevels pointers; plane l receives the padded level image, tightly packed
 *                (w_l + 38) bytes per row — the contents of mvImagePyramid[l]'s parent buffer
 *                (ORBextractor.cc:1113-1128) that Frame::ComputeStereoMatches reads. */
int ps_orb_extract(ps_orb* h, const uint8_t* img, int w, int hgt, int stride, ps_keypoint* kps,
                   uint8_t* desc, int cap, int* n, uint8_t* const* pyramid_out);

/* The object features of a frame — Frame::ExtractObjORB -> OpencvORBDetector (/root/reference/src/Frame.cc:2623-2665):
 * cv::ORB::create(1000, 1.2, 8, 19)->detectAndCompute(im, ObjMask, kp, descriptor).  SURVEY.md 8f-2: OpenCV's own ORB (Harris
 * ranking, its own pyramid) is a different extractor; this entry point is the DECLARED STAND-IN - this library's pipeline on a
 * handle created with (1000, 1.2, 8, 20, 5), keypoints whose level-0 pixel lies outside `mask` (zero bytes) dropped before the
 * quadtree - so object keypoint sets differ from a true PointSLOT run.  mask == NULL: no mask. */
int ps_orb_extract_masked(ps_orb* h, const uint8_t* img, const uint8_t* mask, int w, int hgt, int stride, int mask_stride,
                          ps_keypoint* kps, uint8_t* desc, int cap, int* n);

/* ... and the detector itself: OpenCV 3.4.3's ORB_Impl::detectAndCompute (features2d/src/orb.cpp; firstLevel 0, WTA_K 2,
 * HARRIS_SCORE, patchSize 31) restated for the GPU - INTER_LINEAR_EXACT pyramid and mask pyramid (threshold 254 above level 0),
 * whole-image FAST-9/16 with non-maximum suppression, runByPixelsMask, runByImageBorder(edge_threshold), retainBest(2 N) by
 * FAST score, Harris responses (7 x 7, k = 0.04), retainBest(N), intensity-centroid angles, 7 x 7 sigma-2 blur, 256-bit rBRIEF.
 * Keypoints come out level by level in the order KeyPointsFilter::retainBest leaves them (std::nth_element / std::partition,
 * executed on the host with the same library calls).  UNVERIFIABLE against OpenCV in the build image, like every OpenCV stage of
 * the path; the CPU restatement it is tested against follows the same published source.
 * Frame.cc:2625: cv::ORB::create(1000, 1.2, 8, 19) == ps_cvorb_create(1000, 1.2f, 8, 19, 20, device, &h). */
typedef struct ps_cvorb ps_cvorb;
int ps_cvorb_create(int nfeatures, float scale_factor, int nlevels, int edge_threshold, int fast_threshold, int device, ps_cvorb** out);
void ps_cvorb_destroy(ps_cvorb* h);
/* detectAndCompute(image, mask, keypoints, descriptors); mask == NULL: no mask (non-zero mask bytes keep a keypoint). */
int ps_cvorb_detect_and_compute(ps_cvorb* h, const uint8_t* img, const uint8_t* mask, int w, int hgt, int stride, int mask_stride,
                                ps_keypoint* kps, uint8_t* desc, int cap, int* n);
/* Test access to intermediates of the last call.  what: 0 level image (tight w x h), 1 blurred level, 2 level mask, 3 the FAST
 * keypoints after the mask / border filters in raster order as float rows (x, y, score, Harris response), count in *n,
 * 4 the level size as int32[2]. */
int ps_cvorb_debug_read(ps_cvorb* h, int level, int what, void* out, size_t out_bytes, int* n);

/* Batched, device-resident form of ps_cvorb_detect_and_compute - Frame::ExtractObjORB for many frames at once: `nimg` images of one
 * size in HBM (image i at d_imgs + i * image_pitch) with their object masks (d_masks + i * mask_pitch, non-zero = keep), all
 * work queued on `stream` (a hipStream_t; NULL = the handle's stream) with nothing returning to the host: the two retainBest steps
 * run on the device with libstdc++'s own selection algorithms restated (csrc/retain_best.h), so the keypoints of an image are those
 * of the single-image call, order included.  Work is restricted to the parts of the pyramid a keypoint under the mask can touch.
 * Limits (reported as PS_ERR_CAPACITY by ps_cvorb_batch_fetch, never truncated silently): 2048 FAST keypoints under the mask per
 * level, 2048 keypoints per image.  Outputs stay in HBM: keypoints [nimg][capacity], descriptors [nimg][capacity][32], counts. */
int ps_cvorb_detect_batch_device(ps_cvorb* h, const uint8_t* d_imgs, const uint8_t* d_masks, int nimg, int w, int hgt, int stride,
                                 size_t image_pitch, int mask_stride, size_t mask_pitch, void* stream);
int ps_cvorb_batch_device_outputs(const ps_cvorb* h, const ps_keypoint** d_kps, const uint8_t** d_desc, const int32_t** d_counts,
                                  const int32_t** d_overflow, int32_t* capacity);
int ps_cvorb_batch_fetch(ps_cvorb* h, int image, ps_keypoint* kps, uint8_t* desc, int cap, int* n);

/* Batched, device-resident form of the same call: `nimg` images of identical size already in HBM
 * (image i at d_imgs + i * image_pitch, rows `stride` bytes apart).  Results stay in HBM inside the
 * handle; the call is asynchronous on `stream` (a hipStream_t, NULL = the handle's own stream). */
int ps_orb_extract_batch_device(ps_orb* h, const uint8_t* d_imgs, int nimg, int w, int hgt, int stride,
                                size_t image_pitch, void* stream);
/* Device views of the last batch: keypoints [nimg][kp_capacity], descriptors
 * [nimg][kp_capacity][32], counts [nimg]. */
int ps_orb_batch_device_outputs(const ps_orb* h, const ps_keypoint** d_kps, const uint8_t** d_desc,
                                const int32_t** d_counts, int32_t* kp_capacity);
/* Blocks until the batch is complete and copies image i's results to the host. */
int ps_orb_batch_fetch(ps_orb* h, int image, ps_keypoint* kps, uint8_t* desc, int cap, int* n);
/* Waits for all work queued on the handle. */
int ps_orb_sync(ps_orb* h);

/* The same batched extraction for images in HOST memory (nimg pointers, identical size, rows `stride` bytes apart): one
 * upload per image on the handle's stream, then the batch pipeline.  Asynchronous when the buffers come from
 * ps_pinned_alloc; results are read with ps_orb_batch_fetch / ps_orb_stereo_fetch_frames.  This is Frame::Frame's
 * ExtractORB(0, imLeft) + ExtractORB(1, imRight) (src/Frame.cc:709-712) for many frames at once: image 2k = left,
 * 2k+1 = right of frame k when ps_orb_stereo_match_batch follows. */
int ps_orb_extract_batch(ps_orb* h, const uint8_t* const* imgs, int nimg, int w, int hgt, int stride);

/* Frame::ComputeStereoMatches (/root/reference/src/Frame.cc:2142-2316) on extraction results that
 * are still in HBM: both padded pyramids (mvImagePyramid of the left and the right extractor), keypoints and descriptors.
 * mb = Frame::mb (baseline in metres), mbf = Frame::mbf.  Outputs are indexed like the LEFT image's keypoints:
 * u_right = mvuRight, depth = mvDepth, -1 where the keypoint has no stereo match.
 *   ps_orb_stereo_match_batch : the last batch of `h` holds left/right interleaved (image 2k = left, 2k+1 = right)
 *   ps_orb_stereo_match_pair  : the reference's layout — two extractor objects, one image each (Frame.cc:709-722) */
int ps_orb_stereo_match_batch(ps_orb* h, int npairs, float mb, float mbf);
int ps_orb_stereo_device_outputs(const ps_orb* h, const float** d_uright, const float** d_depth, const int32_t** d_kept);
int ps_orb_stereo_fetch(ps_orb* h, int pair, float* u_right, float* depth, int cap, int* n_left, int* kept);
int ps_orb_stereo_match_pair(ps_orb* left, ps_orb* right, float mb, float mbf, float* u_right, float* depth, int cap, int* n_left);
/* Everything a stereo Frame keeps from its two extractors and ComputeStereoMatches (mvKeys, mDescriptors, mvuRight,
 * mvDepth; src/Frame.cc:709-722), for the first `npairs` pairs of the last ps_orb_stereo_match_batch, in one
 * synchronisation and one pipelined transfer.  In: kps/desc/u_right/depth buffers with room for `cap` keypoints.
 * Out: n (left keypoints), n_right, kept (matches that survive the median cut). */
typedef struct ps_stereo_frame {
  ps_keypoint* kps; uint8_t* desc; float* u_right; float* depth;
  int32_t cap;
  int32_t n, n_right, kept;
} ps_stereo_frame;
int ps_orb_stereo_fetch_frames(ps_orb* h, ps_stereo_frame* frames, int npairs);
/* Frame::ComputeObjStereoMatches (src/Frame.cc:2318-2503): the same matcher on caller-provided key sets (the frame's object
 * features, mvTempObjKeys / mvTempObjKeysRight + descriptors, <= 4096 each) against the two extractors' device-resident
 * pyramids.  u_right / depth: n_left floats (-1 where unmatched); *kept (nullable) = matches that survive the median cut.
 * Keypoints must lie where the reference's 11x11 window + slide stays inside the level image (cv::ORB's edge threshold). */
int ps_orb_stereo_match_keys(ps_orb* left, ps_orb* right, const ps_keypoint* kps_l, const uint8_t* desc_l, int n_left,
                             const ps_keypoint* kps_r, const uint8_t* desc_r, int n_right, float mb, float mbf,
                             float* u_right, float* depth, int* kept);

/* Test/diagnostic access to intermediates of the last call (blocking).  `what`:
 *   0 padded plane (tight, (w_l+38) x (h_l+38) bytes)      1 blurred plane (tight, w_l x h_l)
 *   2 FAST candidates in reference emission order, int32 triples (x, y, score) relative to
 *     minBorder (ORBextractor.cc:822-824); returns the count in *n
 *   3 selected keypoints of the level, int32 triples (x, y, score) in level coordinates, in
 *     DistributeOctTree's output order; count in *n */
int ps_orb_debug_read(ps_orb* h, int image, int level, int what, void* out, size_t out_bytes, int* n);

/* Per-stage GPU time of ps_orb_extract_batch_device, measured with HIP events recorded on the stream
 * the kernels run on.  After ps_orb_enable_stage_timing(h, 1) every batch records one event set (a
 * ring of 64); ps_orb_stage_times synchronises and returns the mean over the recorded batches.
 * names/ms arrays of length `cap`; *n receives the stage count. */
int ps_orb_stage_times(ps_orb* h, const char** names, float* ms, int cap, int* n);
int ps_orb_enable_stage_timing(ps_orb* h, int enable);

/* ------------------------------------------------------------------------------------------------
 * Descriptor matching — replaces the hot members of ORB_SLAM2::ORBmatcher
 * (/root/reference/include/ORBmatcher.h:47-118, src/ORBmatcher.cc).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ps_matcher ps_matcher;
int ps_matcher_create(int device, ps_matcher** out);
void ps_matcher_destroy(ps_matcher* m);
/* GPU time of the kernels of the last ps_match_bruteforce / ps_search_by_projection call (HIP events on the handle's stream). */
int ps_matcher_last_kernel_ms(const ps_matcher* m, float* ms);

/* Bulk form of ORBmatcher::DescriptorDistance (ORBmatcher.cc:2704-2720): out[i * nt + j] = Hamming
 * distance between 32-byte descriptors q[i] and t[j] (0..256). */
int ps_hamming_matrix(ps_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out);

/* ORBmatcher::SearchByBruceMatching(LastFrame, CurrentFrame, nLastOrder, nCurrenOrder, matches)
 * (ORBmatcher.cc:2043-2155) for a batch of independent (object) problems.
 *   q_*   : the last frame's object features of one object: 32-byte descriptors
 *           (mvObjPointsDescriptors), angles of mvObjKeysUn, and q_valid[i] = 1 iff
 *           mvpMapObjectPoints[i] is non-null, not bad, and mvbObjKeysOutlier[i] is false (:2062-2063)
 *   t_*   : the current frame's features of the same object: descriptors, angles of mvObjKeys
 *   query_of_train[j] : out, index i of the query whose MapObjectPoint* the reference stores in
 *           vpMapObjectPointMatches[j], or -1 (NULL)
 *   nmatches : out, the function's return value
 * nn_ratio / check_orientation are the ORBmatcher constructor arguments (mfNNratio, mbCheckOrientation);
 * TH_LOW = 50 and HISTO_LENGTH = 30 are fixed as in ORBmatcher.cc:58-62. nt <= 4096. */
/* MapPoint / MapObjectPoint::ComputeDistinctiveDescriptors (src/MapObjectPoint.cc:379-436, src/MapPoint.cc:366; SURVEY.md
 * 8f-4) for a batch of points: point p owns the descriptor rows [off[p], off[p+1]) (its observations, <= 128);
 * best[p] = row index (relative to off[p]) of the descriptor with the least median distance to the others, -1 if none. */
int ps_distinctive_descriptors(ps_matcher* m, const uint8_t* desc, const int32_t* off, int npoints, int32_t* best);

typedef struct ps_bf_problem {
  const uint8_t* q_desc; const float* q_angle; const uint8_t* q_valid; int32_t nq;
  const uint8_t* t_desc; const float* t_angle; int32_t nt;
  int32_t* query_of_train;
  int32_t nmatches;
} ps_bf_problem;
int ps_match_bruteforce(ps_matcher* m, ps_bf_problem* probs, int nprob, float nn_ratio, int check_orientation);

/* The three ORBmatcher::SearchByProjection overloads as one windowed-matching call, batched:
 *   frame_mode = 1 : SearchByProjection(CurrentFrame, LastFrame, th, bMono)              (ORBmatcher.cc:1613-1756)
 *   frame_mode = 0 : SearchByProjection(F, vpMapPoints, th)                              (:68-155)  and
 *                    SearchByProjection(F, nOrder, vpMapObjectPoints, th) (use_bbox = 1) (:157-248)
 * `train` is the frame being matched INTO (CurrentFrame / F, or one object's feature set of it):
 *   x, y, octave, angle : mvKeysUn (mvObjKeysUn[nOrder]);  u_right : mvuRight;  desc : mDescriptors rows
 *   occupied[j] : 1 iff mvpMapPoints[j] is non-null AND has Observations() > 0 (the reference's skip test)
 *   in_bbox[j]  : Frame::isInBBox(nOrder, x, y) (object variant only)
 *   cell_off / cell_idx : mGrid as CSR, cell = ix * 48 + iy (FRAME_GRID_ROWS), indices in insertion order
 *   min_x, min_y, grid_w_inv, grid_h_inv : mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv
 * Queries are the points being projected, in the reference's loop order:
 *   q_valid    : frame mode: mvpMapPoints[i] && !mvbOutlier[i] of the LAST frame; else mbTrackInView && !isBad()
 *   q_desc     : pMP->GetDescriptor();   q_observed[i] : pMP->Observations() > 0
 *   frame mode : q_xw = GetWorldPos, q_octave = LastFrame.mvKeys[i].octave, q_angle = LastFrame.mvKeysUn[i].angle,
 *                tcw / tlw = CurrentFrame.mTcw / LastFrame.mTcw, fx..mb, bounds = mnMinX,mnMaxX,mnMinY,mnMaxY,
 *                scale_factors = mvScaleFactors, th, mono
 *   otherwise  : q_u, q_v, q_ur = mTrackProjX, mTrackProjY, mTrackProjXR; q_radius = window half-size
 *                (r * scaleFactor[level], or 5 for objects); q_radius_er = r * scaleFactor[level] (uR gate);
 *                q_min_level / q_max_level = level-1 / level (level+1 for objects)
 * th_dist = TH_HIGH (100) or TH_HIGH_FORDYNAMIC (130); ratio_test/nn_ratio = the same-level mfNNratio test;
 * check_orientation = mbCheckOrientation (frame mode).
 *   match_of_train[j] : out, index of the query assigned to slot j; -1 where the call leaves the slot alone (the caller keeps
 *                       its pointer); -2 where the call assigned the slot and the rotation check then reset it to NULL
 *                       (ORBmatcher.cc:1742-1750: the caller writes NULL)
 *   nmatches          : out, the return value
 * A search window may hold any number of candidates in the reference.  Here a window of more than 256 makes its problem run a
 * second time with a wider candidate store (1024 per window, frames of up to 8191 features); only a problem that exceeds that as
 * well fails - nmatches = -1, match_of_train untouched - and the call returns PS_ERR_CAPACITY after serving all other problems. */
typedef struct ps_proj_train {
  int32_t n;
  const float* x; const float* y; const int32_t* octave; const float* angle; const float* u_right; const uint8_t* desc;
  const uint8_t* occupied; const uint8_t* in_bbox;
  const int32_t* cell_off; const int32_t* cell_idx;
  float min_x, min_y, grid_w_inv, grid_h_inv;
} ps_proj_train;
typedef struct ps_proj_problem {
  ps_proj_train train;
  int32_t nq;
  const uint8_t* q_valid; const uint8_t* q_desc; const uint8_t* q_observed; const float* q_angle;
  const float* q_u; const float* q_v; const float* q_ur; const float* q_radius; const float* q_radius_er;
  const int32_t* q_min_level; const int32_t* q_max_level;
  const float* q_xw; const int32_t* q_octave;
  int32_t frame_mode, mono;
  float tcw[16], tlw[16];
  float fx, fy, cx, cy, mbf, mb;
  float bounds[4];
  float scale_factors[8];
  float th;
  int32_t th_dist, ratio_test; float nn_ratio; int32_t check_orientation, use_bbox;
  int32_t* match_of_train;
  int32_t nmatches;
} ps_proj_problem;
int ps_search_by_projection(ps_matcher* m, ps_proj_problem* probs, int nprob);

/* The search half of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:982-1136) and
 * Fuse(ObjectKeyFrame*, const vector<MapObjectPoint*>&, th) (:1138-1260), SURVEY.md 8f-4: for every candidate point the
 * projection into the keyframe, the depth / image-or-box / scale-invariance / viewing-angle gates, PredictScale,
 * KeyFrame::GetFeaturesInArea(u, v, th * scale[level]) and the chi-square-gated (7.8 stereo / 5.99 mono) best Hamming match.
 * best_idx[i] = keyframe feature to fuse with (bestDist <= TH_LOW), -1 otherwise; best_dist[i] = bestDist (256 if none).
 * What the reference does next (Replace / AddObservation, :1113-1131 / :1239-1255) is map surgery and stays with the caller,
 * in candidate order.  train: the keyframe's mvKeysUn / mvObjKeysUn (x, y, octave), mvuRight, descriptors and feature grid
 * (angle / occupied / in_bbox unused).  q_valid[i] = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF); q_pos = GetWorldPos()
 * or GetInObjFramePosition(); q_min_dist / q_max_dist = mfMinDistance / mfMaxDistance (the 0.8 / 1.2 factors are applied
 * inside).  bounds = {minX, maxX, minY, maxY}: IsInImage's mnMin/MaxX/Y or IsInBBox's detection box (doubles, as there). */
typedef struct ps_fuse_problem {
  ps_proj_train train;
  int32_t nq;
  const uint8_t* q_valid;
  const float* q_pos;        /* [nq][3] */
  const float* q_normal;     /* [nq][3] GetNormal() */
  const float* q_min_dist;
  const float* q_max_dist;
  const uint8_t* q_desc;     /* [nq][32] GetDescriptor() */
  float rcw[9], tcw[3], ow[3];   /* GetRotation() row-major, GetTranslation(), GetCameraCenter() */
  float fx, fy, cx, cy, bf;
  double bounds[4];
  float scale_factors[8], inv_level_sigma2[8];
  float log_scale_factor;    /* mfLogScaleFactor */
  int32_t n_levels;          /* mnScaleLevels */
  float th;
  int32_t* best_idx;         /* out [nq] */
  int32_t* best_dist;        /* out [nq] */
} ps_fuse_problem;
int ps_fuse_search(ps_matcher* m, ps_fuse_problem* problems, int nproblems);


/* ------------------------------------------------------------------------------------------------
 * Optimiser — replaces the hot static members of ORB_SLAM2::Optimizer
 * (/root/reference/include/Optimizer.h:51-61, src/Optimizer.cc:249-1075) and the g2o solver stack they
 * instantiate.  All arithmetic is FP64 on float32 inputs, as in the reference.
 * SE3 poses cross this boundary either as the float 4x4 row-major matrix the reference keeps in
 * cv::Mat (Frame::mTcw) or as 7 doubles (tx,ty,tz,qx,qy,qz,qw) = g2o::SE3Quat::toVector().
 * ---------------------------------------------------------------------------------------------- */
typedef struct ps_optimizer ps_optimizer;
int ps_optimizer_create(int device, ps_optimizer** out);
void ps_optimizer_destroy(ps_optimizer* m);
/* GPU time of the last batch's kernel(s), from HIP events on the handle's stream. */
int ps_optimizer_last_kernel_ms(const ps_optimizer* m, float* ms);
/* Optional per-LM-iteration log (chi2, lambda, trials) of the next batches; for parity tests. */
int ps_optimizer_enable_trace(ps_optimizer* m, int enable);
int ps_optimizer_get_trace(const ps_optimizer* m, int problem, double* chi2_lambda_trials, int cap, int* n);

/* Converter::toSE3Quat / Converter::toCvMat (src/Converter.cc:37-71) for callers without Eigen. */
int ps_se3_from_mat4f(const float* m16, double* pose7);
int ps_se3_to_mat4f(const double* pose7, float* m16);

/* Optimizer::PoseOptimization(Frame*) (Optimizer.cc:249-477) for a batch of independent frames.
 *   n          : pFrame->N
 *   xw         : [n][3] world position of mvpMapPoints[i] (float, MapPoint::GetWorldPos)
 *   obs        : [n][3] mvKeysUn[i].pt.x, .pt.y, mvuRight[i] (uR < 0 => monocular edge)
 *   inv_sigma2 : [n]    mvInvLevelSigma2[mvKeysUn[i].octave]
 *   valid      : [n]    1 iff mvpMapPoints[i] != NULL
 *   fx..bf     : Frame::fx, fy, cx, cy, mbf
 *   tcw        : in: pFrame->mTcw; out: the pose passed to pFrame->SetPose
 *   outlier    : [n] in/out pFrame->mvbOutlier (entries with valid == 0 are left untouched)
 *   result     : out, the return value: nInitialCorrespondences - nBad, 0 when < 15 correspondences */
typedef struct ps_pose_problem {
  int32_t n;
  const float* xw; const float* obs; const float* inv_sigma2; const uint8_t* valid;
  float fx, fy, cx, cy, bf;
  float tcw[16];
  uint8_t* outlier;
  int32_t result;
} ps_pose_problem;
int ps_pose_optimize_batch(ps_optimizer* m, ps_pose_problem* probs, int nprob);

/* Optimizer::CFSE3ObjStateOptimization(Frame*, vnNeedToBeOptimized, verbose) (Optimizer.cc:479-753):
 * k objects of one frame optimised as ONE graph (k <= 16).  Object o owns the feature range
 * [off[o], off[o+1]) of the concatenated arrays.
 *   xo         : MapObjectPoint::GetInObjFramePosition (object-frame coordinates)
 *   obs        : mvObjKeysUn[o][j].pt.x, .pt.y, mvuObjKeysRight[o][j]
 *   valid      : 1 iff mvpMapObjectPoints[o][j] != NULL
 *   poses7     : in: MapObject::GetCFInFrameObjState(frame).pose (Tco); out: the optimised vertex
 *                estimate.  The translation of the input is also the measurement of the
 *                EdgeTransConstraintFromDetction prior (information 50 I).
 *   outlier    : in/out mvbObjKeysOutlier
 *   result     : out, 1 (true) / 0 (false: no object or fewer than 15 edges) */
typedef struct ps_cfse3_problem {
  int32_t k;
  const int32_t* off;
  const float* xo; const float* obs; const float* inv_sigma2; const uint8_t* valid;
  float fx, fy, cx, cy, bf;
  double* poses7;
  uint8_t* outlier;
  int32_t result;
} ps_cfse3_problem;
int ps_cfse3_optimize_batch(ps_optimizer* m, ps_cfse3_problem* probs, int nprob);

/* Optimizer::ObjectLocalBundleAdjustment(ObjectKeyFrame*, verbose) (Optimizer.cc:755-1075) on a graph the
 * caller has already collected (:755-818, :827-951 map 1:1 onto these arrays), for a batch of objects.
 *   poses7     : [np][7] in: Converter::toSE3Quat(pKFi->GetPose()) of the local (free) and fixed object
 *                keyframes; out: the optimised estimates (written back with pKF->SetPose, :1033-1064)
 *   pose_flags : [np] bit 0 = setFixed(true) (mnId == 0 or a lFixedCameras entry), bit 1 = VertexSE3Fix with
 *                whether_fixrollpitch (every local keyframe, :834-846); at most 128 free poses
 *   points     : [nl][3] in: MapObjectPoint::GetInObjFrameEigenPosition; out: SetInObjFramePosition (:1066-1074)
 *   e_*        : one entry per (point, observing keyframe): vertex indices, (u, v, uR) with uR < 0 for a
 *                monocular edge, mvInvLevelSigma2[octave]; at most one edge per (pose, point) pair
 *   erase      : [ne] out, 1 where the reference queues (pKFi, pMP) into vToErase (:988-1012)
 *   iterations / trials : out, LM iterations and damping trials executed (optimize(5) + optimize(10))
 *   trace      : NULL or room for 40 x 3 doubles: (chi2, lambda, trials) per LM iteration; n_trace = count */
typedef struct ps_ba_problem {
  int32_t np, nl, ne;
  double* poses7; const uint8_t* pose_flags;
  double* points;
  const int32_t* e_pose; const int32_t* e_point; const float* e_obs; const float* e_inv_sigma2;
  float fx, fy, cx, cy, bf;
  uint8_t* erase;
  int32_t n_erased, iterations, trials, n_trace;
  double* trace;
} ps_ba_problem;
int ps_object_ba_batch(ps_optimizer* m, ps_ba_problem* probs, int nprob);
/* The reprojection test of Tracking::DynamicStaticDiscrimination (src/Tracking.cc:2099-2181) for a batch of
 * tracked detections: every object point is moved as if the object were static, Pc = Tcw_cur * Tcw_last^-1 * (Tco_last * Po),
 * and its chi-square against the current observation is collected per kind (monocular: u_right < 0 / stereo); each list is
 * sorted, values above 5 x median dropped, the rest averaged in sorted order (FP64, bit-identical to the CPU statement).
 * The caller keeps the gates around it (depth range, image prior, :2082-2097) and feeds the averages to
 * DetectionObject::SetDynamicFlag(mono_avg, stereo_avg).  Poses are (tx, ty, tz, qx, qy, qz, qw). */
typedef struct ps_dyn_problem {
  int32_t n;                      /* mvpMapObjectPoints[order].size(), <= 2048 */
  const uint8_t* valid;           /* vMOPs[j] != NULL */
  const double* po;               /* [n][3] GetInObjFrameEigenPosition() */
  const float* obs;               /* [n][3] mvObjKeysUn[order][j].pt.x, .pt.y, mvuObjKeysRight[order][j] */
  const float* inv_sigma2;        /* [n] mvInvLevelSigma2[octave] */
  double last_tco[7];             /* pMO->GetCFInFrameObjState(mLastFrame.mnId).pose */
  double last_tcw[7], cur_tcw[7]; /* mLastFrame.mSETcw, mCurrentFrame.mSETcw */
  double fx, fy, cx, cy;          /* mdCamProjMatrix */
  float mbf;
  double mono_avg, stereo_avg;    /* out: monoDynaValAvg, stereoDynaValAvg (0 with fewer than 5 points of the kind) */
  int32_t mono_n, stereo_n;       /* out: monoPointNum, stereoPointNum after the rejection */
} ps_dyn_problem;
int ps_dynamic_discrimination_batch(ps_optimizer* h, ps_dyn_problem* problems, int nproblems);

/* Optimizer::LocalBundleAdjustment(KeyFrame*, pbStopFlag, Map*) (Optimizer.cc:1077-1417) on a collected
 * graph: same edge types and LM schedule with plain VertexSE3Expmap keyframes (pose_flags bit 1 = 0; bit 0 = fixed for
 * mnId == 0 and the fixed cameras) and world-frame map points.  The abort flag of the reference is not modelled. */
int ps_local_ba_batch(ps_optimizer* m, ps_ba_problem* probs, int nprob);

/* ------------------------------------------------------------------------------------------------
 * Lockstep tracker — the tracking thread's per-frame chain for many independent stereo sequences, resident on the device:
 *   Frame::Frame (ExtractORB x 2, ComputeStereoMatches, AssignFeaturesToGrid)      /root/reference/src/Frame.cc:709-722,1636-1656
 *   StereoInitialization / UpdateLastFrame / TrackWithMotionModel / TrackLocalMap   src/Tracking.cc:2840-3160
 *   (SearchByProjection(cur, last, th) with its 2 th retry, PoseOptimization, isInFrustum, SearchByProjection(F, points),
 *   PoseOptimization) and the constant-velocity model, src/Tracking.cc:1260-1286
 * in localisation mode (mbOnlyTracking), exactly the slice pointslot_amd/host/StereoOdometry.h drives through the calls
 * above — but queued on ONE stream per step with nothing returning to the host: no packing, no PCIe traffic besides the
 * images.  Independent sequences are the unit of parallelism (BASELINE config 4 / SURVEY.md 8e).  Results are identical to
 * the per-call driver's.  A handle is not re-entrant; several handles may run on several threads / streams side by side.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ps_tracker ps_tracker;
typedef struct ps_tracker_config {
  int32_t n_sequences;               /* sequences advanced by one call                                         */
  int32_t width, height;             /* image size (all sequences share the rig)                               */
  float fx, fy, cx, cy, bf;          /* Camera.fx .. Camera.bf of the settings file                            */
  float th_depth;                    /* ThDepth (35 in the KITTI settings)                                     */
  int32_t nfeatures; float scale_factor; int32_t nlevels, ini_th_fast, min_th_fast;   /* ORBextractor.*      */
  int32_t max_steps;                 /* frames per sequence the handle keeps results for                       */
  int32_t device;
  int32_t max_objects;               /* detections per frame (SLOT.MODE 4 object chain; 0 = camera only, <= 16) */
  int32_t max_map_objects;           /* MapObjects a sequence can hold over its life (Tracking's AllObjects; the reference's list is
                                        unbounded, the device table is not): 0 = 8, <= 64.  A detection that would need one more is
                                        ignored and ps_tracker_fetch_objects reports PS_ERR_CAPACITY                            */
} ps_tracker_config;
/* what Tracking::Track leaves per frame and sequence */
typedef struct ps_track_stat {
  int32_t state;          /* 0 NOT_INITIALIZED, 1 OK, 2 LOST after the frame                                   */
  int32_t tracked;        /* the frame has a pose                                                              */
  int32_t n;              /* keypoints of the left image                                                       */
  int32_t mm_matches;     /* SearchByProjection(cur, last): matches of the attempt that was used               */
  int32_t retried;        /* the 2 * th retry ran (Tracking.cc:3042-3048)                                      */
  int32_t matches;        /* after the first PoseOptimization and the outlier discard (Tracking.cc:3062-3082)  */
  int32_t map_matches;    /* ... of which on map points with observations                                      */
  int32_t lm_candidates;  /* local-map points that passed Frame::isInFrustum                                   */
  int32_t lm_inliers;     /* mnMatchesInliers of TrackLocalMap                                                 */
  int32_t overflowed;     /* search windows of this frame that held more than 256 candidates: the matcher's candidate store is
                             bounded here (the reference's GetFeaturesInArea is not), candidates beyond it were not considered and
                             the frame's result may differ from the reference's; the other sequences are unaffected            */
  int32_t reserved[2];
} ps_track_stat;
int ps_tracker_create(const ps_tracker_config* cfg, ps_tracker** out);
void ps_tracker_destroy(ps_tracker* t);
/* One stereo frame of every sequence.  _device: images already in HBM, sequence k's left image at d_imgs + 2k * image_pitch,
 * its right image one pitch further, rows `stride` bytes apart.  The host form takes one pointer per image (asynchronous when
 * they come from ps_pinned_alloc).  Both return as soon as the step is queued. */
int ps_tracker_step_device(ps_tracker* t, const uint8_t* d_imgs, int stride, size_t image_pitch);
int ps_tracker_step(ps_tracker* t, const uint8_t* const* left, const uint8_t* const* right, int stride);
/* SLOT.MODE 4 (offline detections + instance masks), the whole of Tracking::Track's per-frame work: the camera chain above on the
 * background keypoints (Frame::AssignFeatures, src/Frame.cc:762-977: keypoints on mask != 0 leave the static set) and behind it
 * the object chain on the device -
 *   Frame::ExtractObjORB (cv::ORB(1000, 1.2, 8, 19) under the object masks, :2623-2665) and ComputeObjStereoMatches (:2318-2503),
 *   TrackMapObject (pose prediction, InitializeCurrentObjPose's RANSAC centroid, FineTuningUsing2dBox, MapObjectInit;
 *   src/Tracking.cc:1533-1930), TrackLastFrameObjectPoint (temporal points, SearchByBruceMatching, CFSE3ObjStateOptimization,
 *   :2288-2466), TrackObjectLocalMap (isInFrustum(pMP, nOrder), SearchByProjection(F, nOrder, MOPs), CFSE3, :2468-2712), the end of
 *   Track (MapObjectReInit for an object whose tracking failed, :1443-1478,1932-2031)
 * - in the localisation-mode slice pointslot_amd/object_tracker.py documents (an object's local map is the object keyframe of its
 * (re-)initialisation).  d_masks: per sequence the LEFT 8-bit id mask of Frame::ReadKittiSegmentationImage (0 background, 255
 * ignored, instance + 1), sequence k at d_masks + k * mask_pitch; d_dets: [n_sequences][max_objects] detections of the frame in
 * label order, unused slots with id < 0 behind the used ones.  Both in HBM; they must stay unchanged until the step has run.
 * (Environment, read at ps_tracker_create: PS_TRK_OVERLAP=1 runs ExtractObjORB on a second stream beside the camera chain - same
 * results; PS_TRK_DEBUG_SYNC=1 waits after every launch group and reports it on stderr.) */
typedef struct ps_detection {
  int32_t id;            /* DetectionObject::mnObjectID (the label's track id); < 0: empty slot                               */
  int32_t bbox[4];       /* mrectBBox: x, y, width, height (cv::Rect of the label's truncated doubles)                         */
  int32_t reserved[3];
  double scale[3];       /* mScale: length, height, width                                                                      */
  double pose7[7];       /* mTruthPosInCameraFrame.pose (tx, ty, tz, qx, qy, qz, qw): fromMinimalVector(X, Y - h / 2, Z, 0, ry, 0) */
} ps_detection;
/* what the object chain leaves per frame, sequence and detection slot */
typedef struct ps_object_stat {
  int32_t id;             /* the detection's id, -1 for an empty slot                                                          */
  int32_t n, stereo;      /* object features of the detection / of which with depth                                            */
  int32_t tracked;        /* the detection has a MapObject after the frame                                                     */
  int32_t is_new;         /* MapObjectInit ran in this frame                                                                   */
  int32_t track_ok;       /* DetectionObject::mbTrackOK at the end of TrackObjectLocalMap                                      */
  int32_t inliers;        /* mnMatchesInliers                                                                                  */
  int32_t bf_matches;     /* SearchByBruceMatching's return value                                                              */
  int32_t lm_candidates;  /* local points that passed isInFrustum                                                              */
  int32_t lm_matches;     /* SearchByProjection(F, nOrder, MOPs)'s return value                                                */
  int32_t map_points;     /* the frame's MapObjectPoints with observations when the frame was finished                         */
  int32_t reinit;         /* MapObjectReInit ran                                                                               */
  int32_t dynamic;        /* DetectionObject::GetDynamicFlag() after Tracking::DynamicStaticDiscrimination (Tracking.cc:2058)  */
  int32_t mo_dynamic;     /* its MapObject's GetDynamicFlag(), -1 without a MapObject                                          */
  int32_t dyn_n_mono, dyn_n_stereo;   /* points the two averages below were taken over (after the 5 x median rejection)        */
  double tco[7];          /* GetCFInFrameObjState(frame).pose when the frame was finished                                      */
  double dyn_mono, dyn_stereo;        /* mdMonoDynaVal / mdStereoDynaVal: mean chi2 of "the object did not move" (0: not run)  */
} ps_object_stat;
int ps_tracker_step_slot_device(ps_tracker* t, const uint8_t* d_imgs, int stride, size_t image_pitch, const uint8_t* d_masks, int mask_stride,
                                size_t mask_pitch, const ps_detection* d_dets);
/* Blocks, then copies the object results of steps [first_step, first_step + nsteps): [nsteps][n_sequences][max_objects]. */
int ps_tracker_fetch_objects(ps_tracker* t, int first_step, int nsteps, ps_object_stat* out);
int ps_tracker_sync(ps_tracker* t);
int ps_tracker_steps(const ps_tracker* t, int* steps);
/* Blocks, then copies the results of steps [first_step, first_step + nsteps): tcw [nsteps][n_sequences][16] (mTcw row-major,
 * all zero for a frame without a pose) and stats [nsteps][n_sequences]; either pointer may be NULL.  A frame whose search windows
 * overflowed the candidate store says so in its own ps_track_stat::overflowed; the call itself succeeds. */
int ps_tracker_fetch(ps_tracker* t, int first_step, int nsteps, float* tcw, ps_track_stat* stats);
/* Test hook: the next queued step reports `count` overflowed windows for sequence `seq`. */
int ps_tracker_debug_set_overflow(ps_tracker* t, int seq, int count);
/* All sequences back to NOT_INITIALIZED, step counter 0. */
int ps_tracker_reset(ps_tracker* t);
/* GPU time per stage of a step (HIP events on the tracker's stream, mean over the recorded steps, at most 64):
 * orb_extract, stereo_match, track_glue, search_by_projection, pose_optimization, and with objects: object_features,
 * object_stereo_match, object_glue, object_bruteforce, object_cfse3, object_search_by_projection. */
int ps_tracker_enable_stage_timing(ps_tracker* t, int enable);
int ps_tracker_stage_times(ps_tracker* t, const char** names, float* ms, int cap, int* n);
/* The extractor the tracker owns (its per-kernel stage times, debug reads). */
int ps_tracker_orb(ps_tracker* t, ps_orb** orb);

#ifdef __cplusplus
}
#endif
#endif /* POINTSLOT_HIP_H */
