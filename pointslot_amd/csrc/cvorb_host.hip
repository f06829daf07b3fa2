// Host side of the object-feature detector (include/pointslot_hip.h: ps_cvorb_*): OpenCV's own ORB as the reference calls it
//   cv::ORB::create(1000, 1.2, 8, 19)->detectAndCompute(im, ObjMask, kp, descriptor)      /root/reference/src/Frame.cc:2623-2627
// restated from OpenCV 3.4.3 (features2d/src/orb.cpp) - SURVEY.md 8f-2; unverifiable against OpenCV in this image.  Plan
// (level sizes, INTER_LINEAR_EXACT coefficient tables), kernel orchestration, and the two KeyPointsFilter::retainBest steps,
// which run here with std::nth_element / std::partition because their output order is whatever those algorithms leave.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/pointslot_hip.h"
#include "cvorb_plan.h"
#include "ps_common.h"

extern "C" {
void psk_cv_level0(const CvLevelDev*, const uint8_t*, int, const uint8_t*, int, hipStream_t);
void psk_cv_resize(const CvLevelDev*, const CvLevelDev*, const int4*, const int4*, hipStream_t);
void psk_cv_detect(const CvLevelDev*, int, int, int, int32_t*, hipStream_t);
void psk_cv_blur(const CvLevelDev*, const int*, hipStream_t);
void psk_cv_describe(const CvPlanDev*, const CvSel*, int, void*, uint8_t*, hipStream_t);
void psk_cvb_run(const CvbPlan*, int, const uint8_t*, int, size_t, const uint8_t*, int, size_t, int, hipStream_t);
}

struct ps_cvorb {
  int nfeatures, nlevels, edge, fast_th, device;
  double scale_factor;
  hipStream_t stream = nullptr;
  int w = 0, h = 0;
  bool planned = false;
  CvPlanDev plan;                      // device pointers inside
  std::vector<int> cap;                // candidate capacity per level
  std::vector<int> quota;              // nfeaturesPerLevel
  uint8_t* d_buf = nullptr;            // one arena: planes, tables, candidates, selections, outputs
  uint8_t* d_img = nullptr; uint8_t* d_mask = nullptr;   // staged inputs
  int4* d_xtab[CV_MAX_LEVELS] = {}; int4* d_ytab[CV_MAX_LEVELS] = {};
  int32_t* d_total = nullptr;          // [nlevels]
  CvSel* d_sel = nullptr; ps_keypoint* d_kps = nullptr; uint8_t* d_desc = nullptr;
  int sel_cap = 0;
  int kq[7];
  // batched, device-resident form (ps_cvorb_detect_batch_device)
  CvbPlan bplan; bool bplanned = false; int bw = 0, bh = 0, bcap = 0;
  uint8_t* b_buf = nullptr;            // one arena: planes of every image, tables, worklists, candidates, selections, outputs
  uint8_t* b_zero = nullptr; size_t b_zero_bytes = 0;   // the part cleared before every batch (occupancy, counters)
  int b_last_n = 0;
  // last call (for ps_cvorb_debug_read)
  std::vector<std::vector<float>> last_cand;   // per level: rows of 4 floats
  bool last_had_mask = false;
  // r06: a masked single image goes through the batched, device-resident form (a batch of one): no host round trips around retainBest,
  // work proportional to the masked area.  Results land in one page-locked block (count, overflow, keypoints, descriptors: one wait).
  uint8_t* h_out = nullptr; size_t h_out_bytes = 0;
  uint8_t* h_in = nullptr; size_t h_in_bytes = 0;   // page-locked staging of image + mask (a 2-D copy from pageable memory goes row by row: 3 ms per image)
  bool fast = true;                            // PS_CVORB_FAST=0: always the per-level host-selected form below
  bool last_fast = false;                      // the staged inputs of the last call have not been through that form (ps_cvorb_debug_read runs it)
};

namespace {
inline int cv_round(double v) { return (int)nearbyint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

// interpolationLinear<ufixedpoint16>::getCoeffs (resize.cpp): source offset and 8.8 weights per destination index
void exact_table(int ssize, int dsize, std::vector<int4>& tab, double* scale_out = nullptr, int* dmin_out = nullptr, int* dmax_out = nullptr) {
  const double inv_scale = (double)dsize / ssize, scale = 1.0 / inv_scale;
  tab.assign(dsize, int4{0, 0, 0, 0});
  int dmin = 0, dmax = dsize;
  for (int val = 0; val < dsize; val++) {
    const double fval = scale * ((double)val + 0.5) - 0.5;
    const int ival = cv_floor(fval);
    if (ival >= 0 && ssize > 1) {
      if (ival < ssize - 1) {
        const int c1 = cv_round((fval - (double)ival) * 256.0);
        tab[val] = int4{ival, 256 - c1, c1, 0};
      } else { tab[val] = int4{ssize - 1, 0, 0, 0}; dmax = std::min(dmax, val); }
    } else dmin = std::max(dmin, val + 1);
  }
  for (int val = 0; val < dsize; val++) tab[val].w = val < dmin ? 1 : (val >= dmax ? 2 : 0);
  if (scale_out) *scale_out = scale;
  if (dmin_out) *dmin_out = dmin;
  if (dmax_out) *dmax_out = dmax;
}

void retain_best(std::vector<CvSel>& kp, int n_points) {   // KeyPointsFilter::retainBest (features2d/src/keypoint.cpp)
  if (n_points >= 0 && kp.size() > (size_t)n_points) {
    if (n_points == 0) { kp.clear(); return; }
    std::nth_element(kp.begin(), kp.begin() + n_points - 1, kp.end(), [](const CvSel& a, const CvSel& b) { return a.response > b.response; });
    const float ambiguous_response = kp[n_points - 1].response;
    auto new_end = std::partition(kp.begin() + n_points, kp.end(), [ambiguous_response](const CvSel& k) { return k.response >= ambiguous_response; });
    kp.resize(new_end - kp.begin());
  }
}

int build_plan(ps_cvorb* h, int w, int hgt) {
  if (h->d_buf) { hipFree(h->d_buf); h->d_buf = nullptr; }
  CvPlanDev& P = h->plan;
  memset(&P, 0, sizeof(P));
  P.nlevels = h->nlevels;
  {   // umax of the circular patch (orb.cpp computeKeyPoints)
    const int hp = 15;
    int v, v0, vmax = cv_floor(hp * sqrtf(2.f) / 2 + 1), vmin = cv_ceil(hp * sqrtf(2.f) / 2);
    for (v = 0; v <= vmax; ++v) P.umax[v] = cv_round(sqrt((double)hp * hp - v * v));
    for (v = hp, v0 = 0; v >= vmin; --v) { while (P.umax[v0] == P.umax[v0 + 1]) ++v0; P.umax[v] = v0; ++v0; }
  }
  h->quota.assign(h->nlevels, 0);
  {
    const float factor = (float)(1.0 / h->scale_factor);
    float ndesired = h->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)h->nlevels));
    int sum = 0;
    for (int l = 0; l < h->nlevels - 1; l++) { h->quota[l] = cv_round(ndesired); sum += h->quota[l]; ndesired *= factor; }
    h->quota[h->nlevels - 1] = std::max(h->nfeatures - sum, 0);
  }
  std::vector<std::vector<int4>> xt(h->nlevels), yt(h->nlevels);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off += al(bytes + 64); return r; };
  struct Ofs { size_t pad, blur, mask, score, rowcnt, rowoff, cand, xtab, ytab; };
  std::vector<Ofs> ofs(h->nlevels);
  h->cap.assign(h->nlevels, 0);
  for (int l = 0; l < h->nlevels; l++) {
    CvLevelDev& L = P.lv[l];
    L.scale = (float)pow(h->scale_factor, (double)l);
    const float inv_scale = 1.0f / L.scale;
    L.w = cv_round(w * inv_scale); L.h = cv_round(hgt * inv_scale);
    if (L.w <= 2 * h->edge || L.h <= 2 * h->edge || L.w <= 2 * 16 || L.h <= 2 * 16)
      return ps_set_error(PS_ERR_INVALID, "image %dx%d: level %d is %dx%d, too small for the detector", w, hgt, l, L.w, L.h);
    L.stride = (int)((L.w + 2 * CV_BORDER + 63) / 64 * 64);
    h->cap[l] = ((L.w + 1) / 2) * ((L.h + 1) / 2);            // strict 3 x 3 maxima: at most one per 2 x 2 block
    ofs[l].pad = take((size_t)L.stride * (L.h + 2 * CV_BORDER)); ofs[l].blur = take((size_t)L.stride * (L.h + 2 * CV_BORDER)); ofs[l].mask = take((size_t)L.w * L.h);
    ofs[l].score = take((size_t)L.w * L.h); ofs[l].rowcnt = take((size_t)L.h * 4); ofs[l].rowoff = take((size_t)L.h * 4);
    ofs[l].cand = take((size_t)h->cap[l] * 16);
    if (l > 0) {
      exact_table(P.lv[l - 1].w, L.w, xt[l]); exact_table(P.lv[l - 1].h, L.h, yt[l]);
      ofs[l].xtab = take(xt[l].size() * 16); ofs[l].ytab = take(yt[l].size() * 16);
    }
  }
  const size_t o_img = take((size_t)w * hgt), o_mask = take((size_t)w * hgt), o_total = take(h->nlevels * 4);
  h->sel_cap = 0;
  for (int l = 0; l < h->nlevels; l++) h->sel_cap += h->cap[l];   // ties may keep more than the quota; bounded by the candidates
  h->sel_cap = std::min(h->sel_cap, 16 * h->nfeatures + 4096);
  const size_t o_sel = take((size_t)h->sel_cap * sizeof(CvSel)), o_kps = take((size_t)h->sel_cap * sizeof(ps_keypoint)), o_desc = take((size_t)h->sel_cap * 32);
  PS_HIP(hipMalloc(&h->d_buf, off));
  PS_HIP(hipMemsetAsync(h->d_buf, 0, off, h->stream));
  uint8_t* D = h->d_buf;
  for (int l = 0; l < h->nlevels; l++) {
    CvLevelDev& L = P.lv[l];
    L.pad = D + ofs[l].pad; L.blur = D + ofs[l].blur; L.mask = D + ofs[l].mask; L.score = D + ofs[l].score;
    L.rowcnt = (int32_t*)(D + ofs[l].rowcnt); L.rowoff = (int32_t*)(D + ofs[l].rowoff); L.cand = (float4*)(D + ofs[l].cand);
    if (l > 0) {
      h->d_xtab[l] = (int4*)(D + ofs[l].xtab); h->d_ytab[l] = (int4*)(D + ofs[l].ytab);
      PS_HIP(hipMemcpyAsync(h->d_xtab[l], xt[l].data(), xt[l].size() * 16, hipMemcpyHostToDevice, h->stream));
      PS_HIP(hipMemcpyAsync(h->d_ytab[l], yt[l].data(), yt[l].size() * 16, hipMemcpyHostToDevice, h->stream));
    }
  }
  h->d_img = D + o_img; h->d_mask = D + o_mask; h->d_total = (int32_t*)(D + o_total);
  h->d_sel = (CvSel*)(D + o_sel); h->d_kps = (ps_keypoint*)(D + o_kps); h->d_desc = D + o_desc;
  PS_HIP(hipStreamSynchronize(h->stream));   // the tables above are read from vectors that go out of scope
  h->w = w; h->h = hgt; h->planned = true;
  return PS_OK;
}
}  // namespace

extern "C" {

int ps_cvorb_create(int nfeatures, float scale_factor, int nlevels, int edge_threshold, int fast_threshold, int device, ps_cvorb** out) {
  if (!out) return ps_set_error(PS_ERR_INVALID, "null argument");
  if (nfeatures < 1 || !(scale_factor > 1.f) || nlevels < 1 || nlevels > CV_MAX_LEVELS || edge_threshold < 3 || edge_threshold > CV_BORDER - 4 || fast_threshold < 1 || fast_threshold > 254)
    return ps_set_error(PS_ERR_INVALID, "ps_cvorb_create: unsupported configuration (1..8 levels, edge threshold 3..19)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ps_set_error(PS_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return ps_set_error(PS_ERR_INVALID, "bad device ordinal");
  PS_HIP(hipSetDevice(device));
  ps_cvorb* h = new ps_cvorb();
  h->nfeatures = nfeatures; h->nlevels = nlevels; h->edge = edge_threshold; h->fast_th = fast_threshold; h->device = device;
  h->scale_factor = (double)scale_factor;   // ORB::create takes a float, ORB_Impl keeps a double
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; return ps_set_error(PS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
  {   // getGaussianKernel(7, 2) in 8.8 fixed point
    double v[7], sum = 0;
    for (int i = 0; i < 7; i++) { const double x = i - 3; v[i] = exp(-0.5 * x * x / 4.0); sum += v[i]; }
    for (int i = 0; i < 7; i++) h->kq[i] = cv_round(v[i] / sum * 256.0);
  }
  if (const char* f = getenv("PS_CVORB_FAST")) h->fast = atoi(f) != 0;
  *out = h;
  return PS_OK;
}

void ps_cvorb_destroy(ps_cvorb* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) { hipStreamSynchronize(h->stream); hipStreamDestroy(h->stream); }
  if (h->d_buf) hipFree(h->d_buf);
  if (h->b_buf) hipFree(h->b_buf);
  if (h->h_out) hipHostFree(h->h_out);
  if (h->h_in) hipHostFree(h->h_in);
  delete h;
}

}  // extern "C"

extern "C" int psi_cvorb_batch_begin(ps_cvorb* h, int nimg, int w, int hgt, hipStream_t st, uint8_t** occ, int* ocw, int* och);
extern "C" int psi_cvorb_batch_run(ps_cvorb* h, const uint8_t* d_imgs, const uint8_t* d_masks, int nimg, int stride, size_t image_pitch, int mask_stride,
                                   size_t mask_pitch, int occupancy_given, hipStream_t st);

namespace {
// The per-level form on the inputs staged in h->d_img / h->d_mask: full pyramid, FAST + Harris per level, the two retainBest steps on the
// host (std::nth_element / std::partition), descriptors.  kps == nullptr: only the intermediates (ps_cvorb_debug_read).
int single_host_selected(ps_cvorb* h, int w, bool has_mask, ps_keypoint* kps, uint8_t* desc, int cap, int* n) {
  hipStream_t st = h->stream;
  CvPlanDev P = h->plan;
  if (!has_mask) for (int l = 0; l < P.nlevels; l++) P.lv[l].mask = nullptr;
  // pyramid, detection
  psk_cv_level0(&P.lv[0], h->d_img, w, has_mask ? h->d_mask : nullptr, w, st);
  for (int l = 1; l < P.nlevels; l++) psk_cv_resize(&P.lv[l], &P.lv[l - 1], h->d_xtab[l], h->d_ytab[l], st);
  for (int l = 0; l < P.nlevels; l++) psk_cv_detect(&P.lv[l], h->fast_th, h->edge, h->cap[l], h->d_total + l, st);
  for (int l = 0; l < P.nlevels; l++) psk_cv_blur(&P.lv[l], h->kq, st);   // (independent of the selection: queued before the host waits)
  PS_HIP(hipGetLastError());
  std::vector<int32_t> total(P.nlevels, 0);
  PS_HIP(hipMemcpyAsync(total.data(), h->d_total, P.nlevels * 4, hipMemcpyDeviceToHost, st));
  PS_HIP(hipStreamSynchronize(st));
  // KeyPointsFilter::retainBest twice per level, in the order computeKeyPoints applies them
  h->last_cand.assign(P.nlevels, {});
  std::vector<CvSel> selected;
  for (int l = 0; l < P.nlevels; l++) {
    const int cnt = std::min(total[l], h->cap[l]);
    std::vector<float>& c = h->last_cand[l];
    c.resize((size_t)cnt * 4);
    if (cnt > 0) PS_HIP(hipMemcpy(c.data(), P.lv[l].cand, (size_t)cnt * 16, hipMemcpyDeviceToHost));
    // (the level field is constant inside this loop: it carries the candidate's index through the first selection, so that the
    // Harris response computed with the candidate can be attached afterwards, as HarrisResponses does)
    std::vector<CvSel> kp(cnt);
    for (int i = 0; i < cnt; i++) kp[i] = CvSel{(int)c[4 * i], (int)c[4 * i + 1], i, c[4 * i + 2]};
    retain_best(kp, 2 * h->quota[l]);                       // by FAST score: twice the quota (HARRIS_SCORE)
    for (CvSel& k : kp) k.response = c[4 * (size_t)k.level + 3];   // HarrisResponses
    retain_best(kp, h->quota[l]);                           // cull to the quota by the Harris score
    for (CvSel& k : kp) { k.level = l; selected.push_back(k); }
  }
  h->last_fast = false;
  if (!kps && !desc && !n) return PS_OK;
  const int nsel = (int)selected.size();
  if (nsel > h->sel_cap) return ps_set_error(PS_ERR_CAPACITY, "%d keypoints selected, internal capacity %d", nsel, h->sel_cap);
  *n = nsel;
  if (nsel == 0) return PS_OK;
  if (nsel > cap) return ps_set_error(PS_ERR_CAPACITY, "%d keypoints, caller capacity %d", nsel, cap);
  if (!kps || !desc) return ps_set_error(PS_ERR_INVALID, "null output buffer");
  PS_HIP(hipMemcpyAsync(h->d_sel, selected.data(), (size_t)nsel * sizeof(CvSel), hipMemcpyHostToDevice, st));
  psk_cv_describe(&P, h->d_sel, nsel, h->d_kps, h->d_desc, st);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemcpyAsync(kps, h->d_kps, (size_t)nsel * sizeof(ps_keypoint), hipMemcpyDeviceToHost, st));
  PS_HIP(hipMemcpyAsync(desc, h->d_desc, (size_t)nsel * 32, hipMemcpyDeviceToHost, st));
  PS_HIP(hipStreamSynchronize(st));
  return PS_OK;
}

// The batched form on the same staged inputs as a batch of one.  *served = false: the image does not fit that form (its plan's limits, or
// more FAST keypoints under the mask than its per-level / per-image stores hold) - the caller runs the per-level form instead.
int single_batched(ps_cvorb* h, int w, int hgt, ps_keypoint* kps, uint8_t* desc, int cap, int* n, bool* served) {
  *served = false;
  hipStream_t st = h->stream;
  if (psi_cvorb_batch_begin(h, 1, w, hgt, st, nullptr, nullptr, nullptr) != PS_OK) return PS_OK;
  int rc = psi_cvorb_batch_run(h, h->d_img, h->d_mask, 1, w, (size_t)w * hgt, w, (size_t)w * hgt, 0, st);
  if (rc != PS_OK) return rc;
  const size_t ocap = (size_t)h->bplan.ocap, need = 64 + ocap * (sizeof(ps_keypoint) + 32);
  if (need > h->h_out_bytes) {
    if (h->h_out) hipHostFree(h->h_out);
    h->h_out = nullptr; h->h_out_bytes = 0;
    PS_HIP(hipHostMalloc(&h->h_out, need, hipHostMallocDefault));
    h->h_out_bytes = need;
  }
  int32_t* hc = (int32_t*)h->h_out;
  uint8_t* hk = h->h_out + 64;
  uint8_t* hd = hk + ocap * sizeof(ps_keypoint);
  PS_HIP(hipMemcpyAsync(hc, h->bplan.count, 4, hipMemcpyDeviceToHost, st));
  PS_HIP(hipMemcpyAsync(hc + 1, h->bplan.overflow, 4, hipMemcpyDeviceToHost, st));
  PS_HIP(hipMemcpyAsync(hk, h->bplan.kps, ocap * sizeof(ps_keypoint), hipMemcpyDeviceToHost, st));
  PS_HIP(hipMemcpyAsync(hd, h->bplan.desc, ocap * 32, hipMemcpyDeviceToHost, st));
  PS_HIP(hipStreamSynchronize(st));
  if (hc[1]) return PS_OK;                                  // a store overflowed: not served
  *served = true;
  h->last_fast = true;
  const int cnt = hc[0];
  *n = cnt;
  if (cnt == 0) return PS_OK;
  if (cnt > cap) return ps_set_error(PS_ERR_CAPACITY, "%d keypoints, caller capacity %d", cnt, cap);
  if (!kps || !desc) return ps_set_error(PS_ERR_INVALID, "null output buffer");
  memcpy(kps, hk, (size_t)cnt * sizeof(ps_keypoint));
  memcpy(desc, hd, (size_t)cnt * 32);
  return PS_OK;
}
}  // namespace

extern "C" {
int ps_cvorb_detect_and_compute(ps_cvorb* h, const uint8_t* img, const uint8_t* mask, int w, int hgt, int stride, int mask_stride, ps_keypoint* kps,
                                uint8_t* desc, int cap, int* n) {
  if (!h || !n) return ps_set_error(PS_ERR_INVALID, "ps_cvorb_detect_and_compute: null argument");
  *n = 0;
  if (!img || w <= 0 || hgt <= 0) return PS_OK;
  if (stride < w || (mask && mask_stride < w)) return ps_set_error(PS_ERR_INVALID, "stride < width");
  PS_HIP(hipSetDevice(h->device));
  if (!h->planned || h->w != w || h->h != hgt) {
    int rc = build_plan(h, w, hgt);
    if (rc != PS_OK) return rc;
  }
  hipStream_t st = h->stream;
  {
    // image and mask, rows packed, through page-locked staging: two host memcpys and two DMA transfers
    const size_t px = (size_t)w * hgt, need = 2 * px;
    if (need > h->h_in_bytes) {
      if (h->h_in) hipHostFree(h->h_in);
      h->h_in = nullptr; h->h_in_bytes = 0;
      PS_HIP(hipHostMalloc(&h->h_in, need, hipHostMallocDefault));
      h->h_in_bytes = need;
    }
    auto pack = [&](uint8_t* dst, const uint8_t* src, int sstride) {
      if (sstride == w) memcpy(dst, src, px);
      else for (int y = 0; y < hgt; y++) memcpy(dst + (size_t)y * w, src + (size_t)y * sstride, w);
    };
    pack(h->h_in, img, stride);
    PS_HIP(hipMemcpyAsync(h->d_img, h->h_in, px, hipMemcpyHostToDevice, st));
    if (mask) {
      pack(h->h_in + px, mask, mask_stride);
      PS_HIP(hipMemcpyAsync(h->d_mask, h->h_in + px, px, hipMemcpyHostToDevice, st));
    }
  }
  h->last_had_mask = mask != nullptr;
  if (mask && h->fast) {
    bool served = false;
    const int rc = single_batched(h, w, hgt, kps, desc, cap, n, &served);
    if (rc != PS_OK || served) return rc;
    *n = 0;
  }
  return single_host_selected(h, w, mask != nullptr, kps, desc, cap, n);
}

// Test access to intermediates of the last call.  what: 0 level image (tight w x h), 1 blurred level, 2 level mask,
// 3 the FAST keypoints after the mask / border filters as float rows (x, y, score, Harris), count in *n; 4: level size as int32[2]
int ps_cvorb_debug_read(ps_cvorb* h, int level, int what, void* out, size_t out_bytes, int* n) {
  if (!h || !h->planned || level < 0 || level >= h->nlevels || !out) return ps_set_error(PS_ERR_INVALID, "ps_cvorb_debug_read: bad argument");
  PS_HIP(hipSetDevice(h->device));
  if (h->last_fast) {   // the last call was served by the batched form: the per-level intermediates are made now, from the inputs still staged
    const int rc = single_host_selected(h, h->w, h->last_had_mask, nullptr, nullptr, 0, nullptr);
    if (rc != PS_OK) return rc;
  }
  PS_HIP(hipDeviceSynchronize());
  const CvLevelDev& L = h->plan.lv[level];
  if (what == 4) { if (out_bytes < 8) return ps_set_error(PS_ERR_CAPACITY, "buffer too small"); ((int32_t*)out)[0] = L.w; ((int32_t*)out)[1] = L.h; return PS_OK; }
  if (what == 3) {
    const std::vector<float>& c = h->last_cand[level];
    if (c.size() * 4 > out_bytes) return ps_set_error(PS_ERR_CAPACITY, "buffer too small");
    if (!c.empty()) memcpy(out, c.data(), c.size() * 4);
    if (n) *n = (int)(c.size() / 4);
    return PS_OK;
  }
  if ((size_t)L.w * L.h > out_bytes) return ps_set_error(PS_ERR_CAPACITY, "buffer too small");
  if (what == 0) PS_HIP(hipMemcpy2D(out, L.w, L.pad + (size_t)CV_BORDER * L.stride + CV_BORDER, L.stride, L.w, L.h, hipMemcpyDeviceToHost));
  else if (what == 1) PS_HIP(hipMemcpy2D(out, L.w, L.blur + (size_t)CV_BORDER * L.stride + CV_BORDER, L.stride, L.w, L.h, hipMemcpyDeviceToHost));
  else if (what == 2) { if (h->last_had_mask) PS_HIP(hipMemcpy(out, L.mask, (size_t)L.w * L.h, hipMemcpyDeviceToHost)); else memset(out, 0, (size_t)L.w * L.h); }
  else return ps_set_error(PS_ERR_INVALID, "unknown `what` %d", what);
  return PS_OK;
}

}  // extern "C"

namespace {
// plan of the batched form for `cap` images of w x hgt
int build_batch_plan(ps_cvorb* h, int w, int hgt, int cap) {
  if (h->b_buf) { hipFree(h->b_buf); h->b_buf = nullptr; }
  h->bplanned = false;
  CvbPlan& P = h->bplan;
  memset(&P, 0, sizeof(P));
  P.nlevels = h->nlevels; P.edge = h->edge; P.fast_th = h->fast_th; P.w0 = w; P.h0 = hgt;
  P.ocw = (w + 7) / 8; P.och = (hgt + 7) / 8;
  if (P.ocw > 512) return ps_set_error(PS_ERR_INVALID, "the batched object detector supports images up to 4096 pixels wide");
  {
    const int hp = 15;
    int v, v0, vmax = cv_floor(hp * sqrtf(2.f) / 2 + 1), vmin = cv_ceil(hp * sqrtf(2.f) / 2);
    for (v = 0; v <= vmax; ++v) P.umax[v] = cv_round(sqrt((double)hp * hp - v * v));
    for (v = hp, v0 = 0; v >= vmin; --v) { while (P.umax[v0] == P.umax[v0 + 1]) ++v0; P.umax[v] = v0; ++v0; }
    // cvb_describe reads the disc through the byte masks of describe_common.h (make_ictab), which are built from these sixteen numbers
    static const int disc[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (v = 0; v <= hp; ++v)
      if (P.umax[v] != disc[v]) return ps_set_error(PS_ERR_INVALID, "the intensity-centroid disc differs from the kernels' mask table");
  }
  for (int i = 0; i < 4; i++) P.kq[i] = h->kq[i];
  std::vector<int> quota(h->nlevels, 0);
  {
    const float factor = (float)(1.0 / h->scale_factor);
    float ndesired = h->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)h->nlevels));
    int sum = 0;
    for (int l = 0; l < h->nlevels - 1; l++) { quota[l] = cv_round(ndesired); sum += quota[l]; ndesired *= factor; }
    quota[h->nlevels - 1] = std::max(h->nfeatures - sum, 0);
  }
  std::vector<std::vector<int4>> xt(h->nlevels), yt(h->nlevels);
  size_t img_off = 0;
  auto take_img = [&](size_t bytes) { size_t r = img_off; img_off += al(bytes + 64); return r; };
  for (int l = 0; l < h->nlevels; l++) {
    CvbLevel& L = P.lv[l];
    L.scale = (float)pow(h->scale_factor, (double)l);
    const float inv_scale = 1.0f / L.scale;
    L.w = cv_round(w * inv_scale); L.h = cv_round(hgt * inv_scale);
    if (L.w <= 2 * h->edge || L.h <= 2 * h->edge || L.w <= 2 * 24 || L.h <= 2 * 24)
      return ps_set_error(PS_ERR_INVALID, "image %dx%d: level %d is %dx%d, too small for the detector", w, hgt, l, L.w, L.h);
    L.stride = (int)((L.w + 2 * CV_BORDER + 63) / 64 * 64);
    L.tw = (L.w + 2 * CV_BORDER + CVB_TILE - 1) / CVB_TILE; L.th = (L.h + 2 * CV_BORDER + CVB_TILE - 1) / CVB_TILE;
    if (L.tw * L.th > CVB_MAX_TILES) return ps_set_error(PS_ERR_INVALID, "the batched object detector supports levels of up to %d tiles", CVB_MAX_TILES);
    L.cw = (L.w + 2 * CV_BORDER + 7) / 8; L.ch = (L.h + 2 * CV_BORDER + 7) / 8;
    L.cw = std::max(L.cw, 4 * L.tw); L.ch = std::max(L.ch, 4 * L.th);   // every tile owns 4 x 4 cells (cw is a multiple of 4: a tile's cell row is one aligned dword)
    L.cw = (L.cw + 3) & ~3;
    if (L.cw > 1024 || L.ch > 512) return ps_set_error(PS_ERR_INVALID, "image too large for the batched object detector's planning kernel");   // CVB_PLAN_MAXCW / MAXCH
    L.cell_off = P.cell_total; P.cell_total += L.cw * L.ch; P.cell_max = std::max(P.cell_max, L.cw * L.ch);
    L.tile_off = P.tile_total; P.tile_total += L.tw * L.th;
    L.quota = quota[l];
    L.o_pad = take_img((size_t)L.stride * (L.h + 2 * CV_BORDER)); L.o_blur = take_img((size_t)L.stride * (L.h + 2 * CV_BORDER));
    L.o_mask = take_img((size_t)L.w * L.h); L.o_score = 0;
    if (l > 0) {
      exact_table(P.lv[l - 1].w, L.w, xt[l], &L.sx, &L.dminx, &L.dmaxx); exact_table(P.lv[l - 1].h, L.h, yt[l], &L.sy, &L.dminy, &L.dmaxy);
      // cvb_resize stages the source patch of a 32 x 32 tile in a 48 x 48 LDS array
      if ((double)P.lv[l - 1].w / L.w * 32 + 6 > 48 || (double)P.lv[l - 1].h / L.h * 32 + 6 > 48)
        return ps_set_error(PS_ERR_INVALID, "the batched object detector supports scale factors up to 1.3");
    }
  }
  P.arena_pitch = img_off;
  P.ocap = 2048;
  int max_tiles = 0;
  for (int l = 0; l < h->nlevels; l++) max_tiles = std::max(max_tiles, P.lv[l].tw * P.lv[l].th);
  P.wl_cap = cap * max_tiles;
  if (2 * (size_t)P.cell_total + 2 * (size_t)P.cell_max + 2 * (size_t)(P.ocw + 1) * (P.och + 1) + P.tile_total + 64 > 150 * 1024 || P.tile_total > 8 * 512) return ps_set_error(PS_ERR_INVALID, "image too large for the batched object detector's planning kernel");
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off += al(bytes + 64); return r; };
  const size_t o_arena = take(P.arena_pitch * (size_t)cap);
  std::vector<size_t> o_xt(h->nlevels, 0), o_yt(h->nlevels, 0);
  for (int l = 1; l < h->nlevels; l++) { o_xt[l] = take(xt[l].size() * 16); o_yt[l] = take(yt[l].size() * 16); }
  const size_t o_wl = take((size_t)3 * CV_MAX_LEVELS * P.wl_cap * 4);
  const size_t o_kpmap = take((size_t)cap * P.cell_total);
  const size_t o_cand = take((size_t)cap * h->nlevels * CVB_CAND_CAP * sizeof(float4));
  const size_t o_sel = take((size_t)cap * h->nlevels * CVB_CAND_CAP * sizeof(CvSel));
  const size_t o_kps = take((size_t)cap * P.ocap * sizeof(ps_keypoint)), o_desc = take((size_t)cap * P.ocap * 32);
  const size_t o_dump = take(1024);
  // cleared before every batch: occupancy, worklist counters, candidate counters, selection counts, counts, overflow
  const size_t z0 = off;
  const size_t o_occ = take((size_t)cap * P.ocw * P.och), o_wlc = take(3 * CV_MAX_LEVELS * 4), o_ncand = take((size_t)cap * h->nlevels * 4);
  const size_t o_nsel = take((size_t)cap * h->nlevels * 4), o_count = take((size_t)cap * 4), o_ovf = take((size_t)cap * 4);
  const size_t z1 = off;
  hipError_t e = hipMalloc(&h->b_buf, off);
  if (e != hipSuccess) return ps_set_error(PS_ERR_HIP, "hipMalloc(%zu): %s", off, hipGetErrorString(e));
  uint8_t* D = h->b_buf;
  if (const char* fill = getenv("PS_DEBUG_FILL")) PS_HIP(hipMemsetAsync(D, atoi(fill), off, h->stream));   // diagnostic: the inactive tiles stay poisoned
  P.arena = D + o_arena;
  for (int l = 1; l < h->nlevels; l++) {
    P.lv[l].xtab = (const int4*)(D + o_xt[l]); P.lv[l].ytab = (const int4*)(D + o_yt[l]);
    PS_HIP(hipMemcpyAsync(D + o_xt[l], xt[l].data(), xt[l].size() * 16, hipMemcpyHostToDevice, h->stream));
    PS_HIP(hipMemcpyAsync(D + o_yt[l], yt[l].data(), yt[l].size() * 16, hipMemcpyHostToDevice, h->stream));
  }
  P.wl = (uint32_t*)(D + o_wl); P.cand = (float4*)(D + o_cand); P.sel = (CvSel*)(D + o_sel);
  P.kps = (ps_keypoint_pod*)(D + o_kps); P.desc = D + o_desc; P.dump = D + o_dump;
  P.occ = D + o_occ; P.kpmap = D + o_kpmap; P.wl_count = (int32_t*)(D + o_wlc); P.ncand = (int32_t*)(D + o_ncand); P.nsel = (int32_t*)(D + o_nsel);
  P.count = (int32_t*)(D + o_count); P.overflow = (int32_t*)(D + o_ovf);
  h->b_zero = D + z0; h->b_zero_bytes = z1 - z0;
  PS_HIP(hipStreamSynchronize(h->stream));
  h->bw = w; h->bh = hgt; h->bcap = cap; h->bplanned = true;
  return PS_OK;
}
}  // namespace

extern "C" {
// internal (track_host.hip): the handle's stream
hipStream_t psi_cvorb_stream(ps_cvorb* h) { return h->stream; }

// internal (track_host.hip): the batch in two halves, so that a caller that already walks the masks (the tracker's ob_masks kernel)
// can fill the 8 x 8 cell occupancy itself between them: begin = plan + clearing the per-batch counters, run = the kernels
int psi_cvorb_batch_begin(ps_cvorb* h, int nimg, int w, int hgt, hipStream_t st, uint8_t** occ, int* ocw, int* och) {
  if (!h->bplanned || h->bw != w || h->bh != hgt || h->bcap < nimg) {
    int rc = build_batch_plan(h, w, hgt, nimg);
    if (rc != PS_OK) return rc;
  }
  PS_HIP(hipMemsetAsync(h->b_zero, 0, h->b_zero_bytes, st));
  if (occ) *occ = h->bplan.occ;
  if (ocw) *ocw = h->bplan.ocw;
  if (och) *och = h->bplan.och;
  return PS_OK;
}
int psi_cvorb_batch_run(ps_cvorb* h, const uint8_t* d_imgs, const uint8_t* d_masks, int nimg, int stride, size_t image_pitch, int mask_stride,
                        size_t mask_pitch, int occupancy_given, hipStream_t st) {
  psk_cvb_run(&h->bplan, nimg, d_imgs, stride, image_pitch, d_masks, mask_stride, mask_pitch, occupancy_given, st);
  PS_HIP(hipGetLastError());
  h->b_last_n = nimg;
  return PS_OK;
}

// developer access (tools/cvorb_batch_bench.py): entries of the three tile worklists per level after the last batch and the tiles of
// one image's level - how much of the pyramid the batch worked on
int psi_cvorb_batch_worklist_counts(ps_cvorb* h, int32_t* counts /* [3][CV_MAX_LEVELS] */, int32_t* tiles /* [CV_MAX_LEVELS] */) {
  if (!h || !h->bplanned || !counts || !tiles) return ps_set_error(PS_ERR_INVALID, "no batch plan yet");
  PS_HIP(hipSetDevice(h->device));
  PS_HIP(hipDeviceSynchronize());
  PS_HIP(hipMemcpy(counts, h->bplan.wl_count, 3 * CV_MAX_LEVELS * 4, hipMemcpyDeviceToHost));
  for (int l = 0; l < CV_MAX_LEVELS; l++) tiles[l] = l < h->nlevels ? h->bplan.lv[l].tw * h->bplan.lv[l].th : 0;
  return PS_OK;
}

int ps_cvorb_detect_batch_device(ps_cvorb* h, const uint8_t* d_imgs, const uint8_t* d_masks, int nimg, int w, int hgt, int stride, size_t image_pitch,
                                 int mask_stride, size_t mask_pitch, void* stream) {
  if (!h || !d_imgs || !d_masks || nimg < 1 || w < 1 || hgt < 1 || stride < w || mask_stride < w)
    return ps_set_error(PS_ERR_INVALID, "ps_cvorb_detect_batch_device: bad argument");
  PS_HIP(hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  int rc = psi_cvorb_batch_begin(h, nimg, w, hgt, st, nullptr, nullptr, nullptr);
  if (rc != PS_OK) return rc;
  return psi_cvorb_batch_run(h, d_imgs, d_masks, nimg, stride, image_pitch, mask_stride, mask_pitch, 0, st);
}

int ps_cvorb_batch_device_outputs(const ps_cvorb* h, const ps_keypoint** d_kps, const uint8_t** d_desc, const int32_t** d_counts,
                                  const int32_t** d_overflow, int32_t* capacity) {
  if (!h || !h->bplanned) return ps_set_error(PS_ERR_INVALID, "no batch plan yet");
  if (d_kps) *d_kps = (const ps_keypoint*)h->bplan.kps;
  if (d_desc) *d_desc = h->bplan.desc;
  if (d_counts) *d_counts = h->bplan.count;
  if (d_overflow) *d_overflow = h->bplan.overflow;
  if (capacity) *capacity = h->bplan.ocap;
  return PS_OK;
}

int ps_cvorb_batch_fetch(ps_cvorb* h, int image, ps_keypoint* kps, uint8_t* desc, int cap, int* n) {
  if (!h || !h->bplanned || image < 0 || image >= h->b_last_n || !n) return ps_set_error(PS_ERR_INVALID, "ps_cvorb_batch_fetch: bad argument");
  PS_HIP(hipSetDevice(h->device));
  PS_HIP(hipDeviceSynchronize());
  int32_t cnt = 0, ovf = 0;
  PS_HIP(hipMemcpy(&cnt, h->bplan.count + image, 4, hipMemcpyDeviceToHost));
  PS_HIP(hipMemcpy(&ovf, h->bplan.overflow + image, 4, hipMemcpyDeviceToHost));
  if (ovf) return ps_set_error(PS_ERR_CAPACITY, "image %d: more than %d FAST keypoints under the mask on a level, or more than %d keypoints in all", image, CVB_CAND_CAP, h->bplan.ocap);
  *n = cnt;
  if (cnt > cap) return ps_set_error(PS_ERR_CAPACITY, "%d keypoints, caller capacity %d", cnt, cap);
  if (cnt > 0) {
    if (!kps || !desc) return ps_set_error(PS_ERR_INVALID, "null output buffer");
    PS_HIP(hipMemcpy(kps, (const ps_keypoint*)h->bplan.kps + (size_t)image * h->bplan.ocap, (size_t)cnt * sizeof(ps_keypoint), hipMemcpyDeviceToHost));
    PS_HIP(hipMemcpy(desc, h->bplan.desc + (size_t)image * h->bplan.ocap * 32, (size_t)cnt * 32, hipMemcpyDeviceToHost));
  }
  return PS_OK;
}
}  // extern "C"
