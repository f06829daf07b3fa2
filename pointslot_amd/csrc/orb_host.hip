// Host side of the ORB extractor C-ABI (include/pointslot_hip.h): plan construction (level
// geometry, FAST cell grid, resize coefficient tables, arena layout), stream orchestration and the
// host<->device copies.  Replaces ORB_SLAM2::ORBextractor — /root/reference/src/ORBextractor.cc.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/pointslot_hip.h"
#include "orb_plan.h"
#include "ps_common.h"

extern "C" {
void psk_orb_launch_pyramid(const OrbPlan*, int, uint8_t*, const uint8_t*, int, size_t, const int4*, int, hipStream_t);
void psk_orb_launch_fast(const OrbPlan*, uint8_t*, int, const uint8_t*, int, size_t, const int4*, hipStream_t);
void psk_orb_launch_quadtree(const OrbPlan*, uint8_t*, int, hipStream_t);
void psk_orb_launch_blur(const OrbPlan*, uint8_t*, int, hipStream_t);
void psk_orb_launch_border(const OrbPlan*, uint8_t*, int, hipStream_t);
int psk_orb_blur_rows();
void psk_orb_launch_describe(const OrbPlan*, uint8_t*, void*, uint8_t*, int32_t*, int, hipStream_t);
void psk_orb_launch_level_fused(const OrbPlan*, int, uint8_t*, const uint8_t*, int, size_t, const int4*, int, hipStream_t);
void psk_stereo_launch(const OrbPlan*, const StPair*, int, int, float, float, hipStream_t);
}

namespace {

inline int cv_round(double v) { return (int)nearbyint(v); }   // cvRound: round half to even
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

enum { ST_PYRAMID = 0, ST_FAST, ST_QUADTREE, ST_BLUR, ST_DESCRIBE, ST_COUNT };
const char* kStageNames[ST_COUNT] = {"orb_pyramid_level", "orb_fast_cells", "orb_quadtree", "orb_blur",
                                     "orb_describe"};
// with the fused level kernel the first stage is pyramid + border + blur (8 launches of orb_level_fused) and the fourth is empty
const char* kStageNamesFused[ST_COUNT] = {"orb_level_fused", "orb_fast_cells", "orb_quadtree", "orb_blur",
                                          "orb_describe"};

}  // namespace

struct ps_orb {
  ps_orb_config cfg;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> quota;
  OrbPlan plan;
  bool planned = false;
  std::vector<int4> tabs_host;
  // device
  uint8_t* d_arena = nullptr;
  int4* d_tabs = nullptr;
  ps_keypoint* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  int32_t* d_counts = nullptr;
  // stereo matcher outputs, indexed like the keypoints of the LEFT image: [max_batch][kp_cap]
  float* d_uright = nullptr; float* d_depth = nullptr; int32_t* d_sad = nullptr; int32_t* d_kept = nullptr;
  StPair* d_pairs = nullptr;
  std::vector<StPair> pairs_host;   // what d_pairs holds
  uint8_t* d_stscratch = nullptr;   // [max_batch][PS_ST_SCRATCH]
  int last_npairs = 0;
  uint8_t* d_objkeys = nullptr;   // scratch of ps_orb_stereo_match_keys (caller-provided key sets + their outputs)
  uint8_t* h_objkeys = nullptr;
  uint8_t* d_img = nullptr;       // staging for ps_orb_extract / ps_orb_extract_batch (host images)
  size_t d_img_bytes = 0;
  // object-feature variant: masks of the images of the NEXT batch (ps_orb_extract_masked sets and clears it around its batch)
  const uint8_t* d_mask = nullptr; int mask_stride = 0; size_t mask_pitch = 0;
  uint8_t* d_mask_buf = nullptr; size_t d_mask_bytes = 0;
  uint8_t* h_frames = nullptr;    // pinned staging of ps_orb_stereo_fetch_frames
  size_t h_frames_bytes = 0;
  uint8_t* h_in = nullptr; size_t h_in_bytes = 0;   // pinned staging of ps_orb_extract's image when the caller's buffer is pageable
  uint8_t* h_one = nullptr;       // pinned staging of the single-image calls (ps_orb_extract, ps_orb_stereo_match_pair): count + the whole
  size_t h_one_bytes = 0;         // output capacity come back in one transfer group and one wait, not one blocking copy per array
  hipStream_t stream = nullptr;
  // host-image batches: the upload of a batch runs on its own stream, behind the level-0 kernel of the batch before it (the only
  // reader of the staging buffer) and ahead of its own kernels - it overlaps the rest of the previous batch's work
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_input_free = nullptr, ev_uploaded = nullptr;
  bool input_read_pending = false;
  // stage timing: a ring of event sets so that consecutive batches can be timed without a host
  // synchronisation in between; ps_orb_stage_times() averages over the recorded batches.
  static const int RING = 32;
  static const int MAXCHUNK = 32;
  hipEvent_t ev[RING][MAXCHUNK][ST_COUNT + 1] = {};
  int timed_chunks[RING] = {};
  bool fused = true;              // one launch per level writes plane + border + blur (PS_ORB_FUSED=0: separate kernels)
  int chunk = 0;                  // 0 = whole batch per launch (kernels are latency-bound: fewer, larger launches win; PS_ORB_CHUNK overrides)
  bool timing = false;
  int timed_batches = 0;
  int last_nimg = 0;
  hipStream_t last_stream = nullptr;   // stream of the last batch (the handle's own unless the caller passed one)
};

namespace {

// ORBextractor ctor tables, /root/reference/src/ORBextractor.cc:415-446.  scaleFactor is stored in a
// double member (include/ORBextractor.h:98), so the products below are double then narrowed.
void build_tables(ps_orb* h) {
  const int nl = h->cfg.nlevels;
  const double sf = (double)h->cfg.scale_factor;
  h->scale.assign(nl, 1.f);
  h->sigma2.assign(nl, 1.f);
  for (int i = 1; i < nl; i++) {
    h->scale[i] = (float)(h->scale[i - 1] * sf);
    h->sigma2[i] = h->scale[i] * h->scale[i];
  }
  h->inv_scale.resize(nl);
  h->inv_sigma2.resize(nl);
  for (int i = 0; i < nl; i++) {
    h->inv_scale[i] = 1.0f / h->scale[i];
    h->inv_sigma2[i] = 1.0f / h->sigma2[i];
  }
  h->quota.resize(nl);
  const float factor = (float)(1.0f / sf);
  float want = h->cfg.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    h->quota[l] = cv_round(want);
    sum += h->quota[l];
    want *= factor;
  }
  h->quota[nl - 1] = h->cfg.nfeatures - sum > 0 ? h->cfg.nfeatures - sum : 0;
}

void level_size(const ps_orb* h, int w, int hgt, int l, int* wl, int* hl) {
  const float s = h->inv_scale[l];   // ORBextractor.cc:1111-1112
  *wl = cv_round((float)w * s);
  *hl = cv_round((float)hgt * s);
}

// OpenCV 3.4 resize(INTER_LINEAR) coefficient tables for src (sw x sh) -> dst (dw x dh), 8-bit:
// fx = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx; clamp; alpha = sat<short>(c*2048)
inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
void resize_tables(int sw, int sh, int dw, int dh, std::vector<int4>& out, uint32_t* xoff, uint32_t* yoff) {
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  *xoff = (uint32_t)out.size();
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const float c0 = 1.f - fx, c1 = fx;
    int4 e;
    e.x = sx;
    e.y = sx + 1 < sw ? sx + 1 : sx;
    e.z = sat_short(cv_round(c0 * 2048.f));
    e.w = sat_short(cv_round(c1 * 2048.f));
    out.push_back(e);
  }
  *yoff = (uint32_t)out.size();
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    const float c0 = 1.f - fy, c1 = fy;
    auto clip = [&](int v) { return v >= 0 ? (v < sh ? v : sh - 1) : 0; };
    int4 e;
    e.x = clip(sy);
    e.y = clip(sy + 1);
    e.z = sat_short(cv_round(c0 * 2048.f));
    e.w = sat_short(cv_round(c1 * 2048.f));
    out.push_back(e);
  }
}

int free_device(ps_orb* h) {
  if (h->d_arena) hipFree(h->d_arena);
  if (h->d_tabs) hipFree(h->d_tabs);
  if (h->d_kps) hipFree(h->d_kps);
  if (h->d_desc) hipFree(h->d_desc);
  if (h->d_counts) hipFree(h->d_counts);
  if (h->d_uright) hipFree(h->d_uright);
  if (h->d_depth) hipFree(h->d_depth);
  if (h->d_sad) hipFree(h->d_sad);
  if (h->d_kept) hipFree(h->d_kept);
  if (h->d_pairs) hipFree(h->d_pairs);
  h->pairs_host.clear();
  if (h->d_stscratch) hipFree(h->d_stscratch);
  h->d_stscratch = nullptr;
  if (h->d_objkeys) hipFree(h->d_objkeys);
  if (h->h_objkeys) hipHostFree(h->h_objkeys);
  if (h->h_frames) hipHostFree(h->h_frames);
  if (h->h_one) hipHostFree(h->h_one);
  if (h->h_in) hipHostFree(h->h_in);
  h->h_in = nullptr; h->h_in_bytes = 0;
  h->d_objkeys = nullptr; h->h_objkeys = nullptr; h->h_frames = nullptr; h->h_frames_bytes = 0; h->h_one = nullptr; h->h_one_bytes = 0;
  h->d_uright = nullptr; h->d_depth = nullptr; h->d_sad = nullptr; h->d_kept = nullptr; h->d_pairs = nullptr;
  h->d_arena = nullptr; h->d_tabs = nullptr; h->d_kps = nullptr; h->d_desc = nullptr; h->d_counts = nullptr;
  return 0;
}

int build_plan(ps_orb* h, int w, int hgt) {
  OrbPlan& P = h->plan;
  memset(&P, 0, sizeof(P));
  const int nl = h->cfg.nlevels;
  P.nlevels = nl;
  P.img_w = w;
  P.img_h = hgt;
  P.ini_th = h->cfg.ini_th_fast;
  P.min_th = h->cfg.min_th_fast;
  h->tabs_host.clear();
  size_t off = 0;
  uint32_t cand_elems = 0, key_elems = 0;
  int cells = 0, sel = 0;
  for (int l = 0; l < nl; l++) {
    OrbLevel& L = P.lv[l];
    level_size(h, w, hgt, l, &L.w, &L.h);
    if (L.w - 2 * PS_MINB < 30 || L.h - 2 * PS_MINB < 30 || L.w > 4000 || L.h > 4000)   // at least one 30-px FAST cell
      return ps_set_error(PS_ERR_INVALID, "image %dx%d: level %d is %dx%d, outside the supported range", w, hgt, l, L.w, L.h);
    L.stride = (int)align_up(L.w + 2 * PS_EDGE, 64);
    L.bstride = (int)align_up(L.w, 64);
    off = align_up(off, 256);
    L.plane_off = (uint32_t)off;
    off += (size_t)L.stride * (L.h + 2 * PS_EDGE);
    // FAST cell grid: ORBextractor.cc:773-787
    const int maxBX = L.w - PS_EDGE + 3, maxBY = L.h - PS_EDGE + 3;
    const float width = (float)(maxBX - PS_MINB), height = (float)(maxBY - PS_MINB);
    L.n_cols = (int)(width / 30.f);
    L.n_rows = (int)(height / 30.f);
    L.w_cell = (int)ceilf(width / L.n_cols);
    L.h_cell = (int)ceilf(height / L.n_rows);
    if (L.w_cell + 6 > PS_FAST_WIN || L.h_cell + 6 > PS_FAST_WIN)
      return ps_set_error(PS_ERR_INVALID, "FAST cell %dx%d exceeds the kernel window", L.w_cell, L.h_cell);
    L.cell_base = cells;
    cells += L.n_cols * L.n_rows;
    if (L.n_cols * L.n_rows > 3 * PS_QT_NCAP)
      return ps_set_error(PS_ERR_INVALID, "level %d has %d cells (> %d)", l, L.n_cols * L.n_rows, 3 * PS_QT_NCAP);
    L.cell_cap = ((L.w_cell + 1) / 2) * ((L.h_cell + 1) / 2);   // strict 3x3 maxima: <= 1 per 2x2 block
    L.cand_off = cand_elems;
    cand_elems += (uint32_t)(L.n_cols * L.n_rows * L.cell_cap);
    L.key_cap = L.n_cols * L.n_rows * L.cell_cap;
    if (L.key_cap >= (1 << 20)) return ps_set_error(PS_ERR_INVALID, "level %d key capacity too large", l);
    L.key_off = key_elems;
    key_elems += 2u * (uint32_t)L.key_cap;
    L.quota = h->quota[l];
    // DistributeOctTree init: ORBextractor.cc:543-545
    L.n_ini = (int)roundf((float)(maxBX - PS_MINB) / (float)(maxBY - PS_MINB));
    if (L.n_ini < 1) return ps_set_error(PS_ERR_INVALID, "level %d is taller than wide; unsupported", l);
    L.h_x = (float)(maxBX - PS_MINB) / (float)L.n_ini;
    if (L.quota + 4 > PS_QT_NCAP || 4 * L.n_ini > PS_QT_NCAP)
      return ps_set_error(PS_ERR_INVALID, "feature quota %d exceeds the quadtree node capacity", L.quota);
    L.sel_off = sel;
    L.sel_cap = (L.quota + 3 > 4 * L.n_ini ? L.quota + 3 : 4 * L.n_ini) + 1;
    L.sel_cap = (L.sel_cap + 3) & ~3;      // levels start at multiples of four slots: a wave of orb_describe (four keypoints) never straddles two
    sel += L.sel_cap;
    L.scale = h->scale[l];
    L.inv_scale = h->inv_scale[l];
    L.kp_size = (float)(int)(31 * h->scale[l]);   // ORBextractor.cc:839: const int scaledPatchSize
    if (l > 0) resize_tables(P.lv[l - 1].w, P.lv[l - 1].h, L.w, L.h, h->tabs_host, &L.xtab_off, &L.ytab_off);
  }
  for (int l = 0; l < nl; l++) {
    OrbLevel& L = P.lv[l];
    off = align_up(off, 256);
    L.blur_off = (uint32_t)off;
    off += (size_t)L.bstride * L.h;
  }
  {
    int nb = 0;
    const int rows_per_block = 4 * psk_orb_blur_rows();
    for (int l = 0; l < nl; l++) {
      P.lv[l].blur_blk_base = nb;
      nb += ((P.lv[l].w + 255) / 256) * ((P.lv[l].h + rows_per_block - 1) / rows_per_block);
    }
    P.blur_blocks = nb;
    nb = 0;
    for (int l = 0; l < nl; l++) {
      const OrbLevel& L = P.lv[l];
      const int PW = L.w + 2 * PS_EDGE, ngx = (PW + 3) / 4, nleft = (PS_EDGE + 3) / 4, nright = ngx - (PS_EDGE + L.w) / 4;
      P.lv[l].border_blk_base = nb;
      nb += (2 * PS_EDGE * ngx + L.h * (nleft + nright) + 255) / 256;
    }
    P.border_blocks = nb;
  }
  P.n_cells = cells;
  {   // per cell: its level and its position in the level's cell grid (orb_fast_cells reads the word with one scalar load instead of
      // searching the level and dividing by the grid width on the scalar unit)
    P.celltab_off = (uint32_t)h->tabs_host.size();
    std::vector<uint32_t> ct((size_t)((cells + 3) & ~3), 0u);
    for (int l = 0; l < nl; l++)
      for (int c = 0; c < P.lv[l].n_cols * P.lv[l].n_rows; c++)
        ct[(size_t)P.lv[l].cell_base + c] = (uint32_t)l | ((uint32_t)(c % P.lv[l].n_cols) << 4) | ((uint32_t)(c / P.lv[l].n_cols) << 18);
    for (size_t i = 0; i < ct.size(); i += 4) h->tabs_host.push_back(make_int4((int)ct[i], (int)ct[i + 1], (int)ct[i + 2], (int)ct[i + 3]));
  }
  P.sel_total = sel;
  P.kp_cap = (int)align_up(sel, 64);
  off = align_up(off, 256); P.cellcnt_off = off; off += (size_t)cells * 4;
  off = align_up(off, 256); P.cand_base = off;   off += (size_t)cand_elems * 4;
  off = align_up(off, 256); P.key_base = off;    off += (size_t)key_elems * 4;
  off = align_up(off, 256); P.sel_base = off;    off += (size_t)sel * 4;
  off = align_up(off, 256); P.selcnt_off = off;  off += PS_ORB_MAX_LEVELS * 4;
  P.ncand_off = off; off += PS_ORB_MAX_LEVELS * 4;
  P.arena_bytes = align_up(off, 4096);
  if (P.arena_bytes >= (1ull << 32)) return ps_set_error(PS_ERR_INVALID, "arena too large");

  free_device(h);
  const int B = h->cfg.max_batch;
  PS_HIP(hipMalloc(&h->d_arena, P.arena_bytes * B));
  {   // PS_DEBUG_FILL (diagnostic): a byte pattern instead of zeros, to prove that nothing depends on the initial content
    const char* fill = getenv("PS_DEBUG_FILL");
    PS_HIP(hipMemsetAsync(h->d_arena, fill ? atoi(fill) : 0, P.arena_bytes * B, h->stream));
  }
  PS_HIP(hipMalloc(&h->d_tabs, (h->tabs_host.size() + 1) * sizeof(int4)));
  if (!h->tabs_host.empty())
    PS_HIP(hipMemcpyAsync(h->d_tabs, h->tabs_host.data(), h->tabs_host.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
  PS_HIP(hipMalloc(&h->d_kps, (size_t)B * P.kp_cap * sizeof(ps_keypoint)));
  PS_HIP(hipMalloc(&h->d_desc, (size_t)B * P.kp_cap * 32));
  PS_HIP(hipMalloc(&h->d_counts, (size_t)B * 4));
  PS_HIP(hipMalloc(&h->d_uright, (size_t)B * P.kp_cap * 4));
  PS_HIP(hipMalloc(&h->d_depth, (size_t)B * P.kp_cap * 4));
  PS_HIP(hipMalloc(&h->d_sad, (size_t)B * P.kp_cap * 4));
  PS_HIP(hipMalloc(&h->d_kept, (size_t)B * 4));
  PS_HIP(hipMalloc(&h->d_pairs, (size_t)B * sizeof(StPair)));
  PS_HIP(hipMalloc(&h->d_stscratch, (size_t)B * PS_ST_SCRATCH));
  PS_HIP(hipMemsetAsync(h->d_counts, 0, (size_t)B * 4, h->stream));
  PS_HIP(hipStreamSynchronize(h->stream));
  h->planned = true;
  return PS_OK;
}

// The batch is processed in chunks of `chunk` images: one chunk's arenas (~7 MB per image) stay resident in the
// 256 MB Infinity Cache between the five stages instead of round-tripping through HBM.
int run_batch(ps_orb* h, const uint8_t* d_imgs, int nimg, int stride, size_t pitch, hipStream_t st) {
  const OrbPlan* P = &h->plan;
  const bool tm = h->timing;
  const int chunk = h->chunk > 0 ? h->chunk : nimg;
  const int nchunks = (nimg + chunk - 1) / chunk;
  if (tm && nchunks > ps_orb::MAXCHUNK) return ps_set_error(PS_ERR_CAPACITY, "stage timing supports at most %d chunks", ps_orb::MAXCHUNK);
  const int slot = h->timed_batches % ps_orb::RING;
  for (int c = 0; c < nchunks; c++) {
    const int i0 = c * chunk, n = nimg - i0 < chunk ? nimg - i0 : chunk;
    uint8_t* arena = h->d_arena + (size_t)i0 * P->arena_bytes;
    const uint8_t* imgs = d_imgs + (size_t)i0 * pitch;
    hipEvent_t* ev = h->ev[slot][c];
    if (tm) PS_HIP(hipEventRecord(ev[0], st));
    if (h->fused) {
      for (int l = 0; l < P->nlevels; l++) psk_orb_launch_level_fused(P, l, arena, imgs, stride, pitch, h->d_tabs, n, st);
    } else {
      for (int l = 0; l < P->nlevels; l++) psk_orb_launch_pyramid(P, l, arena, imgs, stride, pitch, h->d_tabs, n, st);
      psk_orb_launch_border(P, arena, n, st);
    }
    if (c == nchunks - 1 && h->ev_input_free && d_imgs == h->d_img) {   // the staging buffer may be overwritten from here on
      PS_HIP(hipEventRecord(h->ev_input_free, st));
      h->input_read_pending = true;
    }
    if (tm) PS_HIP(hipEventRecord(ev[1], st));
    psk_orb_launch_fast(P, arena, n, h->d_mask ? h->d_mask + (size_t)i0 * h->mask_pitch : nullptr, h->mask_stride, h->mask_pitch, h->d_tabs, st);
    if (tm) PS_HIP(hipEventRecord(ev[2], st));
    psk_orb_launch_quadtree(P, arena, n, st);
    if (tm) PS_HIP(hipEventRecord(ev[3], st));
    if (!h->fused) psk_orb_launch_blur(P, arena, n, st);
    if (tm) PS_HIP(hipEventRecord(ev[4], st));
    psk_orb_launch_describe(P, arena, h->d_kps + (size_t)i0 * P->kp_cap, h->d_desc + (size_t)i0 * P->kp_cap * 32,
                            h->d_counts + i0, n, st);
    if (tm) PS_HIP(hipEventRecord(ev[5], st));
  }
  PS_HIP(hipGetLastError());
  if (tm) { h->timed_chunks[slot] = nchunks; h->timed_batches++; }
  h->last_nimg = nimg;
  h->last_npairs = 0;   // stereo results of the previous batch are gone
  return PS_OK;
}

}  // namespace

extern "C" {

int ps_orb_create(const ps_orb_config* cfg, ps_orb** out) {
  if (!cfg || !out) return ps_set_error(PS_ERR_INVALID, "ps_orb_create: null argument");
  if (cfg->nlevels < 1 || cfg->nlevels > PS_ORB_MAX_LEVELS || cfg->nfeatures < 1 || cfg->scale_factor <= 1.f ||
      cfg->max_batch < 1 || cfg->ini_th_fast < cfg->min_th_fast || cfg->min_th_fast < 1)
    return ps_set_error(PS_ERR_INVALID, "ps_orb_create: unsupported configuration");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return ps_set_error(PS_ERR_NO_DEVICE, "no HIP device visible");
  if (cfg->device < 0 || cfg->device >= ndev) return ps_set_error(PS_ERR_INVALID, "bad device ordinal");
  PS_HIP(hipSetDevice(cfg->device));
  ps_orb* h = new ps_orb();
  h->cfg = *cfg;
  build_tables(h);
  hipError_t e;
#ifdef PS_DEV_CU_PARTITION
  // developer experiment, compiled in with -DPS_DEV_CU_PARTITION only (profiles/r05_cu_partition.txt: rejected): the k-th
  // extractor handle of the process gets its own share of the compute units: PS_CU_PARTITION=N cuts the CU mask bits into N ranges (the driver
  // deals consecutive bits round-robin over the 8 XCDs, so a range is a slice of every XCD), PS_CU_SHARE=k gives a handle k consecutive ranges.
  // The masked stream is a BLOCKING stream (it synchronises with the null stream), and the handle counter is process-wide and not thread-safe.
  if (const char* part = getenv("PS_CU_PARTITION")) {
    static int created = 0;
    const int N = atoi(part) > 0 ? atoi(part) : 1, K = getenv("PS_CU_SHARE") ? atoi(getenv("PS_CU_SHARE")) : 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) { delete h; return ps_set_error(PS_ERR_HIP, "hipGetDeviceProperties failed"); }
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int kk = 0; kk < K; kk++) {
      const int r = (created + kk) % N;
      for (int c = r * ncu / N; c < (r + 1) * ncu / N; c++) mask[c >> 5] |= 1u << (c & 31);
    }
    created++;
    e = hipExtStreamCreateWithCUMask(&h->stream, (uint32_t)mask.size(), mask.data());
  } else
#endif
  e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; return ps_set_error(PS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
  for (int r = 0; r < ps_orb::RING; r++)
    for (int c = 0; c < ps_orb::MAXCHUNK; c++)
      for (int i = 0; i <= ST_COUNT; i++) hipEventCreate(&h->ev[r][c][i]);
  if (!(getenv("PS_ORB_COPY_STREAM") && getenv("PS_ORB_COPY_STREAM")[0] == '0')) {
    if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_input_free, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_uploaded, hipEventDisableTiming) != hipSuccess) {
      ps_orb_destroy(h);
      return ps_set_error(PS_ERR_HIP, "copy stream / event creation failed");
    }
  }
  if (const char* e = getenv("PS_ORB_CHUNK")) h->chunk = atoi(e);
  if (const char* e = getenv("PS_ORB_FUSED")) h->fused = atoi(e) != 0;
  *out = h;
  return PS_OK;
}

void ps_orb_destroy(ps_orb* h) {
  if (!h) return;
  hipSetDevice(h->cfg.device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->copy_stream) { hipStreamSynchronize(h->copy_stream); hipStreamDestroy(h->copy_stream); h->copy_stream = nullptr; }
  if (h->ev_input_free) hipEventDestroy(h->ev_input_free);
  if (h->ev_uploaded) hipEventDestroy(h->ev_uploaded);
  free_device(h);
  if (h->d_img) hipFree(h->d_img);
  if (h->d_mask_buf) hipFree(h->d_mask_buf);
  for (int r = 0; r < ps_orb::RING; r++)
    for (int c = 0; c < ps_orb::MAXCHUNK; c++)
      for (int i = 0; i <= ST_COUNT; i++) if (h->ev[r][c][i]) hipEventDestroy(h->ev[r][c][i]);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

// accessors for track_host.hip (the lockstep tracker owns an extractor and queues its own kernels behind it)
hipStream_t psi_orb_stream(ps_orb* h) { return h->stream; }
const OrbPlan* psi_orb_plan(ps_orb* h) { return &h->plan; }
uint8_t* psi_orb_arena(ps_orb* h) { return h->d_arena; }
int psi_orb_prepare(ps_orb* h, int w, int hgt) {
  PS_HIP(hipSetDevice(h->cfg.device));
  if (!h->planned || h->plan.img_w != w || h->plan.img_h != hgt) return build_plan(h, w, hgt);
  return PS_OK;
}

int ps_orb_get_tables(const ps_orb* h, float* s, float* is, float* s2, float* is2, int32_t* q) {
  if (!h) return ps_set_error(PS_ERR_INVALID, "null handle");
  for (int i = 0; i < h->cfg.nlevels; i++) {
    if (s) s[i] = h->scale[i];
    if (is) is[i] = h->inv_scale[i];
    if (s2) s2[i] = h->sigma2[i];
    if (is2) is2[i] = h->inv_sigma2[i];
    if (q) q[i] = h->quota[i];
  }
  return PS_OK;
}

int ps_orb_level_size(const ps_orb* h, int w, int hgt, int level, int32_t* wl, int32_t* hl) {
  if (!h || level < 0 || level >= h->cfg.nlevels || !wl || !hl) return ps_set_error(PS_ERR_INVALID, "bad argument");
  level_size(h, w, hgt, level, wl, hl);
  return PS_OK;
}

int ps_orb_extract_batch_device(ps_orb* h, const uint8_t* d_imgs, int nimg, int w, int hgt, int stride,
                                size_t image_pitch, void* stream) {
  if (!h || !d_imgs || nimg < 1 || w < 1 || hgt < 1 || stride < w)
    return ps_set_error(PS_ERR_INVALID, "ps_orb_extract_batch_device: bad argument");
  if (nimg > h->cfg.max_batch) return ps_set_error(PS_ERR_CAPACITY, "nimg %d > max_batch %d", nimg, h->cfg.max_batch);
  PS_HIP(hipSetDevice(h->cfg.device));
  if (!h->planned || h->plan.img_w != w || h->plan.img_h != hgt) {
    int rc = build_plan(h, w, hgt);
    if (rc != PS_OK) return rc;
  }
  h->last_stream = stream ? (hipStream_t)stream : h->stream;
  return run_batch(h, d_imgs, nimg, stride, image_pitch, h->last_stream);
}

int ps_orb_batch_device_outputs(const ps_orb* h, const ps_keypoint** d_kps, const uint8_t** d_desc,
                                const int32_t** d_counts, int32_t* kp_capacity) {
  if (!h || !h->planned) return ps_set_error(PS_ERR_INVALID, "no batch has been run");
  if (d_kps) *d_kps = h->d_kps;
  if (d_desc) *d_desc = h->d_desc;
  if (d_counts) *d_counts = h->d_counts;
  if (kp_capacity) *kp_capacity = h->plan.kp_cap;
  return PS_OK;
}

int ps_orb_sync(ps_orb* h) {
  if (!h) return ps_set_error(PS_ERR_INVALID, "null handle");
  PS_HIP(hipSetDevice(h->cfg.device));
  PS_HIP(hipDeviceSynchronize());
  return PS_OK;
}

int ps_orb_batch_fetch(ps_orb* h, int image, ps_keypoint* kps, uint8_t* desc, int cap, int* n) {
  if (!h || !h->planned || image < 0 || image >= h->last_nimg || !n)
    return ps_set_error(PS_ERR_INVALID, "ps_orb_batch_fetch: bad argument");
  PS_HIP(hipSetDevice(h->cfg.device));
  PS_HIP(hipDeviceSynchronize());
  int32_t cnt = 0;
  PS_HIP(hipMemcpy(&cnt, h->d_counts + image, 4, hipMemcpyDeviceToHost));
  *n = cnt;
  if (cnt > cap) return ps_set_error(PS_ERR_CAPACITY, "%d keypoints, caller capacity %d", cnt, cap);
  if (cnt > 0) {
    if (!kps || !desc) return ps_set_error(PS_ERR_INVALID, "null output buffer");
    PS_HIP(hipMemcpy(kps, h->d_kps + (size_t)image * h->plan.kp_cap, (size_t)cnt * sizeof(ps_keypoint), hipMemcpyDeviceToHost));
    PS_HIP(hipMemcpy(desc, h->d_desc + (size_t)image * h->plan.kp_cap * 32, (size_t)cnt * 32, hipMemcpyDeviceToHost));
  }
  return PS_OK;
}

int ps_orb_extract(ps_orb* h, const uint8_t* img, int w, int hgt, int stride, ps_keypoint* kps, uint8_t* desc,
                   int cap, int* n, uint8_t* const* pyramid_out) {
  if (!h || !n) return ps_set_error(PS_ERR_INVALID, "ps_orb_extract: null argument");
  *n = 0;
  if (!img || w <= 0 || hgt <= 0) return PS_OK;   // empty image: silent return (ORBextractor.cc:1046-1047)
  if (stride < w) return ps_set_error(PS_ERR_INVALID, "stride < width");
  PS_HIP(hipSetDevice(h->cfg.device));
  const size_t bytes = (size_t)stride * hgt;
  if (bytes > h->d_img_bytes) {
    if (h->d_img) hipFree(h->d_img);
    h->d_img = nullptr;
    PS_HIP(hipMalloc(&h->d_img, bytes));
    h->d_img_bytes = bytes;
  }
  {
    // a copy from pageable memory is staged by the runtime in pieces with a wait each; from page-locked memory it is one DMA transfer
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, img) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    const uint8_t* src = img;
    if (!pinned) {
      if (bytes > h->h_in_bytes) {
        if (h->h_in) hipHostFree(h->h_in);
        h->h_in = nullptr; h->h_in_bytes = 0;
        PS_HIP(hipHostMalloc(&h->h_in, bytes, hipHostMallocDefault));
        h->h_in_bytes = bytes;
      }
      memcpy(h->h_in, img, bytes);
      src = h->h_in;
    }
    PS_HIP(hipMemcpyAsync(h->d_img, src, bytes, hipMemcpyHostToDevice, h->stream));
  }
  int rc = ps_orb_extract_batch_device(h, h->d_img, 1, w, hgt, stride, bytes, nullptr);
  if (rc != PS_OK) return rc;
  {
    // count, keypoints and descriptors of the one image: three copies into page-locked memory behind the kernels, ONE wait on this handle's
    // stream (ps_orb_batch_fetch waits for the whole device - the other extractor's thread included - and blocks once per array)
    const size_t kc = (size_t)h->plan.kp_cap, need = 64 + kc * (sizeof(ps_keypoint) + 32);
    if (need > h->h_one_bytes) {
      if (h->h_one) hipHostFree(h->h_one);
      h->h_one = nullptr; h->h_one_bytes = 0;
      PS_HIP(hipHostMalloc(&h->h_one, need, hipHostMallocDefault));
      h->h_one_bytes = need;
    }
    hipStream_t st = h->last_stream ? h->last_stream : h->stream;
    uint8_t* hk = h->h_one + 64;
    uint8_t* hd = hk + kc * sizeof(ps_keypoint);
    PS_HIP(hipMemcpyAsync(h->h_one, h->d_counts, 4, hipMemcpyDeviceToHost, st));
    PS_HIP(hipMemcpyAsync(hk, h->d_kps, kc * sizeof(ps_keypoint), hipMemcpyDeviceToHost, st));
    PS_HIP(hipMemcpyAsync(hd, h->d_desc, kc * 32, hipMemcpyDeviceToHost, st));
    PS_HIP(hipStreamSynchronize(st));
    const int cnt = *(const int32_t*)h->h_one;
    *n = cnt;
    if (cnt > cap) return ps_set_error(PS_ERR_CAPACITY, "%d keypoints, caller capacity %d", cnt, cap);
    if (cnt > 0) {
      if (!kps || !desc) return ps_set_error(PS_ERR_INVALID, "null output buffer");
      memcpy(kps, hk, (size_t)cnt * sizeof(ps_keypoint));
      memcpy(desc, hd, (size_t)cnt * 32);
    }
  }
  if (pyramid_out) {
    for (int l = 0; l < h->plan.nlevels; l++) {
      const OrbLevel& L = h->plan.lv[l];
      if (!pyramid_out[l]) continue;
      PS_HIP(hipMemcpy2D(pyramid_out[l], L.w + 2 * PS_EDGE, h->d_arena + L.plane_off, L.stride, L.w + 2 * PS_EDGE,
                         L.h + 2 * PS_EDGE, hipMemcpyDeviceToHost));
    }
  }
  return PS_OK;
}

// The object features of a frame (Frame::ExtractObjORB -> OpencvORBDetector, /root/reference/src/Frame.cc:2623-2665): the
// reference runs OpenCV's own cv::ORB::create(1000, 1.2, 8, 19)->detectAndCompute(im, ObjMask, kp, descriptor).  This entry point
// is the declared stand-in of SURVEY.md 8f-2: THIS extractor's pipeline (a2-a8) on the image, with the FAST keypoints whose
// level-0 pixel lies outside the mask dropped before DistributeOctTree, so that the per-level quotas are spent inside the mask.
int ps_orb_extract_masked(ps_orb* h, const uint8_t* img, const uint8_t* mask, int w, int hgt, int stride, int mask_stride, ps_keypoint* kps,
                          uint8_t* desc, int cap, int* n) {
  if (!h || !n) return ps_set_error(PS_ERR_INVALID, "ps_orb_extract_masked: null argument");
  *n = 0;
  if (!img || w <= 0 || hgt <= 0) return PS_OK;
  if (!mask) return ps_orb_extract(h, img, w, hgt, stride, kps, desc, cap, n, nullptr);   // detectAndCompute with an empty mask
  if (stride < w || mask_stride < w) return ps_set_error(PS_ERR_INVALID, "stride < width");
  PS_HIP(hipSetDevice(h->cfg.device));
  const size_t mbytes = (size_t)mask_stride * hgt;
  if (mbytes > h->d_mask_bytes) {
    PS_HIP(hipStreamSynchronize(h->stream));
    if (h->d_mask_buf) hipFree(h->d_mask_buf);
    h->d_mask_buf = nullptr; h->d_mask_bytes = 0;
    PS_HIP(hipMalloc(&h->d_mask_buf, mbytes));
    h->d_mask_bytes = mbytes;
  }
  PS_HIP(hipMemcpyAsync(h->d_mask_buf, mask, mbytes, hipMemcpyHostToDevice, h->stream));
  h->d_mask = h->d_mask_buf; h->mask_stride = mask_stride; h->mask_pitch = mbytes;
  const int rc = ps_orb_extract(h, img, w, hgt, stride, kps, desc, cap, n, nullptr);
  h->d_mask = nullptr;
  return rc;
}

int ps_orb_extract_batch(ps_orb* h, const uint8_t* const* imgs, int nimg, int w, int hgt, int stride) {
  if (!h || !imgs || nimg < 1 || w < 1 || hgt < 1 || stride < w) return ps_set_error(PS_ERR_INVALID, "ps_orb_extract_batch: bad argument");
  if (nimg > h->cfg.max_batch) return ps_set_error(PS_ERR_CAPACITY, "nimg %d > max_batch %d", nimg, h->cfg.max_batch);
  for (int i = 0; i < nimg; i++)
    if (!imgs[i]) return ps_set_error(PS_ERR_INVALID, "ps_orb_extract_batch: image %d is null", i);
  PS_HIP(hipSetDevice(h->cfg.device));
  const size_t pitch = (size_t)stride * hgt, bytes = pitch * nimg;
  if (bytes > h->d_img_bytes) {
    PS_HIP(hipDeviceSynchronize());
    if (h->d_img) hipFree(h->d_img);
    h->d_img = nullptr;
    PS_HIP(hipMalloc(&h->d_img, bytes));
    h->d_img_bytes = bytes;
  }
  // uploads asynchronous when the caller's buffers are pinned (ps_pinned_alloc), on the copy stream: behind the last reader of the
  // staging buffer, ahead of this batch's kernels.  Images that follow each other in host memory go in ONE transfer: a 0.47 MB
  // copy reaches half the PCIe rate of a multi-megabyte one.
  hipStream_t cs = h->copy_stream ? h->copy_stream : h->stream;
  if (h->copy_stream && h->input_read_pending) PS_HIP(hipStreamWaitEvent(cs, h->ev_input_free, 0));
  for (int i = 0; i < nimg;) {
    int j = i + 1;
    while (j < nimg && imgs[j] == imgs[j - 1] + pitch) j++;
    PS_HIP(hipMemcpyAsync(h->d_img + pitch * i, imgs[i], pitch * (size_t)(j - i), hipMemcpyHostToDevice, cs));
    i = j;
  }
  if (h->copy_stream) {
    PS_HIP(hipEventRecord(h->ev_uploaded, cs));
    PS_HIP(hipStreamWaitEvent(h->stream, h->ev_uploaded, 0));
  }
  return ps_orb_extract_batch_device(h, h->d_img, nimg, w, hgt, stride, pitch, nullptr);
}

int ps_orb_stereo_fetch_frames(ps_orb* h, ps_stereo_frame* frames, int npairs) {
  if (!h || !h->planned || !frames || npairs < 1 || npairs > h->last_npairs || 2 * npairs > h->last_nimg)
    return ps_set_error(PS_ERR_INVALID, "ps_orb_stereo_fetch_frames: needs ps_orb_stereo_match_batch over >= npairs pairs");
  PS_HIP(hipSetDevice(h->cfg.device));
  const OrbPlan& P = h->plan;
  const size_t cap = P.kp_cap;
  // pinned staging: counts + kept, then per pair keypoints | descriptors | uRight | depth at full capacity
  const size_t o_cnt = 0, o_kept = (size_t)h->last_nimg * 4, o_data = (o_kept + (size_t)npairs * 4 + 255) & ~(size_t)255;
  const size_t per = cap * (sizeof(ps_keypoint) + 32 + 4 + 4), need = o_data + per * npairs;
  if (need > h->h_frames_bytes) {
    if (h->h_frames) hipHostFree(h->h_frames);
    h->h_frames = nullptr;
    PS_HIP(hipHostMalloc(&h->h_frames, need, hipHostMallocDefault));
    h->h_frames_bytes = need;
  }
  uint8_t* H = h->h_frames;
  PS_HIP(hipMemcpyAsync(H + o_cnt, h->d_counts, (size_t)h->last_nimg * 4, hipMemcpyDeviceToHost, h->stream));
  PS_HIP(hipMemcpyAsync(H + o_kept, h->d_kept, (size_t)npairs * 4, hipMemcpyDeviceToHost, h->stream));
  // four transfers for the whole batch: the left images are every second row of the [image][kp_cap] result arrays
  uint8_t* B0 = H + o_data;
  PS_HIP(hipMemcpy2DAsync(B0, per, h->d_kps, 2 * cap * sizeof(ps_keypoint), cap * sizeof(ps_keypoint), npairs, hipMemcpyDeviceToHost, h->stream));
  PS_HIP(hipMemcpy2DAsync(B0 + cap * sizeof(ps_keypoint), per, h->d_desc, 2 * cap * 32, cap * 32, npairs, hipMemcpyDeviceToHost, h->stream));
  PS_HIP(hipMemcpy2DAsync(B0 + cap * (sizeof(ps_keypoint) + 32), per, h->d_uright, cap * 4, cap * 4, npairs, hipMemcpyDeviceToHost, h->stream));
  PS_HIP(hipMemcpy2DAsync(B0 + cap * (sizeof(ps_keypoint) + 36), per, h->d_depth, cap * 4, cap * 4, npairs, hipMemcpyDeviceToHost, h->stream));
  PS_HIP(hipStreamSynchronize(h->stream));
  const int32_t* cnt = reinterpret_cast<const int32_t*>(H + o_cnt);
  const int32_t* kept = reinterpret_cast<const int32_t*>(H + o_kept);
  for (int k = 0; k < npairs; k++) {
    ps_stereo_frame& F = frames[k];
    const int n = cnt[2 * k];
    F.n = n; F.n_right = cnt[2 * k + 1]; F.kept = kept[k];
    if (n > F.cap) return ps_set_error(PS_ERR_CAPACITY, "pair %d: %d keypoints, caller capacity %d", k, n, F.cap);
    if (n > 0 && (!F.kps || !F.desc || !F.u_right || !F.depth)) return ps_set_error(PS_ERR_INVALID, "pair %d: null output buffer", k);
  }
  ps_parallel_for(npairs, per * npairs / 2, [&](int k) {
    ps_stereo_frame& F = frames[k];
    const size_t n = (size_t)F.n;
    if (n == 0) return;
    const uint8_t* B = H + o_data + per * k;
    memcpy(F.kps, B, n * sizeof(ps_keypoint));
    memcpy(F.desc, B + cap * sizeof(ps_keypoint), n * 32);
    memcpy(F.u_right, B + cap * (sizeof(ps_keypoint) + 32), n * 4);
    memcpy(F.depth, B + cap * (sizeof(ps_keypoint) + 36), n * 4);
  });
  return PS_OK;
}

int ps_orb_debug_read(ps_orb* h, int image, int level, int what, void* out, size_t out_bytes, int* n) {
  if (!h || !h->planned || image < 0 || image >= h->cfg.max_batch || level < 0 || level >= h->plan.nlevels || !out)
    return ps_set_error(PS_ERR_INVALID, "ps_orb_debug_read: bad argument");
  PS_HIP(hipSetDevice(h->cfg.device));
  PS_HIP(hipDeviceSynchronize());
  const OrbPlan& P = h->plan;
  const OrbLevel& L = P.lv[level];
  const uint8_t* base = h->d_arena + (size_t)image * P.arena_bytes;
  if (what == 0) {
    const size_t pw = L.w + 2 * PS_EDGE, ph = L.h + 2 * PS_EDGE;
    if (out_bytes < pw * ph) return ps_set_error(PS_ERR_CAPACITY, "buffer too small");
    PS_HIP(hipMemcpy2D(out, pw, base + L.plane_off, L.stride, pw, ph, hipMemcpyDeviceToHost));
    if (n) *n = (int)(pw * ph);
  } else if (what == 1) {
    if (out_bytes < (size_t)L.w * L.h) return ps_set_error(PS_ERR_CAPACITY, "buffer too small");
    PS_HIP(hipMemcpy2D(out, L.w, base + L.blur_off, L.bstride, L.w, L.h, hipMemcpyDeviceToHost));
    if (n) *n = L.w * L.h;
  } else if (what == 2) {
    const int ncell = L.n_cols * L.n_rows;
    std::vector<int32_t> cnt(ncell);
    std::vector<uint32_t> slots((size_t)ncell * L.cell_cap);
    PS_HIP(hipMemcpy(cnt.data(), base + P.cellcnt_off + (size_t)L.cell_base * 4, (size_t)ncell * 4, hipMemcpyDeviceToHost));
    PS_HIP(hipMemcpy(slots.data(), base + P.cand_base + (size_t)L.cand_off * 4, slots.size() * 4, hipMemcpyDeviceToHost));
    int32_t* o = (int32_t*)out;
    size_t k = 0;
    for (int c = 0; c < ncell; c++)
      for (int i = 0; i < cnt[c]; i++) {
        if ((k + 1) * 12 > out_bytes) return ps_set_error(PS_ERR_CAPACITY, "buffer too small");
        const uint32_t e = slots[(size_t)c * L.cell_cap + i];
        o[k * 3] = e & 0xFFF; o[k * 3 + 1] = (e >> 12) & 0xFFF; o[k * 3 + 2] = (int)(e >> 24) - 1;
        k++;
      }
    if (n) *n = (int)k;
  } else if (what == 3) {
    int32_t cnts[PS_ORB_MAX_LEVELS];
    PS_HIP(hipMemcpy(cnts, base + P.selcnt_off, sizeof(cnts), hipMemcpyDeviceToHost));
    const int c = cnts[level];
    if ((size_t)c * 12 > out_bytes) return ps_set_error(PS_ERR_CAPACITY, "buffer too small");
    std::vector<uint32_t> sel(c > 0 ? c : 1);
    if (c > 0) PS_HIP(hipMemcpy(sel.data(), base + P.sel_base + (size_t)L.sel_off * 4, (size_t)c * 4, hipMemcpyDeviceToHost));
    int32_t* o = (int32_t*)out;
    for (int i = 0; i < c; i++) {
      o[i * 3] = sel[i] & 0xFFF; o[i * 3 + 1] = (sel[i] >> 12) & 0xFFF; o[i * 3 + 2] = (int)(sel[i] >> 24) - 1;
    }
    if (n) *n = c;
  } else {
    return ps_set_error(PS_ERR_INVALID, "unknown `what` %d", what);
  }
  return PS_OK;
}

int ps_orb_enable_stage_timing(ps_orb* h, int enable) {
  if (!h) return ps_set_error(PS_ERR_INVALID, "null handle");
  h->timing = enable != 0;
  h->timed_batches = 0;
  return PS_OK;
}

int ps_orb_stage_times(ps_orb* h, const char** names, float* ms, int cap, int* n) {
  if (!h || !n) return ps_set_error(PS_ERR_INVALID, "null argument");
  if (h->timed_batches <= 0) return ps_set_error(PS_ERR_INVALID, "no batch was run with stage timing enabled");
  PS_HIP(hipSetDevice(h->cfg.device));
  PS_HIP(hipDeviceSynchronize());
  const int nb = h->timed_batches < ps_orb::RING ? h->timed_batches : ps_orb::RING;
  *n = ST_COUNT;
  for (int i = 0; i < ST_COUNT && i < cap; i++) {
    if (names) names[i] = h->fused ? kStageNamesFused[i] : kStageNames[i];
    if (ms) {
      double acc = 0;   // per batch: sum over its chunks; then the mean over the recorded batches
      for (int r = 0; r < nb; r++)
        for (int c = 0; c < h->timed_chunks[r]; c++) {
          float t = 0;
          PS_HIP(hipEventElapsedTime(&t, h->ev[r][c][i], h->ev[r][c][i + 1]));
          acc += t;
        }
      ms[i] = (float)(acc / nb);
    }
  }
  return PS_OK;
}


// ---- Frame::ComputeStereoMatches (Frame.cc:2142-2316) on device-resident extraction results ----
static int stereo_run(ps_orb* out_h, const std::vector<StPair>& pairs, float mb, float mbf, int max_left = 0) {
  // the pair table is the same from call to call in a tracking loop: uploaded only when it changes (a copy from pageable memory
  // would make the caller wait for everything queued on the stream)
  if (out_h->pairs_host.size() != pairs.size() || memcmp(out_h->pairs_host.data(), pairs.data(), pairs.size() * sizeof(StPair)) != 0) {
    PS_HIP(hipStreamSynchronize(out_h->stream));   // an earlier launch may still read the old table
    out_h->pairs_host = pairs;
    PS_HIP(hipMemcpy(out_h->d_pairs, pairs.data(), pairs.size() * sizeof(StPair), hipMemcpyHostToDevice));
  }
  psk_stereo_launch(&out_h->plan, out_h->d_pairs, (int)pairs.size(), max_left > 0 ? max_left : out_h->plan.kp_cap, mb, mbf, out_h->stream);
  PS_HIP(hipGetLastError());
  out_h->last_npairs = (int)pairs.size();
  return PS_OK;
}

int ps_orb_stereo_match_batch(ps_orb* h, int npairs, float mb, float mbf) {
  if (!h || !h->planned || npairs < 1 || 2 * npairs > h->last_nimg || !(mb > 0) || !(mbf > 0))
    return ps_set_error(PS_ERR_INVALID, "ps_orb_stereo_match_batch: needs a batch with 2 * npairs images (left, right interleaved)");
  if (h->plan.kp_cap > 4096) return ps_set_error(PS_ERR_CAPACITY, "stereo matcher supports at most 4096 keypoints per image");
  PS_HIP(hipSetDevice(h->cfg.device));
  // the matcher runs on the handle's stream: ordered behind the extraction unless that ran on a caller stream
  if (h->last_stream && h->last_stream != h->stream) PS_HIP(hipStreamSynchronize(h->last_stream));
  const OrbPlan& P = h->plan;
  std::vector<StPair> pairs(npairs);
  for (int k = 0; k < npairs; k++) {
    const size_t l = 2 * (size_t)k, r = l + 1;
    StPair& s = pairs[k];
    s.arena_l = h->d_arena + l * P.arena_bytes; s.arena_r = h->d_arena + r * P.arena_bytes;
    s.kps_l = h->d_kps + l * P.kp_cap; s.desc_l = h->d_desc + l * P.kp_cap * 32; s.cnt_l = h->d_counts + l;
    s.kps_r = h->d_kps + r * P.kp_cap; s.desc_r = h->d_desc + r * P.kp_cap * 32; s.cnt_r = h->d_counts + r;
    s.u_right = h->d_uright + (size_t)k * P.kp_cap; s.depth = h->d_depth + (size_t)k * P.kp_cap;
    s.sad = h->d_sad + (size_t)k * P.kp_cap; s.kept = h->d_kept + k;
    s.scratch = h->d_stscratch + (size_t)k * PS_ST_SCRATCH;
  }
  return stereo_run(h, pairs, mb, mbf);
}

int ps_orb_stereo_device_outputs(const ps_orb* h, const float** d_uright, const float** d_depth, const int32_t** d_kept) {
  if (!h || !h->planned) return ps_set_error(PS_ERR_INVALID, "no batch has been run");
  if (d_uright) *d_uright = h->d_uright;
  if (d_depth) *d_depth = h->d_depth;
  if (d_kept) *d_kept = h->d_kept;
  return PS_OK;
}

int ps_orb_stereo_fetch(ps_orb* h, int pair, float* u_right, float* depth, int cap, int* n_left, int* kept) {
  if (!h || !h->planned || pair < 0 || pair >= h->last_npairs || !n_left)
    return ps_set_error(PS_ERR_INVALID, "ps_orb_stereo_fetch: bad argument");
  PS_HIP(hipSetDevice(h->cfg.device));
  PS_HIP(hipDeviceSynchronize());
  // the left image of pair k is image 2k of a batch, image 0 of a two-handle call
  int32_t n = 0, kp = 0;
  const int limg = h->last_npairs == 1 && h->last_nimg == 1 ? 0 : 2 * pair;
  PS_HIP(hipMemcpy(&n, h->d_counts + limg, 4, hipMemcpyDeviceToHost));
  PS_HIP(hipMemcpy(&kp, h->d_kept + pair, 4, hipMemcpyDeviceToHost));
  *n_left = n;
  if (kept) *kept = kp;
  if (n > cap) return ps_set_error(PS_ERR_CAPACITY, "%d keypoints, caller capacity %d", n, cap);
  if (n > 0) {
    if (!u_right || !depth) return ps_set_error(PS_ERR_INVALID, "null output buffer");
    PS_HIP(hipMemcpy(u_right, h->d_uright + (size_t)pair * h->plan.kp_cap, (size_t)n * 4, hipMemcpyDeviceToHost));
    PS_HIP(hipMemcpy(depth, h->d_depth + (size_t)pair * h->plan.kp_cap, (size_t)n * 4, hipMemcpyDeviceToHost));
  }
  return PS_OK;
}

int ps_orb_stereo_match_pair(ps_orb* left, ps_orb* right, float mb, float mbf, float* u_right, float* depth, int cap, int* n_left) {
  if (!left || !right || !left->planned || !right->planned || !n_left || !(mb > 0) || !(mbf > 0))
    return ps_set_error(PS_ERR_INVALID, "ps_orb_stereo_match_pair: both extractors must have processed an image");
  if (left->cfg.device != right->cfg.device || left->plan.img_w != right->plan.img_w || left->plan.img_h != right->plan.img_h ||
      left->plan.nlevels != right->plan.nlevels)
    return ps_set_error(PS_ERR_INVALID, "left and right extractor differ in device, image size or level count");
  if (left->plan.kp_cap > 4096) return ps_set_error(PS_ERR_CAPACITY, "stereo matcher supports at most 4096 keypoints per image");
  PS_HIP(hipSetDevice(left->cfg.device));
  // both extractions must be complete (ps_orb_extract returns with its stream idle; a batch queued with ps_orb_extract_batch_device may not be)
  PS_HIP(hipStreamSynchronize(left->last_stream ? left->last_stream : left->stream));
  PS_HIP(hipStreamSynchronize(right->last_stream ? right->last_stream : right->stream));
  std::vector<StPair> pairs(1);
  StPair& s = pairs[0];
  s.arena_l = left->d_arena; s.arena_r = right->d_arena;
  s.kps_l = left->d_kps; s.desc_l = left->d_desc; s.cnt_l = left->d_counts;
  s.kps_r = right->d_kps; s.desc_r = right->d_desc; s.cnt_r = right->d_counts;
  s.u_right = left->d_uright; s.depth = left->d_depth; s.sad = left->d_sad; s.kept = left->d_kept;
  s.scratch = left->d_stscratch;
  int rc = stereo_run(left, pairs, mb, mbf);
  if (rc != PS_OK) return rc;
  left->last_nimg = 1;
  {
    // counts and the two result arrays in one transfer group and one wait (see ps_orb_extract)
    ps_orb* h = left;
    const size_t kc = (size_t)h->plan.kp_cap, need = 64 + kc * (sizeof(ps_keypoint) + 32);   // (the size ps_orb_extract keeps: one block serves both)
    if (need > h->h_one_bytes) {
      if (h->h_one) hipHostFree(h->h_one);
      h->h_one = nullptr; h->h_one_bytes = 0;
      PS_HIP(hipHostMalloc(&h->h_one, need, hipHostMallocDefault));
      h->h_one_bytes = need;
    }
    float* hu = (float*)(h->h_one + 64);
    float* hd = hu + kc;
    PS_HIP(hipMemcpyAsync(h->h_one, h->d_counts, 4, hipMemcpyDeviceToHost, h->stream));
    PS_HIP(hipMemcpyAsync(hu, h->d_uright, kc * 4, hipMemcpyDeviceToHost, h->stream));
    PS_HIP(hipMemcpyAsync(hd, h->d_depth, kc * 4, hipMemcpyDeviceToHost, h->stream));
    PS_HIP(hipStreamSynchronize(h->stream));
    const int nl = *(const int32_t*)h->h_one;
    *n_left = nl;
    if (nl > cap) return ps_set_error(PS_ERR_CAPACITY, "%d keypoints, caller capacity %d", nl, cap);
    if (nl > 0) {
      if (!u_right || !depth) return ps_set_error(PS_ERR_INVALID, "null output buffer");
      memcpy(u_right, hu, (size_t)nl * 4);
      memcpy(depth, hd, (size_t)nl * 4);
    }
  }
  return PS_OK;
}


// Frame::ComputeObjStereoMatches (Frame.cc:2318-2503): the same matcher on caller-provided key sets (the object features of the
// frame, mvTempObjKeys / mvTempObjKeysRight) against the two handles' device-resident pyramids.
int ps_orb_stereo_match_keys(ps_orb* left, ps_orb* right, const ps_keypoint* kps_l, const uint8_t* desc_l, int n_left,
                             const ps_keypoint* kps_r, const uint8_t* desc_r, int n_right, float mb, float mbf, float* u_right, float* depth,
                             int* kept) {
  if (!left || !right || !left->planned || !right->planned || n_left < 0 || n_right < 0 || !(mb > 0) || !(mbf > 0))
    return ps_set_error(PS_ERR_INVALID, "ps_orb_stereo_match_keys: both extractors must have processed an image");
  if (left->cfg.device != right->cfg.device || left->plan.img_w != right->plan.img_w || left->plan.img_h != right->plan.img_h ||
      left->plan.nlevels != right->plan.nlevels)
    return ps_set_error(PS_ERR_INVALID, "left and right extractor differ in device, image size or level count");
  if (kept) *kept = 0;
  if (n_left == 0) return PS_OK;                       // Frame.cc:2320
  const int CAP = 4096;
  if (n_left > CAP || n_right > CAP) return ps_set_error(PS_ERR_CAPACITY, "stereo matcher supports at most 4096 keypoints per image");
  if (!kps_l || !desc_l || !u_right || !depth || (n_right > 0 && (!kps_r || !desc_r))) return ps_set_error(PS_ERR_INVALID, "null buffer");
  for (int i = 0; i < n_left + n_right; i++) {
    const ps_keypoint& k = i < n_left ? kps_l[i] : kps_r[i - n_left];
    if (k.octave < 0 || k.octave >= left->plan.nlevels) return ps_set_error(PS_ERR_INVALID, "keypoint %d: octave %d", i, k.octave);
  }
  PS_HIP(hipSetDevice(left->cfg.device));
  PS_HIP(hipDeviceSynchronize());
  // scratch layout: [kpsL][kpsR][descL][descR][cntL, cntR, kept, pad][uright][depth][sad]
  const size_t o_kl = 0, o_kr = o_kl + (size_t)CAP * 28, o_dl = o_kr + (size_t)CAP * 28, o_dr = o_dl + (size_t)CAP * 32,
               o_cnt = o_dr + (size_t)CAP * 32, o_ur = o_cnt + 64, o_dp = o_ur + (size_t)CAP * 4, o_sad = o_dp + (size_t)CAP * 4,
               total = o_sad + (size_t)CAP * 4;
  if (!left->d_objkeys) {
    PS_HIP(hipMalloc(&left->d_objkeys, total));
    PS_HIP(hipHostMalloc(&left->h_objkeys, total));
  }
  uint8_t* hb = left->h_objkeys; uint8_t* db = left->d_objkeys;
  memcpy(hb + o_kl, kps_l, (size_t)n_left * 28);
  memcpy(hb + o_dl, desc_l, (size_t)n_left * 32);
  if (n_right > 0) { memcpy(hb + o_kr, kps_r, (size_t)n_right * 28); memcpy(hb + o_dr, desc_r, (size_t)n_right * 32); }
  int32_t* cnt = (int32_t*)(hb + o_cnt);
  cnt[0] = n_left; cnt[1] = n_right; cnt[2] = 0;
  PS_HIP(hipMemcpyAsync(db, hb, o_ur, hipMemcpyHostToDevice, left->stream));
  std::vector<StPair> pairs(1);
  StPair& s = pairs[0];
  s.arena_l = left->d_arena; s.arena_r = right->d_arena;
  s.kps_l = db + o_kl; s.desc_l = db + o_dl; s.cnt_l = (const int32_t*)(db + o_cnt);
  s.kps_r = db + o_kr; s.desc_r = db + o_dr; s.cnt_r = (const int32_t*)(db + o_cnt) + 1;
  s.u_right = (float*)(db + o_ur); s.depth = (float*)(db + o_dp); s.sad = (int32_t*)(db + o_sad); s.kept = (int32_t*)(db + o_cnt) + 2;
  s.scratch = left->d_stscratch;
  const int keep_npairs = left->last_npairs;
  int rc = stereo_run(left, pairs, mb, mbf, n_left);
  left->last_npairs = keep_npairs;                     // the frame's own ComputeStereoMatches results stay fetchable
  if (rc != PS_OK) return rc;
  PS_HIP(hipMemcpyAsync(hb + o_cnt, db + o_cnt, total - o_cnt, hipMemcpyDeviceToHost, left->stream));
  PS_HIP(hipStreamSynchronize(left->stream));
  memcpy(u_right, hb + o_ur, (size_t)n_left * 4);
  memcpy(depth, hb + o_dp, (size_t)n_left * 4);
  if (kept) *kept = cnt[2];
  return PS_OK;
}

}  // extern "C"
