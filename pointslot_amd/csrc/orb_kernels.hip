// CDNA4 (gfx950) kernels of the ORB front-end.  Wave = 64 lanes everywhere.
//
// Pipeline per batch of images (all images of a batch in one launch, image = blockIdx.z / .y):
//   orb_pyramid_level   x nlevels  cascaded bilinear pyramid, writes the 19-px REFLECT_101 border too
//   orb_fast_cells      x 1        one workgroup per 30-px FAST cell: score map in LDS, per-cell NMS,
//                                  threshold fallback, raster-ordered compaction into the cell's slots
//   orb_quadtree        x 1        one workgroup per (image, level): DistributeOctTree as parallel
//                                  key passes + node-level list bookkeeping in LDS
//   orb_blur            x 1        7x7 sigma-2 fixed-point Gaussian, register sliding window
//   orb_describe        x 1        one wave per keypoint: intensity-centroid angle + 256-bit rBRIEF
//
// Reference semantics each kernel reproduces are cited at the kernel.  Integer stages are exact;
// float expressions use __f*_rn intrinsics so that hipcc cannot contract them into FMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "orb_plan.h"

namespace {

__device__ __forceinline__ int reflect101(int p, int len) {
  // cv::borderInterpolate(BORDER_REFLECT_101); the border (19) is smaller than any level here, but
  // loop anyway so tiny levels stay correct.
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// ------------------------------------------------------------------------------------------------
// Pyramid level: /root/reference/src/ORBextractor.cc:1107-1132 (ComputePyramid) with OpenCV 3.4
// resize(INTER_LINEAR) 8UC1 fixed-point arithmetic and copyMakeBorder(REFLECT_101).
// One thread = 4 consecutive bytes of the padded plane (one dword store).
// xtab[dx] = {sx0, sx1, a0, a1}, ytab[dy] = {sy0, sy1, b0, b1} are built on the host exactly as
// OpenCV builds xofs/ialpha/yofs/ibeta (float maths, saturate_cast<short>(c * 2048)).
// ------------------------------------------------------------------------------------------------
template <bool LEVEL0>
__global__ __launch_bounds__(256) void orb_pyramid_level(OrbPlan plan, int level, uint8_t* arena,
                                                        const uint8_t* imgs, int img_stride,
                                                        size_t img_pitch, const int4* tabs) {
  const OrbLevel L = plan.lv[level];
  const int img = blockIdx.z;
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  const int PW = L.w + 2 * PS_EDGE, PH = L.h + 2 * PS_EDGE;
  const int px4 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int py = blockIdx.y * 4 + threadIdx.y;
  if (px4 >= PW || py >= PH) return;
  const int y = reflect101(py - PS_EDGE, L.h);
  uint32_t packed = 0;
  if (LEVEL0) {
    const uint8_t* src = imgs + (size_t)img * img_pitch + (size_t)y * img_stride;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int px = px4 + k;
      uint32_t v = 0;
      if (px < PW) v = src[reflect101(px - PS_EDGE, L.w)];
      packed |= v << (8 * k);
    }
  } else {
    const OrbLevel S = plan.lv[level - 1];
    const uint8_t* src = base + S.plane_off + (size_t)PS_EDGE * S.stride + PS_EDGE;
    const int4 ty = tabs[L.ytab_off + y];
    const uint8_t* r0 = src + (size_t)ty.x * S.stride;
    const uint8_t* r1 = src + (size_t)ty.y * S.stride;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int px = px4 + k;
      uint32_t v = 0;
      if (px < PW) {
        const int4 tx = tabs[L.xtab_off + reflect101(px - PS_EDGE, L.w)];
        const int h0 = (int)r0[tx.x] * tx.z + (int)r0[tx.y] * tx.w;
        const int h1 = (int)r1[tx.x] * tx.z + (int)r1[tx.y] * tx.w;
        int o = (((ty.z * (h0 >> 4)) >> 16) + ((ty.w * (h1 >> 4)) >> 16) + 2) >> 2;
        o = o < 0 ? 0 : (o > 255 ? 255 : o);
        v = (uint32_t)o;
      }
      packed |= v << (8 * k);
    }
  }
  *reinterpret_cast<uint32_t*>(base + L.plane_off + (size_t)py * L.stride + px4) = packed;
}

// ------------------------------------------------------------------------------------------------
// FAST cells: ORBextractor.cc:765-829 + OpenCV 3.4 FAST_t<16>/cornerScore<16> with NMS.
//
// For a pixel let s = max over the 16 nine-pixel arcs of min(|signed diff|) (dark or bright arc).
// The pixel is a corner at threshold t iff s > t and OpenCV's score is s - 1, for ANY t (the score
// does not depend on t once the pixel is a corner).  A keypoint at threshold t is a corner whose
// score is strictly greater than the 8 neighbours' scores, where neighbours outside the cell's
// candidate area [3, w-4] x [3, h-4] score 0 (FAST runs on the cell ROI).  Hence:
//   keypoint_t(p) = s(p) > t  AND  s(p) > s(n) for all in-cell neighbours n,
// the cell uses t = iniThFAST unless that yields no keypoint, then minThFAST (ORBextractor.cc:809-816).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int fast_score(const uint8_t* c, int ts, int min_th) {
  // ring in OpenCV's makeOffsets order
  const int v = c[0];
  int d[16];
  d[0] = v - c[3 * ts];      d[1] = v - c[3 * ts + 1];  d[2] = v - c[2 * ts + 2];  d[3] = v - c[ts + 3];
  d[4] = v - c[3];           d[5] = v - c[-ts + 3];     d[6] = v - c[-2 * ts + 2]; d[7] = v - c[-3 * ts + 1];
  d[8] = v - c[-3 * ts];     d[9] = v - c[-3 * ts - 1]; d[10] = v - c[-2 * ts - 2]; d[11] = v - c[-ts - 3];
  d[12] = v - c[-3];         d[13] = v - c[ts - 3];     d[14] = v - c[2 * ts - 2]; d[15] = v - c[3 * ts - 1];
  uint32_t md = 0, mb = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    md |= (uint32_t)(d[k] > min_th) << k;
    mb |= (uint32_t)(d[k] < -min_th) << k;
  }
  auto arc9 = [](uint32_t m16) {
    uint32_t m = m16 | (m16 << 16);
    uint32_t r = m & (m >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m >> 8;
    return r & 0xFFFFu;
  };
  const uint32_t rd = arc9(md), rb = arc9(mb);
  if ((rd | rb) == 0) return 0;
  // exact: A = max_i min(d[i..i+8]), B = max_i min(-d[i..i+8]) = -min_i max(d[i..i+8])
  int lo2[16], hi2[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    lo2[i] = min(d[i], d[(i + 1) & 15]);
    hi2[i] = max(d[i], d[(i + 1) & 15]);
  }
  int lo4[16], hi4[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    lo4[i] = min(lo2[i], lo2[(i + 2) & 15]);
    hi4[i] = max(hi2[i], hi2[(i + 2) & 15]);
  }
  int A = -256, Bn = 256;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int lo9 = min(min(lo4[i], lo4[(i + 4) & 15]), d[(i + 8) & 15]);
    const int hi9 = max(max(hi4[i], hi4[(i + 4) & 15]), d[(i + 8) & 15]);
    A = max(A, lo9);
    Bn = min(Bn, hi9);
  }
  return max(A, -Bn);
}

#define FAST_T 256
__global__ __launch_bounds__(FAST_T) void orb_fast_cells(OrbPlan plan, uint8_t* arena) {
  __shared__ uint8_t tile[PS_FAST_WIN * PS_FAST_WIN];
  __shared__ uint8_t smap[(PS_FAST_WIN - 4) * (PS_FAST_WIN - 4)];   // (cw+2) x (ch+2), zero ring
  __shared__ uint8_t flag[(PS_FAST_WIN - 6) * (PS_FAST_WIN - 6)];
  __shared__ int wsum[FAST_T / 64];
  __shared__ int tcount[FAST_T];
  __shared__ int n20;

  const int cell = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  int level = 0;
#pragma unroll
  for (int l = 1; l < PS_ORB_MAX_LEVELS; l++)
    if (l < plan.nlevels && cell >= plan.lv[l].cell_base) level = l;
  const OrbLevel L = plan.lv[level];
  const int ci = cell - L.cell_base;
  const int ci_y = ci / L.n_cols, ci_x = ci - ci_y * L.n_cols;
  int32_t* cellcnt = reinterpret_cast<int32_t*>(base + plan.cellcnt_off);
  const int maxBX = L.w - PS_MINB, maxBY = L.h - PS_MINB;
  const int iniX = PS_MINB + ci_x * L.w_cell, iniY = PS_MINB + ci_y * L.h_cell;
  const int maxX = min(iniX + L.w_cell + 6, maxBX), maxY = min(iniY + L.h_cell + 6, maxBY);
  const int ww = maxX - iniX, wh = maxY - iniY;   // FAST ROI
  const int cw = ww - 6, ch = wh - 6;             // candidate area
  if (iniX >= maxBX - 3 || iniY >= maxBY - 3 || cw <= 0 || ch <= 0) {
    if (tid == 0) cellcnt[cell] = 0;
    return;
  }
  const int ts = PS_FAST_WIN, ss = cw + 2;
  const uint8_t* plane = base + L.plane_off + (size_t)(PS_EDGE + iniY) * L.stride + PS_EDGE + iniX;
  for (int i = tid; i < ww * wh; i += FAST_T) {
    const int y = i / ww, x = i - y * ww;
    tile[y * ts + x] = plane[(size_t)y * L.stride + x];
  }
  for (int i = tid; i < ss * (ch + 2); i += FAST_T) smap[i] = 0;
  if (tid == 0) n20 = 0;
  __syncthreads();
  const int npx = cw * ch;
  for (int p = tid; p < npx; p += FAST_T) {
    const int y = p / cw, x = p - y * cw;
    const int s = fast_score(&tile[(y + 3) * ts + x + 3], ts, plan.min_th);
    smap[(y + 1) * ss + x + 1] = (uint8_t)s;
  }
  __syncthreads();
  int my20 = 0;
  for (int p = tid; p < npx; p += FAST_T) {
    const int y = p / cw, x = p - y * cw;
    const uint8_t* m = &smap[(y + 1) * ss + x + 1];
    const int s = m[0];
    uint8_t f = 0;
    if (s > 0) {
      const bool lmax = s > m[-1] && s > m[1] && s > m[-ss - 1] && s > m[-ss] && s > m[-ss + 1] &&
                        s > m[ss - 1] && s > m[ss] && s > m[ss + 1];
      if (lmax) {
        f = 1;
        if (s > plan.ini_th) { f = 3; my20++; }
      }
    }
    flag[p] = f;
  }
  if (my20) atomicAdd(&n20, my20);
  __syncthreads();
  const uint8_t want = n20 > 0 ? 2 : 1;
  // raster-ordered compaction: thread t owns pixels [t*per, (t+1)*per)
  const int per = (npx + FAST_T - 1) / FAST_T;
  const int b = tid * per, e = min(b + per, npx);
  int mine = 0;
  for (int p = b; p < e; p++) mine += (flag[p] & want) ? 1 : 0;
  // block exclusive scan of `mine`
  int incl = mine;
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const int o = __shfl_up(incl, dlt);
    if (lane >= dlt) incl += o;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int k = 0; k < FAST_T / 64; k++) {
    if (k < wv) woff += wsum[k];
    total += wsum[k];
  }
  int pos = woff + incl - mine;
  (void)tcount;
  uint32_t* slots = reinterpret_cast<uint32_t*>(base + plan.cand_base) + L.cand_off + (size_t)ci * L.cell_cap;
  for (int p = b; p < e; p++) {
    if (flag[p] & want) {
      const int y = p / cw, x = p - y * cw;
      const int s = smap[(y + 1) * ss + x + 1];
      // coordinates relative to (minBorderX, minBorderY): local + j*wCell (ORBextractor.cc:822-824)
      const uint32_t xr = (uint32_t)(x + 3 + ci_x * L.w_cell), yr = (uint32_t)(y + 3 + ci_y * L.h_cell);
      if (pos < L.cell_cap) slots[pos] = xr | (yr << 12) | ((uint32_t)s << 24);
      pos++;
    }
  }
  if (tid == 0) cellcnt[cell] = min(total, L.cell_cap);
}

// ------------------------------------------------------------------------------------------------
// Quadtree distribution: ORBextractor.cc:539-763 (DistributeOctTree) + :481-537 (DivideNode).
//
// The reference keeps a std::list of nodes, each owning a vector of keys.  Here a key only carries
// its node index; the list is an array in list order that is rebuilt after every pass:
//   new list = [children of the processed parents, last processed parent first, each parent's
//               children in order n4,n3,n2,n1 (they were push_front-ed n1..n4)] ++ [unprocessed
//               nodes in their old order].
// A "full" pass (the outer while) processes every node with more than one key in list order; a
// "careful" round (the inner while, entered when size + 3*nToExpand > N) processes the nodes created
// in the previous pass in descending (key count, address) order and stops as soon as size >= N.
// The heap-address tie-break of the reference is modelled by node creation order.
// Key order inside a node only matters for the final "first maximum wins" selection, which is
// reproduced by taking max (score, -original index).
// ------------------------------------------------------------------------------------------------
#define QT_T 512
#define QT_INV 0x80000000u

struct QtShared {
  uint32_t boxa[2][PS_QT_NCAP];   // x0 | y0 << 16
  uint32_t boxb[2][PS_QT_NCAP];   // x1 | y1 << 16
  uint32_t cnt[2][PS_QT_NCAP];    // key count | QT_INV (member of vSizeAndPointerToNode)
  uint32_t seq[2][PS_QT_NCAP];    // creation order
  uint32_t child[PS_QT_NCAP * 4];
  int32_t rank[PS_QT_NCAP];       // node -> rank in processing order, -1 = not a candidate
  int32_t ord[PS_QT_NCAP];        // rank -> node
  int32_t cpre[PS_QT_NCAP + 1];   // exclusive prefix (rank order) of non-empty child counts
  int32_t surv[PS_QT_NCAP];       // node -> new index when it survives unprocessed
  int32_t tmp[QT_T];
  int32_t total;
  int32_t cut;
  int32_t n_expand;
};

// exclusive in-place scan of a[0..n) (n <= capacity of a), returns the total to every thread.
__device__ int qt_exscan(int32_t* a, int n, QtShared& s) {
  const int t = threadIdx.x;
  const int chunk = (n + QT_T - 1) / QT_T;
  const int b = min(t * chunk, n), e = min(b + chunk, n);
  int sum = 0;
  for (int i = b; i < e; i++) sum += a[i];
  s.tmp[t] = sum;
  __syncthreads();
  if (t < 64) {
    int loc[QT_T / 64];
    int ss = 0;
#pragma unroll
    for (int k = 0; k < QT_T / 64; k++) { loc[k] = ss; ss += s.tmp[t * (QT_T / 64) + k]; }
    int incl = ss;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (t >= d) incl += o;
    }
    const int excl = incl - ss;
#pragma unroll
    for (int k = 0; k < QT_T / 64; k++) s.tmp[t * (QT_T / 64) + k] = excl + loc[k];
    if (t == 63) s.total = incl;
  }
  __syncthreads();
  int run = s.tmp[t];
  for (int i = b; i < e; i++) { const int v = a[i]; a[i] = run; run += v; }
  const int total = s.total;
  __syncthreads();
  return total;
}

__device__ __forceinline__ int qt_quadrant(uint32_t kxy, uint32_t ba, uint32_t bb) {
  const int x0 = ba & 0xFFFF, y0 = ba >> 16, x1 = bb & 0xFFFF, y1 = bb >> 16;
  const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);  // ceil(float(d)/2)
  const int kx = kxy & 0xFFFF, ky = kxy >> 16;
  return (kx < mx ? 0 : 1) + (ky < my ? 0 : 2);
}

__global__ __launch_bounds__(QT_T) void orb_quadtree(OrbPlan plan, uint8_t* arena) {
  __shared__ QtShared s;
  const int level = blockIdx.x, img = blockIdx.y, t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const OrbLevel L = plan.lv[level];
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  const int32_t* cellcnt = reinterpret_cast<const int32_t*>(base + plan.cellcnt_off) + L.cell_base;
  const uint32_t* slots = reinterpret_cast<const uint32_t*>(base + plan.cand_base) + L.cand_off;
  uint32_t* kxy = reinterpret_cast<uint32_t*>(base + plan.key_base) + L.key_off;
  uint32_t* kns = kxy + L.key_cap;   // node | score << 16
  uint32_t* sel = reinterpret_cast<uint32_t*>(base + plan.sel_base) + L.sel_off;
  int32_t* selcnt = reinterpret_cast<int32_t*>(base + plan.selcnt_off);
  int32_t* ncand_out = reinterpret_cast<int32_t*>(base + plan.ncand_off);
  const int N = L.quota;
  const int ncell = L.n_cols * L.n_rows;

  // ---- gather the cells' candidates in reference emission order (cell-major, raster inside) ----
  for (int c = t; c < ncell; c += QT_T) s.surv[c] = cellcnt[c];
  for (int i = t; i < L.n_ini * 4; i += QT_T) s.child[i] = 0;
  __syncthreads();
  const int n = qt_exscan(s.surv, ncell, s);
  if (t == 0) ncand_out[level] = n;
  if (n == 0) {
    if (t == 0) selcnt[level] = 0;
    return;
  }
  for (int c = wave; c < ncell; c += QT_T / 64) {
    const int cn = cellcnt[c], off = s.surv[c];
    for (int k = lane; k < cn; k += 64) {
      const uint32_t e = slots[(size_t)c * L.cell_cap + k];
      const uint32_t x = e & 0xFFF, y = (e >> 12) & 0xFFF, sc = e >> 24;
      // vpIniNodes[kp.pt.x / hX] (ORBextractor.cc:569): float division, truncation
      const int ni = (int)__fdiv_rn((float)x, L.h_x);
      kxy[off + k] = x | (y << 16);
      kns[off + k] = (uint32_t)ni | (sc << 16);
      atomicAdd(&s.child[ni], 1u);
    }
  }
  __syncthreads();
  // ---- initial nodes (ORBextractor.cc:543-586); empty ones stay in the array with count 0 and
  // are dropped at the first rebuild, which is when the reference has already erased them --------
  int cur = 0;
  for (int i = t; i < L.n_ini; i += QT_T) {
    const int x0 = (int)__fmul_rn(L.h_x, (float)i), x1 = (int)__fmul_rn(L.h_x, (float)(i + 1));
    s.boxa[0][i] = (uint32_t)x0;                               // y0 = 0
    s.boxb[0][i] = (uint32_t)x1 | ((uint32_t)(L.h - 2 * PS_MINB) << 16);
    s.cnt[0][i] = s.child[i];
    s.seq[0][i] = (uint32_t)i;
  }
  __syncthreads();
  int A = L.n_ini;   // array length
  int nn = 0;        // list size = nodes with keys
  for (int i = 0; i < L.n_ini; i++) nn += s.cnt[0][i] > 0 ? 1 : 0;
  uint32_t seq_base = (uint32_t)L.n_ini;
  bool careful = false;

  for (;;) {
    const int prev_size = nn;
    uint32_t* boxa = s.boxa[cur]; uint32_t* boxb = s.boxb[cur];
    uint32_t* cnt = s.cnt[cur]; uint32_t* sq = s.seq[cur];
    // A: clear child counters
    for (int i = t; i < A * 4; i += QT_T) s.child[i] = 0;
    if (t == 0) { s.cut = 0x7fffffff; s.n_expand = 0; }
    __syncthreads();
    // B: count keys per child of every candidate node
    for (int k = t; k < n; k += QT_T) {
      const int i = kns[k] & 0xFFFF;
      const uint32_t c = cnt[i];
      const bool cand = careful ? (c & QT_INV) != 0 : (c & ~QT_INV) > 1;
      if (cand) atomicAdd(&s.child[i * 4 + qt_quadrant(kxy[k], boxa[i], boxb[i])], 1u);
    }
    __syncthreads();
    // C: processing order
    int ncand;
    if (!careful) {
      for (int i = t; i < A; i += QT_T) s.rank[i] = (cnt[i] & ~QT_INV) > 1 ? 1 : 0;
      __syncthreads();
      // exclusive scan -> rank; remember candidacy in ord[] temporarily
      for (int i = t; i < A; i += QT_T) s.ord[i] = s.rank[i];
      __syncthreads();
      ncand = qt_exscan(s.rank, A, s);
      for (int i = t; i < A; i += QT_T) if (!s.ord[i]) s.rank[i] = -1;
      __syncthreads();
    } else {
      // descending (count, seq): rank = number of candidates that sort after this one
      int local = 0;
      for (int i = t; i < A; i += QT_T) {
        int r = -1;
        const uint32_t ci = cnt[i];
        if (ci & QT_INV) {
          r = 0;
          const uint32_t cc = ci & ~QT_INV, si = sq[i];
          for (int j = 0; j < A; j++) {
            const uint32_t cj = cnt[j];
            if (cj & QT_INV) {
              const uint32_t cjj = cj & ~QT_INV;
              r += (cjj > cc || (cjj == cc && sq[j] > si)) ? 1 : 0;
            }
          }
          local++;
        }
        s.rank[i] = r;
      }
      s.tmp[t] = local;
      __syncthreads();
      if (t == 0) { int acc = 0; for (int k = 0; k < QT_T; k++) acc += s.tmp[k]; s.total = acc; }
      __syncthreads();
      ncand = s.total;
      __syncthreads();
    }
    for (int i = t; i < A; i += QT_T) if (s.rank[i] >= 0) s.ord[s.rank[i]] = i;
    __syncthreads();
    // non-empty children per candidate, in rank order
    for (int r = t; r < ncand; r += QT_T) {
      const int i = s.ord[r];
      s.cpre[r] = (s.child[i * 4] > 0) + (s.child[i * 4 + 1] > 0) + (s.child[i * 4 + 2] > 0) +
                  (s.child[i * 4 + 3] > 0);
    }
    if (t == 0) s.cpre[ncand] = 0;
    __syncthreads();
    qt_exscan(s.cpre, ncand + 1, s);   // cpre[r] = sum_{r'<r}, cpre[ncand] = total
    // cutoff: first rank after which size >= N (careful rounds only, ORBextractor.cc:729-730)
    int m = ncand - 1;
    if (careful) {
      for (int r = t; r < ncand; r += QT_T)
        if (nn + s.cpre[r + 1] - (r + 1) >= N) atomicMin(&s.cut, r);
      __syncthreads();
      if (s.cut != 0x7fffffff) m = s.cut;
    }
    const int nproc = m + 1;
    const int TC = nproc > 0 ? s.cpre[nproc] : 0;
    // survivors: nodes with keys that are not processed
    for (int i = t; i < A; i += QT_T) {
      const bool processed = s.rank[i] >= 0 && s.rank[i] <= m;
      s.surv[i] = (!processed && (cnt[i] & ~QT_INV) > 0) ? 1 : 0;
    }
    __syncthreads();
    const int nsurv = qt_exscan(s.surv, A, s);
    const int nn_new = TC + nsurv;
    // E: write the new list into the other buffer
    uint32_t* nboxa = s.boxa[cur ^ 1]; uint32_t* nboxb = s.boxb[cur ^ 1];
    uint32_t* ncnt = s.cnt[cur ^ 1]; uint32_t* nsq = s.seq[cur ^ 1];
    for (int r = t; r < nproc; r += QT_T) {
      const int i = s.ord[r];
      const uint32_t ba = boxa[i], bb = boxb[i];
      const int x0 = ba & 0xFFFF, y0 = ba >> 16, x1 = bb & 0xFFFF, y1 = bb >> 16;
      const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
      const uint32_t c0 = s.child[i * 4], c1 = s.child[i * 4 + 1], c2 = s.child[i * 4 + 2], c3 = s.child[i * 4 + 3];
      const uint32_t cc[4] = {c0, c1, c2, c3};
      const uint32_t ca[4] = {(uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)mx | ((uint32_t)y0 << 16),
                              (uint32_t)x0 | ((uint32_t)my << 16), (uint32_t)mx | ((uint32_t)my << 16)};
      const uint32_t cb[4] = {(uint32_t)mx | ((uint32_t)my << 16), (uint32_t)x1 | ((uint32_t)my << 16),
                              (uint32_t)mx | ((uint32_t)y1 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16)};
      const int cr = s.cpre[r + 1] - s.cpre[r];
      const int pbase = TC - s.cpre[r] - cr;   // children of later-processed parents come first
      int before = 0;                           // non-empty children with smaller q
      int expand = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (cc[q] > 0) {
          const int after = cr - before - 1;    // non-empty children with larger q precede it
          const int pos = pbase + after;
          nboxa[pos] = ca[q];
          nboxb[pos] = cb[q];
          ncnt[pos] = cc[q] | (cc[q] > 1 ? QT_INV : 0u);
          nsq[pos] = seq_base + (uint32_t)(s.cpre[r] + before);
          expand += cc[q] > 1 ? 1 : 0;
          before++;
        }
      }
      if (expand) atomicAdd(&s.n_expand, expand);
    }
    for (int i = t; i < A; i += QT_T) {
      const bool processed = s.rank[i] >= 0 && s.rank[i] <= m;
      if (!processed && (cnt[i] & ~QT_INV) > 0) {
        const int pos = TC + s.surv[i];
        nboxa[pos] = boxa[i];
        nboxb[pos] = boxb[i];
        ncnt[pos] = cnt[i] & ~QT_INV;
        nsq[pos] = sq[i];
      }
    }
    // F: re-home the keys
    for (int k = t; k < n; k += QT_T) {
      const uint32_t kv = kns[k];
      const int i = kv & 0xFFFF;
      const int r = s.rank[i];
      int ni;
      if (r >= 0 && r <= m) {
        const int q = qt_quadrant(kxy[k], boxa[i], boxb[i]);
        const int cr = s.cpre[r + 1] - s.cpre[r];
        int before = 0;
        for (int qq = 0; qq < q; qq++) before += s.child[i * 4 + qq] > 0 ? 1 : 0;
        ni = TC - s.cpre[r] - cr + (cr - before - 1);
      } else {
        ni = TC + s.surv[i];
      }
      kns[k] = (kv & 0xFFFF0000u) | (uint32_t)ni;
    }
    __syncthreads();
    const int n_expand = s.n_expand;
    __syncthreads();
    seq_base += (uint32_t)TC;
    A = nn_new;
    nn = nn_new;
    cur ^= 1;
    // termination: ORBextractor.cc:669-737
    if (nn >= N || nn == prev_size) break;
    if (!careful && nn + 3 * n_expand > N) careful = true;
  }

  // ---- retain the best key of every node (ORBextractor.cc:742-760): max response, first wins ----
  uint32_t* best = s.child;
  for (int i = t; i < nn; i += QT_T) best[i] = 0;
  __syncthreads();
  for (int k = t; k < n; k += QT_T) {
    const uint32_t kv = kns[k];
    atomicMax(&best[kv & 0xFFFF], ((kv >> 16) << 20) | (0xFFFFFu - (uint32_t)k));
  }
  __syncthreads();
  for (int i = t; i < nn; i += QT_T) {
    const uint32_t bv = best[i];
    const uint32_t k = 0xFFFFFu - (bv & 0xFFFFFu);
    const uint32_t xy = kxy[k];
    const uint32_t x = (xy & 0xFFFF) + PS_MINB, y = (xy >> 16) + PS_MINB;   // ORBextractor.cc:843-844
    if (i < L.sel_cap) sel[i] = x | (y << 12) | ((bv >> 20) << 24);
  }
  if (t == 0) selcnt[level] = min(nn, L.sel_cap);
}

// ------------------------------------------------------------------------------------------------
// GaussianBlur(7x7, sigma 2, REFLECT_101) on CV_8U as OpenCV 3.4.3 computes it (fixed-point path):
// 8.8 kernel {18,34,49,55,49,34,18}, horizontal sums exact in 16 bits, vertical in 32 bits, one
// rounding (x + 2^15) >> 16, saturated.  ORBextractor.cc:1085-1086.  The padded plane's border is
// REFLECT_101 of the level, so reading the padded plane reproduces the border handling.
// One wave = a strip of 64 columns; each lane slides a 7-deep window of horizontal sums down ROWS.
// ------------------------------------------------------------------------------------------------
#define BLUR_ROWS 32
__global__ __launch_bounds__(256) void orb_blur(OrbPlan plan, uint8_t* arena, int level) {
  const OrbLevel L = plan.lv[level];
  const int img = blockIdx.z;
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y0 = (blockIdx.y * 4 + threadIdx.y) * BLUR_ROWS;
  if (x >= L.w || y0 >= L.h) return;
  const uint8_t* src = base + L.plane_off + (size_t)PS_EDGE * L.stride + PS_EDGE + x;
  uint8_t* dst = base + L.blur_off + x;
  const int rows = min(BLUR_ROWS, L.h - y0);
  uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0;
  for (int r = -3; r < rows + 3; r++) {
    const uint8_t* p = src + (ptrdiff_t)(y0 + r) * L.stride;
    const uint32_t hs = 18u * (p[-3] + p[3]) + 34u * (p[-2] + p[2]) + 49u * (p[-1] + p[1]) + 55u * p[0];
    h0 = h1; h1 = h2; h2 = h3; h3 = h4; h4 = h5; h5 = h6; h6 = hs;
    if (r >= 3) {
      const uint32_t v = 18u * (h0 + h6) + 34u * (h1 + h5) + 49u * (h2 + h4) + 55u * h3;
      const uint32_t o = (v + 32768u) >> 16;
      dst[(size_t)(y0 + r - 3) * L.bstride] = (uint8_t)(o > 255u ? 255u : o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor: ORBextractor.cc:77-104 (IC_Angle, cv::fastAtan2), :108-147
// (computeOrbDescriptor), :1095-1101 (scaling), one wave per selected keypoint.
// ------------------------------------------------------------------------------------------------
__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};
// circular patch of radius 15: row v has half-width umax[v] (ORBextractor.cc:451-469)
__constant__ int8_t c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  // OpenCV 3.4 atan_f32 (mathfuncs_core.simd.hpp), evaluated without FMA contraction
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

struct PsKeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };

__global__ __launch_bounds__(256) void orb_describe(OrbPlan plan, uint8_t* arena, PsKeyPoint* out_kps,
                                                   uint8_t* out_desc, int32_t* out_counts) {
  const int img = blockIdx.y;
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  uint8_t* base = arena + (size_t)img * plan.arena_bytes;
  const int32_t* selcnt = reinterpret_cast<const int32_t*>(base + plan.selcnt_off);
  if (slot >= plan.sel_total) return;
  int level = 0;
#pragma unroll
  for (int l = 1; l < PS_ORB_MAX_LEVELS; l++)
    if (l < plan.nlevels && slot >= plan.lv[l].sel_off) level = l;
  const OrbLevel L = plan.lv[level];
  const int k = slot - L.sel_off;
  int offset = 0, total = 0;
#pragma unroll
  for (int l = 0; l < PS_ORB_MAX_LEVELS; l++) {
    if (l < plan.nlevels) {
      const int c = selcnt[l];
      if (l < level) offset += c;
      total += c;
    }
  }
  if (slot == 0 && lane == 0) out_counts[img] = min(total, plan.kp_cap);
  if (k >= selcnt[level]) return;
  const int oi = offset + k;
  if (oi >= plan.kp_cap) return;
  const uint32_t e = (reinterpret_cast<const uint32_t*>(base + plan.sel_base) + L.sel_off)[k];
  const int kx = e & 0xFFF, ky = (e >> 12) & 0xFFF, sc = e >> 24;

  // ---- IC_Angle: m10 = sum u*I, m01 = sum v*I over the disc ----
  const uint8_t* center = base + L.plane_off + (size_t)(PS_EDGE + ky) * L.stride + PS_EDGE + kx;
  int m10 = 0, m01 = 0;
  // 31 rows x up to 31 columns: lanes 0..30 take column u = lane - 15 of two rows per step
  {
    const int u = (lane & 31) - 15;
    const int half = lane >> 5;
    if ((lane & 31) < 31) {
      for (int vv = -15 + half; vv <= 15; vv += 2) {
        const int av = vv < 0 ? -vv : vv;
        if (u >= -c_umax[av] && u <= c_umax[av]) {
          const int val = center[(ptrdiff_t)vv * L.stride + u];
          m10 += u * val;
          m01 += vv * val;
        }
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    m10 += __shfl_xor(m10, d);
    m01 += __shfl_xor(m01, d);
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);

  // ---- steered BRIEF on the blurred level: lane handles tests 4*lane .. 4*lane+3 ----
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float arad = __fmul_rn(angle, factorPI);
  const float a = (float)cos((double)arad), b = (float)sin((double)arad);
  const uint8_t* bc = base + L.blur_off + (size_t)ky * L.bstride + kx;
  uint32_t nib = 0;
#pragma unroll
  for (int tst = 0; tst < 4; tst++) {
    const int8_t* pp = &c_pattern[(lane * 4 + tst) * 4];
    const float x0 = (float)pp[0], y0 = (float)pp[1], x1 = (float)pp[2], y1 = (float)pp[3];
    const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
    const int q0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
    const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
    const int q1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
    const int t0 = bc[(ptrdiff_t)r0 * L.bstride + q0], t1 = bc[(ptrdiff_t)r1 * L.bstride + q1];
    nib |= (uint32_t)(t0 < t1) << tst;
  }
  // assemble 8 lanes (32 bits) into one dword on lanes 0,8,16,...
  uint32_t word = nib << (4 * (lane & 7));
  word |= __shfl_xor(word, 1);
  word |= __shfl_xor(word, 2);
  word |= __shfl_xor(word, 4);
  uint32_t* drow = reinterpret_cast<uint32_t*>(out_desc + ((size_t)img * plan.kp_cap + oi) * 32);
  if ((lane & 7) == 0) drow[lane >> 3] = word;
  if (lane == 0) {
    PsKeyPoint kp;
    kp.x = (float)kx;
    kp.y = (float)ky;
    if (level != 0) { kp.x = __fmul_rn(kp.x, L.scale); kp.y = __fmul_rn(kp.y, L.scale); }
    kp.size = L.kp_size;
    kp.angle = angle;
    kp.response = (float)(sc - 1);
    kp.octave = level;
    kp.class_id = -1;
    out_kps[(size_t)img * plan.kp_cap + oi] = kp;
  }
}

}  // namespace

// ---- launchers (called from orb_host.hip) --------------------------------------------------------
extern "C" void psk_orb_launch_pyramid(const OrbPlan* plan, int level, uint8_t* arena, const uint8_t* imgs,
                                       int img_stride, size_t img_pitch, const int4* tabs, int nimg,
                                       hipStream_t st) {
  const OrbLevel& L = plan->lv[level];
  const int PW = L.w + 2 * PS_EDGE, PH = L.h + 2 * PS_EDGE;
  dim3 blk(64, 4), grd((PW + 255) / 256, (PH + 3) / 4, nimg);
  if (level == 0)
    hipLaunchKernelGGL(orb_pyramid_level<true>, grd, blk, 0, st, *plan, level, arena, imgs, img_stride, img_pitch, tabs);
  else
    hipLaunchKernelGGL(orb_pyramid_level<false>, grd, blk, 0, st, *plan, level, arena, imgs, img_stride, img_pitch, tabs);
}
extern "C" void psk_orb_launch_fast(const OrbPlan* plan, uint8_t* arena, int nimg, hipStream_t st) {
  hipLaunchKernelGGL(orb_fast_cells, dim3(plan->n_cells, nimg), dim3(FAST_T), 0, st, *plan, arena);
}
extern "C" void psk_orb_launch_quadtree(const OrbPlan* plan, uint8_t* arena, int nimg, hipStream_t st) {
  hipLaunchKernelGGL(orb_quadtree, dim3(plan->nlevels, nimg), dim3(QT_T), 0, st, *plan, arena);
}
extern "C" void psk_orb_launch_blur(const OrbPlan* plan, uint8_t* arena, int nimg, hipStream_t st) {
  for (int l = 0; l < plan->nlevels; l++) {
    const OrbLevel& L = plan->lv[l];
    dim3 blk(64, 4), grd((L.w + 63) / 64, (L.h + BLUR_ROWS * 4 - 1) / (BLUR_ROWS * 4), nimg);
    hipLaunchKernelGGL(orb_blur, grd, blk, 0, st, *plan, arena, l);
  }
}
extern "C" void psk_orb_launch_describe(const OrbPlan* plan, uint8_t* arena, void* kps, uint8_t* desc,
                                        int32_t* counts, int nimg, hipStream_t st) {
  hipLaunchKernelGGL(orb_describe, dim3((plan->sel_total + 3) / 4, nimg), dim3(256), 0, st, *plan, arena,
                     (PsKeyPoint*)kps, desc, counts);
}
