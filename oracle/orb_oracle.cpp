// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library.  The product path (pointslot_amd/) never
// links, imports or calls anything in oracle/.
//
// CPU restatement (C++17, no dependencies) of the reference's ORB front-end:
//   ORB_SLAM2::ORBextractor  — /root/reference/src/ORBextractor.cc:410-470 (ctor tables),
//   :1107-1132 (ComputePyramid), :765-853 (ComputeKeyPointsOctTree), :481-763 (DivideNode /
//   DistributeOctTree), :77-104 (IC_Angle), :108-147 (computeOrbDescriptor), :1043-1105 (operator()).
//
// PARITY UNPINNED: the reference has no tests / golden vectors and its pixel
// arithmetic lives in OpenCV 3.4.x, which is not in /root/reference and cannot be built here.  The
// OpenCV stages (resize INTER_LINEAR 8U, copyMakeBorder REFLECT_101, FAST-9/16 + cornerScore + NMS,
// GaussianBlur 7x7 sigma 2 fixed-point, fastAtan2, cvRound) are restated below from OpenCV 3.4.3's
// published algorithms.  What IS pinned: the constants of SURVEY.md section 4-1 (feature quotas,
// pyramid sizes, umax, pattern hash) — see tests/test_oracle_orb.py.
//
// Deliberate, documented modelling choices where the reference is not reproducible by construction:
//  * DistributeOctTree sorts pair<int, ExtractorNode*> (ORBextractor.cc:680): ties on the key count
//    are broken by heap addresses.  The oracle breaks them by node creation order (later = larger).
//  * float expressions are evaluated without FMA contraction (compile with -ffp-contract=off).
//  * cos/sin of the keypoint angle are evaluated in double and rounded to float.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <vector>
#include <cfloat>
#include <cstddef>
#include <climits>
using std::ptrdiff_t;

namespace {

const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;

const int kPattern[1024] = {
#include "orb_pattern.inc"
};

inline int cvRound(double v) { return (int)std::nearbyint(v); }  // round-half-even (default FE mode)
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

struct KeyPoint {  // layout-compatible with cv::KeyPoint (28 bytes)
  float x, y, size, angle, response;
  int octave, class_id;
};

struct Plane {  // padded level: data is (w+38) x (h+38), stride = w+38
  int w = 0, h = 0, stride = 0;
  std::vector<uint8_t> buf;
  uint8_t* roi() { return buf.data() + EDGE_THRESHOLD * stride + EDGE_THRESHOLD; }
  const uint8_t* roi() const { return buf.data() + EDGE_THRESHOLD * stride + EDGE_THRESHOLD; }
};

struct Extractor {
  int nfeatures, nlevels, iniThFAST, minThFAST;
  double scaleFactor;  // the reference stores the float argument in a double member (ORBextractor.h:98)
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> mnFeaturesPerLevel, umax;
  std::vector<Plane> pyr;                         // mvImagePyramid (padded)
  std::vector<std::vector<uint8_t>> blur;         // blurred levels (w x h, tight)
  std::vector<std::vector<KeyPoint>> cand;        // vToDistributeKeys per level (coords rel. to minBorder)
  std::vector<std::vector<KeyPoint>> kps;         // selected per level (level coords, oriented)
  std::vector<KeyPoint> out_kps;
  std::vector<uint8_t> out_desc;
  // object-feature variant (the declared stand-in for cv::ORB + mask): candidates outside the mask are dropped
  // before DistributeOctTree.  mask: level-0 image, non-zero = keep; nullptr = the plain extractor
  const uint8_t* mask = nullptr;
  int mask_stride = 0, img_w = 0, img_h = 0;
};

// ---- ctor tables: ORBextractor.cc:410-470 -------------------------------------------------------
void init_tables(Extractor& E) {
  const int nl = E.nlevels;
  E.mvScaleFactor.assign(nl, 1.f);
  E.mvLevelSigma2.assign(nl, 1.f);
  for (int i = 1; i < nl; i++) {
    E.mvScaleFactor[i] = (float)(E.mvScaleFactor[i - 1] * E.scaleFactor);
    E.mvLevelSigma2[i] = E.mvScaleFactor[i] * E.mvScaleFactor[i];
  }
  E.mvInvScaleFactor.resize(nl);
  E.mvInvLevelSigma2.resize(nl);
  for (int i = 0; i < nl; i++) {
    E.mvInvScaleFactor[i] = 1.0f / E.mvScaleFactor[i];
    E.mvInvLevelSigma2[i] = 1.0f / E.mvLevelSigma2[i];
  }
  E.mnFeaturesPerLevel.resize(nl);
  float factor = (float)(1.0f / E.scaleFactor);
  float nDesired = E.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    E.mnFeaturesPerLevel[l] = cvRound(nDesired);
    sum += E.mnFeaturesPerLevel[l];
    nDesired *= factor;
  }
  E.mnFeaturesPerLevel[nl - 1] = std::max(E.nfeatures - sum, 0);

  E.umax.assign(HALF_PATCH_SIZE + 1, 0);
  int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
  int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) E.umax[v] = cvRound(std::sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (E.umax[v0] == E.umax[v0 + 1]) ++v0;
    E.umax[v] = v0;
    ++v0;
  }
}

// ---- OpenCV copyMakeBorder(BORDER_REFLECT_101): index -k -> k, n-1+k -> n-1-k -------------------
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

void fill_border(Plane& P) {
  const int W = P.w + 2 * EDGE_THRESHOLD, H = P.h + 2 * EDGE_THRESHOLD;
  uint8_t* base = P.buf.data();
  const uint8_t* roi = P.roi();
  for (int y = 0; y < H; y++) {
    int sy = reflect101(y - EDGE_THRESHOLD, P.h);
    for (int x = 0; x < W; x++) {
      int sx = reflect101(x - EDGE_THRESHOLD, P.w);
      if (sy == y - EDGE_THRESHOLD && sx == x - EDGE_THRESHOLD) continue;
      base[y * P.stride + x] = roi[sy * P.stride + sx];
    }
  }
}

// ---- OpenCV 3.4 resize(INTER_LINEAR), CV_8UC1, native path (imgproc/src/resize.cpp: the
// HResizeLinear<uchar,int,short,2048> / VResizeLinear<uchar,int,short,FixedPtCast<..,22>> pair) ----
inline short sat_short_round(float v) {
  int iv = cvRound(v);
  return (short)std::min(std::max(iv, -32768), 32767);
}

void resize_linear_8u(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                      int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    float c0 = 1.f - fx, c1 = fx;
    ialpha[dx * 2] = sat_short_round(c0 * 2048.f);
    ialpha[dx * 2 + 1] = sat_short_round(c1 * 2048.f);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloor(fy);
    fy -= sy;
    yofs[dy] = sy;
    float c0 = 1.f - fy, c1 = fy;
    ibeta[dy * 2] = sat_short_round(c0 * 2048.f);
    ibeta[dy * 2 + 1] = sat_short_round(c1 * 2048.f);
  }
  std::vector<int> row0(dw), row1(dw);
  auto hresize = [&](int sy, std::vector<int>& out) {
    const uint8_t* S = src + (size_t)sy * sstride;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      int s1 = (sx + 1 < sw) ? S[sx + 1] : S[sx];  // alpha1 == 0 there (fx forced to 0)
      out[dx] = S[sx] * ialpha[dx * 2] + s1 * ialpha[dx * 2 + 1];
    }
  };
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    hresize(sy0, row0);
    hresize(sy1, row1);
    const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int x = 0; x < dw; x++) {
      int v = (((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2;
      D[x] = (uint8_t)std::min(std::max(v, 0), 255);
    }
  }
}

// ---- ComputePyramid: ORBextractor.cc:1107-1132 -------------------------------------------------
void compute_pyramid(Extractor& E, const uint8_t* img, int w, int h, int stride) {
  E.pyr.assign(E.nlevels, Plane());
  for (int l = 0; l < E.nlevels; l++) {
    float scale = E.mvInvScaleFactor[l];
    Plane& P = E.pyr[l];
    P.w = cvRound((float)w * scale);
    P.h = cvRound((float)h * scale);
    P.stride = P.w + 2 * EDGE_THRESHOLD;
    P.buf.assign((size_t)P.stride * (P.h + 2 * EDGE_THRESHOLD), 0);
    if (l == 0) {
      for (int y = 0; y < h; y++) std::memcpy(P.roi() + (size_t)y * P.stride, img + (size_t)y * stride, w);
    } else {
      const Plane& S = E.pyr[l - 1];
      resize_linear_8u(S.roi(), S.w, S.h, S.stride, P.roi(), P.w, P.h, P.stride);
    }
    fill_border(P);
  }
}

// ---- OpenCV 3.4 FAST_t<16> with nonmax suppression (features2d/src/fast.cpp) and
// cornerScore<16> (fast_score.cpp), restated; run on a cell ROI -----------------------------------
int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int v = ptr[0];
  short d[N];
  for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

void fast9_16_nms(const uint8_t* img, int cols, int rows, int step, int threshold,
                  std::vector<KeyPoint>& keypoints) {
  static const int offsets16[][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1},
                                     {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                     {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};
  const int K = 8, N = 16 + K + 1;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = offsets16[k][0] + offsets16[k][1] * step;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  keypoints.clear();
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t threshold_tab[512];
  for (int i = -255; i <= 255; i++)
    threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  if (cols < 7 || rows < 7) return;  // loops below are empty for such ROIs
  std::vector<uint8_t> bufmem((size_t)cols * 3, 0);
  uint8_t* buf[3] = {bufmem.data(), bufmem.data() + cols, bufmem.data() + 2 * cols};
  std::vector<int> cpmem((size_t)(cols + 1) * 3, 0);
  int* cpbuf[3] = {cpmem.data() + 1, cpmem.data() + 1 + cols + 1, cpmem.data() + 1 + 2 * (cols + 1)};
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t* ptr = img + (size_t)i * step + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3];
    std::memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t* tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          int vt = v + threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      int j = cornerpos[k];
      int score = prev[j];
      if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
          score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
        KeyPoint kp{(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1};
        keypoints.push_back(kp);
      }
    }
  }
}

// ---- DivideNode / DistributeOctTree: ORBextractor.cc:481-763 -------------------------------------
struct Pt { int x, y; };
struct Node {
  std::vector<KeyPoint> vKeys;
  Pt UL, UR, BL, BR;
  std::list<Node>::iterator lit;
  bool bNoMore = false;
  long seq = 0;  // creation order: stands in for the heap address the reference sorts by
};

void divide_node(const Node& p, Node& n1, Node& n2, Node& n3, Node& n4) {
  const int halfX = (int)std::ceil(static_cast<float>(p.UR.x - p.UL.x) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(p.BR.y - p.UL.y) / 2);
  n1.UL = p.UL;
  n1.UR = Pt{p.UL.x + halfX, p.UL.y};
  n1.BL = Pt{p.UL.x, p.UL.y + halfY};
  n1.BR = Pt{p.UL.x + halfX, p.UL.y + halfY};
  n2.UL = n1.UR; n2.UR = p.UR; n2.BL = n1.BR; n2.BR = Pt{p.UR.x, p.UL.y + halfY};
  n3.UL = n1.BL; n3.UR = n1.BR; n3.BL = p.BL; n3.BR = Pt{n1.BR.x, p.BL.y};
  n4.UL = n3.UR; n4.UR = n2.BR; n4.BL = n3.BR; n4.BR = p.BR;
  for (const KeyPoint& kp : p.vKeys) {
    if (kp.x < n1.UR.x) {
      if (kp.y < n1.BR.y) n1.vKeys.push_back(kp);
      else n3.vKeys.push_back(kp);
    } else if (kp.y < n1.BR.y)
      n2.vKeys.push_back(kp);
    else
      n4.vKeys.push_back(kp);
  }
  if (n1.vKeys.size() == 1) n1.bNoMore = true;
  if (n2.vKeys.size() == 1) n2.bNoMore = true;
  if (n3.vKeys.size() == 1) n3.bNoMore = true;
  if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

std::vector<KeyPoint> distribute_octtree(const std::vector<KeyPoint>& keys, int minX, int maxX,
                                         int minY, int maxY, int N) {
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<Node> lNodes;
  std::vector<Node*> vpIni(nIni);
  long seq = 0;
  for (int i = 0; i < nIni; i++) {
    Node ni;
    ni.UL = Pt{(int)(hX * static_cast<float>(i)), 0};
    ni.UR = Pt{(int)(hX * static_cast<float>(i + 1)), 0};
    ni.BL = Pt{ni.UL.x, maxY - minY};
    ni.BR = Pt{ni.UR.x, maxY - minY};
    ni.seq = seq++;
    lNodes.push_back(ni);
    vpIni[i] = &lNodes.back();
  }
  for (const KeyPoint& kp : keys) vpIni[(size_t)(kp.x / hX)]->vKeys.push_back(kp);
  for (auto lit = lNodes.begin(); lit != lNodes.end();) {
    if (lit->vKeys.size() == 1) { lit->bNoMore = true; ++lit; }
    else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
    else ++lit;
  }
  typedef std::pair<int, Node*> SP;
  auto sp_less = [](const SP& a, const SP& b) {
    if (a.first != b.first) return a.first < b.first;
    return a.second->seq < b.second->seq;  // reference: raw pointer comparison
  };
  auto push_children = [&](Node* ch[4], std::vector<SP>& vec, int* nToExpand) {
    for (int c = 0; c < 4; c++) {
      if (ch[c]->vKeys.size() > 0) {
        ch[c]->seq = seq++;
        lNodes.push_front(*ch[c]);
        if (ch[c]->vKeys.size() > 1) {
          if (nToExpand) (*nToExpand)++;
          vec.push_back(std::make_pair((int)ch[c]->vKeys.size(), &lNodes.front()));
          lNodes.front().lit = lNodes.begin();
        }
      }
    }
  };
  bool bFinish = false;
  std::vector<SP> vSizeAndPointerToNode;
  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    auto lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->bNoMore) { ++lit; continue; }
      Node n1, n2, n3, n4;
      divide_node(*lit, n1, n2, n3, n4);
      Node* ch[4] = {&n1, &n2, &n3, &n4};
      push_children(ch, vSizeAndPointerToNode, &nToExpand);
      lit = lNodes.erase(lit);
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
      bFinish = true;
    } else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<SP> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end(), sp_less);
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          Node n1, n2, n3, n4;
          divide_node(*vPrev[j].second, n1, n2, n3, n4);
          Node* ch[4] = {&n1, &n2, &n3, &n4};
          push_children(ch, vSizeAndPointerToNode, nullptr);
          lNodes.erase(vPrev[j].second->lit);
          if ((int)lNodes.size() >= N) break;
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }
  std::vector<KeyPoint> res;
  for (auto& nd : lNodes) {
    const KeyPoint* best = &nd.vKeys[0];
    float maxResponse = best->response;
    for (size_t k = 1; k < nd.vKeys.size(); k++)
      if (nd.vKeys[k].response > maxResponse) { best = &nd.vKeys[k]; maxResponse = best->response; }
    res.push_back(*best);
  }
  return res;
}

// ---- OpenCV 3.4 fastAtan2 (core/src/mathfuncs_core.simd.hpp, scalar atan_f32) ------------------
float fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  float ax = std::abs(x), ay = std::abs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---- IC_Angle: ORBextractor.cc:77-104 ------------------------------------------------------------
float ic_angle(const Plane& P, float px, float py, const std::vector<int>& u_max) {
  int m_01 = 0, m_10 = 0;
  const int step = P.stride;
  const uint8_t* center = P.roi() + (size_t)cvRound(py) * step + cvRound(px);
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0, d = u_max[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2((float)m_01, (float)m_10);
}

// ---- ComputeKeyPointsOctTree: ORBextractor.cc:765-853 -------------------------------------------
void compute_keypoints(Extractor& E) {
  E.cand.assign(E.nlevels, {});
  E.kps.assign(E.nlevels, {});
  const float W = 30;
  for (int level = 0; level < E.nlevels; ++level) {
    const Plane& P = E.pyr[level];
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = P.w - EDGE_THRESHOLD + 3, maxBorderY = P.h - EDGE_THRESHOLD + 3;
    std::vector<KeyPoint>& vToDistributeKeys = E.cand[level];
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBorderY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBorderY - 3) continue;
      if (maxY > maxBorderY) maxY = (float)maxBorderY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBorderX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBorderX - 3) continue;
        if (maxX > maxBorderX) maxX = (float)maxBorderX;
        const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
        const uint8_t* roi = P.roi() + (size_t)y0 * P.stride + x0;
        std::vector<KeyPoint> vKeysCell;
        fast9_16_nms(roi, x1 - x0, y1 - y0, P.stride, E.iniThFAST, vKeysCell);
        if (vKeysCell.empty()) fast9_16_nms(roi, x1 - x0, y1 - y0, P.stride, E.minThFAST, vKeysCell);
        for (KeyPoint& kp : vKeysCell) {
          kp.x += j * wCell;
          kp.y += i * hCell;
          if (E.mask) {   // the candidate's pixel in the level-0 image: cvRound(level coordinate * scale), clipped
            const float sc = E.mvScaleFactor[level];
            const int mx = std::min(std::max(cvRound((kp.x + minBorderX) * sc), 0), E.img_w - 1);
            const int my = std::min(std::max(cvRound((kp.y + minBorderY) * sc), 0), E.img_h - 1);
            if (E.mask[(size_t)my * E.mask_stride + mx] == 0) continue;
          }
          vToDistributeKeys.push_back(kp);
        }
      }
    }
    std::vector<KeyPoint>& keypoints = E.kps[level];
    keypoints = distribute_octtree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                   E.mnFeaturesPerLevel[level]);
    const int scaledPatchSize = (int)(PATCH_SIZE * E.mvScaleFactor[level]);
    for (KeyPoint& kp : keypoints) {
      kp.x += minBorderX;
      kp.y += minBorderY;
      kp.octave = level;
      kp.size = (float)scaledPatchSize;
    }
  }
  for (int level = 0; level < E.nlevels; ++level)
    for (KeyPoint& kp : E.kps[level]) kp.angle = ic_angle(E.pyr[level], kp.x, kp.y, E.umax);
}

// ---- OpenCV 3.4.3 GaussianBlur(7x7, sigma 2) on CV_8U: fixed-point path (imgproc/src/smooth.cpp,
// fixedSmoothInvoker<uint8_t, ufixedpoint16>): kernel = round(256 * normalised gaussian) in 8.8,
// horizontal pass exact in u16, vertical pass u32, one rounding (x + 2^15) >> 16, saturate ----------
void gaussian_kernel_q8(int k[7]) {
  double v[7], sum = 0;
  for (int i = 0; i < 7; i++) { double x = i - 3; v[i] = std::exp(-0.5 * x * x / 4.0); sum += v[i]; }
  for (int i = 0; i < 7; i++) k[i] = cvRound(v[i] / sum * 256.0);
}

void gaussian_blur7(const Plane& P, std::vector<uint8_t>& out) {
  int k[7];
  gaussian_kernel_q8(k);
  const int w = P.w, h = P.h;
  out.assign((size_t)w * h, 0);
  std::vector<uint16_t> hbuf((size_t)w * (h + 6));
  const uint8_t* roi = P.roi();
  for (int y = -3; y < h + 3; y++) {
    const uint8_t* row = roi + (ptrdiff_t)reflect101(y, h) * P.stride;
    for (int x = 0; x < w; x++) {
      uint32_t s = 0;
      for (int i = 0; i < 7; i++) s += (uint32_t)k[i] * row[reflect101(x + i - 3, w)];
      hbuf[(size_t)(y + 3) * w + x] = (uint16_t)std::min<uint32_t>(s, 65535u);
    }
  }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t s = 0;
      for (int j = 0; j < 7; j++) s += (uint32_t)k[j] * hbuf[(size_t)(y + j) * w + x];
      uint32_t r = (s + 32768u) >> 16;
      out[(size_t)y * w + x] = (uint8_t)std::min<uint32_t>(r, 255u);
    }
}

// ---- computeOrbDescriptor: ORBextractor.cc:108-147 ----------------------------------------------
void orb_descriptor(const KeyPoint& kpt, const uint8_t* img, int step, uint8_t* desc) {
  const float factorPI = (float)(M_PI / 180.f);
  float angle = (float)kpt.angle * factorPI;
  float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
  const uint8_t* center = img + (ptrdiff_t)cvRound(kpt.y) * step + cvRound(kpt.x);
  const int* pat = kPattern;
  auto get = [&](int idx) -> int {
    float px = (float)pat[idx * 2], py = (float)pat[idx * 2 + 1];
    int yy = cvRound(px * b + py * a);
    int xx = cvRound(px * a - py * b);
    return center[yy * step + xx];
  };
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int t = 0; t < 8; t++) {
      int t0 = get(2 * t), t1 = get(2 * t + 1);
      val |= (t0 < t1) << t;
    }
    desc[i] = (uint8_t)val;
  }
}

void run(Extractor& E, const uint8_t* img, int w, int h, int stride) {
  E.out_kps.clear();
  E.out_desc.clear();
  compute_pyramid(E, img, w, h, stride);
  compute_keypoints(E);
  E.blur.assign(E.nlevels, {});
  for (int level = 0; level < E.nlevels; ++level) {
    std::vector<KeyPoint>& keypoints = E.kps[level];
    gaussian_blur7(E.pyr[level], E.blur[level]);  // (the reference skips empty levels; harmless)
    if (keypoints.empty()) continue;
    size_t off = E.out_desc.size();
    E.out_desc.resize(off + keypoints.size() * 32);
    for (size_t i = 0; i < keypoints.size(); i++)
      orb_descriptor(keypoints[i], E.blur[level].data(), E.pyr[level].w, &E.out_desc[off + i * 32]);
    float scale = E.mvScaleFactor[level];
    for (const KeyPoint& kp0 : keypoints) {
      KeyPoint kp = kp0;
      if (level != 0) { kp.x *= scale; kp.y *= scale; }
      E.out_kps.push_back(kp);
    }
  }
}

// =================================================================================================
// cv::ORB::create(nfeatures, scaleFactor, nlevels, edgeThreshold)->detectAndCompute(image, mask, keypoints, descriptors)
// as Frame::ExtractObjORB / OpencvORBDetector call it (/root/reference/src/Frame.cc:2623-2627): OpenCV's OWN
// ORB - firstLevel 0, WTA_K 2, HARRIS_SCORE, patchSize 31, fastThreshold 20 - restated from OpenCV 3.4.3's
// features2d/src/orb.cpp (ORB_Impl::detectAndCompute, computeKeyPoints, HarrisResponses, ICAngles, computeOrbDescriptors),
// imgproc/src/resize.cpp (INTER_LINEAR_EXACT: resize_bitExact with ufixedpoint16 coefficients) and
// features2d/src/keypoint.cpp (runByImageBorder, runByPixelsMask, retainBest).  UNVERIFIABLE HERE like every OpenCV stage of
// this file (OpenCV is not in the image).  Known implementation dependence: KeyPointsFilter::retainBest leaves its survivors in
// the order std::nth_element / std::partition produce - the restatement calls the same two algorithms of the same libstdc++.
// =================================================================================================
struct CvLevel { int w = 0, h = 0, stride = 0; float scale = 1.f; std::vector<uint8_t> pad, mask, blur; const uint8_t* roi() const { return pad.data() + (size_t)CV_BORDER * stride + CV_BORDER; } static const int CV_BORDER = 23; };

// resize(src, dst, dsize, 0, 0, INTER_LINEAR_EXACT) on CV_8UC1: coefficients in 8.8 fixed point from IEEE-double source
// coordinates, horizontal pass in 8.8 (u16), vertical pass in 16.16 (u32), one rounding (v + 2^15) >> 16
void linear_exact_coeffs(int ssize, int dsize, std::vector<int>& ofs, std::vector<int>& c0, std::vector<int>& c1, int& dmin, int& dmax) {
  const double inv_scale = (double)dsize / ssize, scale = 1.0 / inv_scale;
  ofs.assign(dsize, 0); c0.assign(dsize, 0); c1.assign(dsize, 0);
  dmin = 0; dmax = dsize;
  for (int val = 0; val < dsize; val++) {
    const double fval = scale * ((double)val + 0.5) - 0.5;
    const int ival = cvFloor(fval);
    if (ival >= 0 && ssize > 1) {
      if (ival < ssize - 1) {
        ofs[val] = ival;
        c1[val] = cvRound((fval - (double)ival) * 256.0);       // ufixedpoint16(softdouble)
        c0[val] = 256 - c1[val];                                 // fixedpoint::one() - coeffs[1]
      } else { ofs[val] = ssize - 1; dmax = std::min(dmax, val); }
    } else dmin = std::max(dmin, val + 1);
  }
}
void resize_linear_exact_8u(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  std::vector<int> xo, xa0, xa1, yo, yb0, yb1;
  int xmin, xmax, ymin, ymax;
  linear_exact_coeffs(sw, dw, xo, xa0, xa1, xmin, xmax);
  linear_exact_coeffs(sh, dh, yo, yb0, yb1, ymin, ymax);
  auto hline = [&](int sy, std::vector<uint32_t>& out) {           // hlineResize<uint8_t, ufixedpoint16, 2>
    const uint8_t* S = src + (size_t)sy * sstride;
    out.resize(dw);
    for (int x = 0; x < dw; x++) {
      if (x < xmin) out[x] = (uint32_t)S[0] << 8;
      else if (x >= xmax) out[x] = (uint32_t)S[sw - 1] << 8;
      else out[x] = (uint32_t)xa0[x] * S[xo[x]] + (uint32_t)xa1[x] * S[xo[x] + 1];
    }
  };
  std::vector<uint32_t> r0, r1;
  for (int y = 0; y < dh; y++) {
    uint8_t* D = dst + (size_t)y * dstride;
    if (y < ymin || y >= ymax) {                                    // rows outside the source: the first / last row alone
      hline(y < ymin ? 0 : sh - 1, r0);
      for (int x = 0; x < dw; x++) D[x] = (uint8_t)std::min<uint32_t>((r0[x] + 128u) >> 8, 255u);
      continue;
    }
    hline(yo[y], r0); hline(yo[y] + 1, r1);
    for (int x = 0; x < dw; x++) {                                  // vlineResize: ufixedpoint16 * ufixedpoint16 -> 16.16, saturating cast
      const uint64_t v = (uint64_t)r0[x] * (uint32_t)yb0[y] + (uint64_t)r1[x] * (uint32_t)yb1[y];
      D[x] = (uint8_t)std::min<uint64_t>((v + 32768u) >> 16, 255u);
    }
  }
}

struct CvOrb {
  int nfeatures = 1000, nlevels = 8, edgeThreshold = 19, fastThreshold = 20;
  double scaleFactor = 1.2;
  std::vector<CvLevel> levels;
  std::vector<KeyPoint> kps;
  std::vector<uint8_t> desc;
  std::vector<std::vector<KeyPoint>> fast_all;   // diagnostics: per level, FAST keypoints after the mask / border filters (raster order), Harris in .angle
};

void cv_retain_best(std::vector<KeyPoint>& keypoints, int n_points) {   // KeyPointsFilter::retainBest (keypoint.cpp)
  if (n_points >= 0 && keypoints.size() > (size_t)n_points) {
    if (n_points == 0) { keypoints.clear(); return; }
    std::nth_element(keypoints.begin(), keypoints.begin() + n_points - 1, keypoints.end(),
                     [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
    const float ambiguous_response = keypoints[n_points - 1].response;
    auto new_end = std::partition(keypoints.begin() + n_points, keypoints.end(), [ambiguous_response](const KeyPoint& k) { return k.response >= ambiguous_response; });
    keypoints.resize(new_end - keypoints.begin());
  }
}

void cv_orb_run(CvOrb& O, const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mask_stride) {
  const int B = CvLevel::CV_BORDER;   // max(edgeThreshold, descPatchSize = ceil(15 sqrt 2) = 22, HARRIS_BLOCK_SIZE / 2) + 1
  O.levels.assign(O.nlevels, CvLevel());
  O.kps.clear(); O.desc.clear();
  // ---- pyramid: level l is resized from level l - 1 (level 1 from the image), REFLECT_101 border; the mask likewise with
  // a zero border and, above level 0, threshold(254, TOZERO) ----
  for (int l = 0; l < O.nlevels; l++) {
    CvLevel& L = O.levels[l];
    L.scale = (float)std::pow(O.scaleFactor, (double)l);            // getScale(level, firstLevel = 0, scaleFactor)
    const float inv_scale = 1.0f / L.scale;
    L.w = cvRound(w * inv_scale); L.h = cvRound(h * inv_scale);
    L.stride = L.w + 2 * B;
    L.pad.assign((size_t)L.stride * (L.h + 2 * B), 0);
    uint8_t* roi = L.pad.data() + (size_t)B * L.stride + B;
    if (l == 0) for (int y = 0; y < h; y++) std::memcpy(roi + (size_t)y * L.stride, img + (size_t)y * stride, w);
    else resize_linear_exact_8u(O.levels[l - 1].roi(), O.levels[l - 1].w, O.levels[l - 1].h, O.levels[l - 1].stride, roi, L.w, L.h, L.stride);
    for (int y = -B; y < L.h + B; y++)
      for (int x = -B; x < L.w + B; x++)
        if (y < 0 || y >= L.h || x < 0 || x >= L.w) roi[(ptrdiff_t)y * L.stride + x] = roi[(ptrdiff_t)reflect101(y, L.h) * L.stride + reflect101(x, L.w)];
    if (mask) {
      L.mask.assign((size_t)L.w * L.h, 0);
      if (l == 0) for (int y = 0; y < h; y++) std::memcpy(&L.mask[(size_t)y * w], mask + (size_t)y * mask_stride, w);
      else {
        resize_linear_exact_8u(O.levels[l - 1].mask.data(), O.levels[l - 1].w, O.levels[l - 1].h, O.levels[l - 1].w, L.mask.data(), L.w, L.h, L.w);
        for (uint8_t& m : L.mask) m = m > 254 ? m : 0;
      }
    }
  }
  // ---- computeKeyPoints ----
  std::vector<int> nfeaturesPerLevel(O.nlevels);
  {
    const float factor = (float)(1.0 / O.scaleFactor);
    float ndesired = O.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)O.nlevels));
    int sum = 0;
    for (int l = 0; l < O.nlevels - 1; l++) { nfeaturesPerLevel[l] = cvRound(ndesired); sum += nfeaturesPerLevel[l]; ndesired *= factor; }
    nfeaturesPerLevel[O.nlevels - 1] = std::max(O.nfeatures - sum, 0);
  }
  std::vector<int> umax(HALF_PATCH_SIZE + 2);
  {
    int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1), vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = cvRound(std::sqrt((double)HALF_PATCH_SIZE * HALF_PATCH_SIZE - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) { while (umax[v0] == umax[v0 + 1]) ++v0; umax[v] = v0; ++v0; }
  }
  std::vector<KeyPoint> all;
  std::vector<int> counters(O.nlevels);
  O.fast_all.assign(O.nlevels, {});
  for (int l = 0; l < O.nlevels; l++) {
    const CvLevel& L = O.levels[l];
    std::vector<KeyPoint> kp;
    fast9_16_nms(L.roi(), L.w, L.h, L.stride, O.fastThreshold, kp);                // FastFeatureDetector::detect ...
    if (mask) {                                                                      // ... then KeyPointsFilter::runByPixelsMask
      std::vector<KeyPoint> in;
      for (const KeyPoint& k : kp) if (L.mask[(size_t)(int)(k.y + 0.5f) * L.w + (int)(k.x + 0.5f)] != 0) in.push_back(k);
      kp.swap(in);
    }
    {                                                                                // KeyPointsFilter::runByImageBorder(edgeThreshold)
      std::vector<KeyPoint> in;
      const int b = O.edgeThreshold;
      if (L.h > 2 * b && L.w > 2 * b)
        for (const KeyPoint& k : kp) if (k.x >= b && k.x < L.w - b && k.y >= b && k.y < L.h - b) in.push_back(k);
      kp.swap(in);
    }
    O.fast_all[l] = kp;
    cv_retain_best(kp, 2 * nfeaturesPerLevel[l]);                                   // HARRIS_SCORE: twice the quota by FAST score
    counters[l] = (int)kp.size();
    for (KeyPoint& k : kp) { k.octave = l; k.size = PATCH_SIZE * L.scale; }
    all.insert(all.end(), kp.begin(), kp.end());
  }
  if (all.empty()) return;
  // HarrisResponses(imagePyramid, layerInfo, allKeypoints, 7, 0.04f)
  auto harris = [&](const KeyPoint& k) {
    const CvLevel& L = O.levels[k.octave];
    const int blockSize = 7, r = blockSize / 2, step = L.stride;
    const float scale = 1.f / ((1 << 2) * blockSize * 255.f), scale_sq_sq = scale * scale * scale * scale;
    const uint8_t* ptr0 = L.roi() + (ptrdiff_t)(cvRound(k.y) - r) * step + cvRound(k.x) - r;
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < blockSize; i++)
      for (int j = 0; j < blockSize; j++) {
        const uint8_t* ptr = ptr0 + i * step + j;
        const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
        const int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
        a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
      }
    return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
  };
  for (int l = 0; l < O.nlevels; l++) for (KeyPoint& k : O.fast_all[l]) { KeyPoint t = k; t.octave = l; k.angle = harris(t); }
  for (KeyPoint& k : all) k.response = harris(k);
  std::vector<KeyPoint> best;
  int offset = 0;
  for (int l = 0; l < O.nlevels; l++) {
    std::vector<KeyPoint> kp(all.begin() + offset, all.begin() + offset + counters[l]);
    offset += counters[l];
    cv_retain_best(kp, nfeaturesPerLevel[l]);                                       // cull to the quota by the Harris score
    best.insert(best.end(), kp.begin(), kp.end());
  }
  // ICAngles, then pt *= scale
  for (KeyPoint& k : best) {
    const CvLevel& L = O.levels[k.octave];
    const uint8_t* center = L.roi() + (ptrdiff_t)cvRound(k.y) * L.stride + cvRound(k.x);
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
      int v_sum = 0, d = umax[v];
      for (int u = -d; u <= d; ++u) {
        const int val_plus = center[u + v * L.stride], val_minus = center[u - v * L.stride];
        v_sum += (val_plus - val_minus);
        m_10 += u * (val_plus + val_minus);
      }
      m_01 += v * v_sum;
    }
    k.angle = fast_atan2((float)m_01, (float)m_10);
  }
  for (KeyPoint& k : best) { const float sc = O.levels[k.octave].scale; k.x *= sc; k.y *= sc; }
  // ---- descriptors: GaussianBlur(7 x 7, sigma 2, REFLECT_101) of every level, then computeOrbDescriptors (WTA_K = 2) ----
  int kq[7];
  gaussian_kernel_q8(kq);
  // (OpenCV blurs every level in place inside the padded pyramid buffer: the border around it keeps the unblurred REFLECT_101
  // copies, which computeOrbDescriptors reads for pattern points that fall outside the level - blur is a padded plane too)
  for (CvLevel& L : O.levels) {
    L.blur = L.pad;
    uint8_t* broi = L.blur.data() + (size_t)B * L.stride + B;
    std::vector<uint16_t> hbuf((size_t)L.w * (L.h + 6));
    for (int y = -3; y < L.h + 3; y++) {
      const uint8_t* row = L.roi() + (ptrdiff_t)y * L.stride;     // rows / columns beyond the image: the REFLECT_101 border (23 >= 3)
      for (int x = 0; x < L.w; x++) {
        uint32_t sm = 0;
        for (int i = 0; i < 7; i++) sm += (uint32_t)kq[i] * row[x + i - 3];
        hbuf[(size_t)(y + 3) * L.w + x] = (uint16_t)std::min<uint32_t>(sm, 65535u);
      }
    }
    for (int y = 0; y < L.h; y++)
      for (int x = 0; x < L.w; x++) {
        uint32_t sm = 0;
        for (int j = 0; j < 7; j++) sm += (uint32_t)kq[j] * hbuf[(size_t)(y + j) * L.w + x];
        broi[(size_t)y * L.stride + x] = (uint8_t)std::min<uint32_t>((sm + 32768u) >> 16, 255u);
      }
  }
  O.kps = best;
  O.desc.assign(best.size() * 32, 0);
  for (size_t j = 0; j < best.size(); j++) {
    const KeyPoint& kpt = best[j];
    const CvLevel& L = O.levels[kpt.octave];
    const float scale = 1.f / L.scale;
    float angle = kpt.angle;
    angle *= (float)(M_PI / 180.f);
    const float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
    const uint8_t* center = L.blur.data() + (ptrdiff_t)(B + cvRound(kpt.y * scale)) * L.stride + B + cvRound(kpt.x * scale);
    const int* pat = kPattern;
    auto get = [&](int idx) -> int {
      const float x = pat[idx * 2] * a - pat[idx * 2 + 1] * b, y = pat[idx * 2] * b + pat[idx * 2 + 1] * a;
      return center[cvRound(y) * L.stride + cvRound(x)];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
      int val = 0;
      for (int t = 0; t < 8; t++) { const int t0 = get(2 * t), t1 = get(2 * t + 1); val |= (t0 < t1) << t; }
      O.desc[j * 32 + i] = (uint8_t)val;
    }
  }
}

}  // namespace

extern "C" {

// cv::ORB::create(nfeatures, scaleFactor, nlevels, edgeThreshold)->detectAndCompute(image, mask, ...): see cv_orb_run above
void* orc_cvorb_create(int nfeatures, float scaleFactor, int nlevels, int edgeThreshold, int fastThreshold) {
  CvOrb* O = new CvOrb();
  O->nfeatures = nfeatures; O->scaleFactor = (double)scaleFactor; O->nlevels = nlevels; O->edgeThreshold = edgeThreshold; O->fastThreshold = fastThreshold;
  return O;
}
void orc_cvorb_destroy(void* h) { delete (CvOrb*)h; }
int orc_cvorb_run(void* h, const uint8_t* img, int w, int hgt, int stride, const uint8_t* mask, int mask_stride) {
  CvOrb& O = *(CvOrb*)h;
  cv_orb_run(O, img, w, hgt, stride, mask, mask_stride);
  return (int)O.kps.size();
}
void orc_cvorb_result(void* h, void* kps28, uint8_t* desc) {
  CvOrb& O = *(CvOrb*)h;
  if (!O.kps.empty()) { std::memcpy(kps28, O.kps.data(), O.kps.size() * sizeof(KeyPoint)); std::memcpy(desc, O.desc.data(), O.desc.size()); }
}
void orc_cvorb_level_dims(void* h, int level, int* w, int* hgt) { CvOrb& O = *(CvOrb*)h; *w = O.levels[level].w; *hgt = O.levels[level].h; }
// level image without the border (tight w x h); what = 0 image, 1 blurred, 2 mask (zeros when no mask was given)
void orc_cvorb_level_plane(void* h, int level, int what, uint8_t* out) {
  const CvLevel& L = ((CvOrb*)h)->levels[level];
  for (int y = 0; y < L.h; y++)
    for (int x = 0; x < L.w; x++)
      out[(size_t)y * L.w + x] = what == 0 ? L.roi()[(ptrdiff_t)y * L.stride + x]
                               : (what == 1 ? L.blur[(size_t)(CvLevel::CV_BORDER + y) * L.stride + CvLevel::CV_BORDER + x] : (L.mask.empty() ? 0 : L.mask[(size_t)y * L.w + x]));
}
// FAST keypoints of a level after the mask and border filters, raster order: rows of (x, y, FAST score, Harris response) floats
int orc_cvorb_level_fast(void* h, int level, float* out, int cap) {
  const std::vector<KeyPoint>& v = ((CvOrb*)h)->fast_all[level];
  for (size_t i = 0; i < v.size() && (int)i < cap; i++) { out[4 * i] = v[i].x; out[4 * i + 1] = v[i].y; out[4 * i + 2] = v[i].response; out[4 * i + 3] = v[i].angle; }
  return (int)v.size();
}

void* orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
  Extractor* E = new Extractor();
  E->nfeatures = nfeatures;
  E->scaleFactor = scaleFactor;
  E->nlevels = nlevels;
  E->iniThFAST = iniThFAST;
  E->minThFAST = minThFAST;
  init_tables(*E);
  return E;
}
void orc_orb_destroy(void* h) { delete (Extractor*)h; }

// tables: which = 0 scale, 1 invScale, 2 sigma2, 3 invSigma2 (floats); quotas / umax as ints
void orc_orb_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int* quotas, int* umax16) {
  Extractor& E = *(Extractor*)h;
  for (int i = 0; i < E.nlevels; i++) {
    scale[i] = E.mvScaleFactor[i]; inv_scale[i] = E.mvInvScaleFactor[i];
    sigma2[i] = E.mvLevelSigma2[i]; inv_sigma2[i] = E.mvInvLevelSigma2[i];
    quotas[i] = E.mnFeaturesPerLevel[i];
  }
  for (int i = 0; i < 16; i++) umax16[i] = E.umax[i];
}

int orc_orb_run(void* h, const uint8_t* img, int w, int hgt, int stride) {
  Extractor& E = *(Extractor*)h;
  if (!img || w <= 0 || hgt <= 0) { E.out_kps.clear(); E.out_desc.clear(); return 0; }
  run(E, img, w, hgt, stride);
  return (int)E.out_kps.size();
}
// the object-feature stand-in: the same pipeline with the FAST candidates restricted to mask != 0 before the quadtree
int orc_orb_run_masked(void* h, const uint8_t* img, int w, int hgt, int stride, const uint8_t* mask, int mask_stride) {
  Extractor& E = *(Extractor*)h;
  if (!img || w <= 0 || hgt <= 0) { E.out_kps.clear(); E.out_desc.clear(); return 0; }
  E.mask = mask; E.mask_stride = mask_stride; E.img_w = w; E.img_h = hgt;
  run(E, img, w, hgt, stride);
  E.mask = nullptr;
  return (int)E.out_kps.size();
}
void orc_orb_result(void* h, void* kps28, uint8_t* desc) {
  Extractor& E = *(Extractor*)h;
  if (!E.out_kps.empty()) std::memcpy(kps28, E.out_kps.data(), E.out_kps.size() * sizeof(KeyPoint));
  if (!E.out_desc.empty()) std::memcpy(desc, E.out_desc.data(), E.out_desc.size());
}
void orc_orb_level_dims(void* h, int level, int* w, int* hgt) {
  Extractor& E = *(Extractor*)h;
  *w = E.pyr[level].w; *hgt = E.pyr[level].h;
}
// padded plane, tight (w+38) x (h+38)
void orc_orb_level_padded(void* h, int level, uint8_t* out) {
  Extractor& E = *(Extractor*)h;
  std::memcpy(out, E.pyr[level].buf.data(), E.pyr[level].buf.size());
}
void orc_orb_level_blur(void* h, int level, uint8_t* out) {
  Extractor& E = *(Extractor*)h;
  std::memcpy(out, E.blur[level].data(), E.blur[level].size());
}
int orc_orb_level_ncand(void* h, int level) { return (int)((Extractor*)h)->cand[level].size(); }
// candidates as int32 triples (x_rel, y_rel, score) in reference emission order
void orc_orb_level_cand(void* h, int level, int* out) {
  Extractor& E = *(Extractor*)h;
  for (size_t i = 0; i < E.cand[level].size(); i++) {
    out[i * 3] = (int)E.cand[level][i].x; out[i * 3 + 1] = (int)E.cand[level][i].y;
    out[i * 3 + 2] = (int)E.cand[level][i].response;
  }
}
int orc_orb_level_nkp(void* h, int level) { return (int)((Extractor*)h)->kps[level].size(); }
void orc_orb_level_kps(void* h, int level, void* kps28) {
  Extractor& E = *(Extractor*)h;
  if (!E.kps[level].empty())
    std::memcpy(kps28, E.kps[level].data(), E.kps[level].size() * sizeof(KeyPoint));
}
void orc_gaussian_kernel_q8(int* k7) { gaussian_kernel_q8(k7); }
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
int orc_pattern(int i) { return kPattern[i]; }

// Frame::ComputeStereoMatches (/root/reference/src/Frame.cc:2142-2316) on the keypoints / descriptors / pyramids the two
// extractors hold after orc_orb_run (left = hl, right = hr).  u_right / depth: N_left floats (-1 where unmatched).
// Restated literally, including the quirks: `int bestDist` receives the float SAD, the window test uses
// scaleduR0 + L - w, the parabola division is unguarded, and the cut is 1.5f * 1.4f * median of the SADs.
static int stereo_match_keys(Extractor& EL, Extractor& ER, const KeyPoint* KL, const uint8_t* DL, int N, const KeyPoint* KR,
                             const uint8_t* DR, int Nr, float mb, float mbf, float* u_right, float* depth) {
  for (int i = 0; i < N; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; }
  if (N == 0) return 0;
  const int TH_HIGH = 100, TH_LOW = 50;
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const int nRows = EL.pyr[0].h;
  std::vector<std::vector<size_t>> vRowIndices(nRows);
  for (int iR = 0; iR < Nr; iR++) {
    const KeyPoint& kp = KR[iR];
    const float kpY = kp.y;
    const float r = 2.0f * ER.mvScaleFactor[kp.octave];
    const int maxr = (int)std::ceil(kpY + r), minr = (int)std::floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);   // (the reference does not guard; out-of-range rows are UB there)
  }
  const float minZ = mb, minD = 0, maxD = mbf / minZ;
  std::vector<std::pair<int, int>> vDistIdx;
  auto desc_dist = [](const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
  };
  for (int iL = 0; iL < N; iL++) {
    const KeyPoint& kpL = KL[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    for (size_t iC = 0; iC < vCandidates.size(); iC++) {
      const size_t iR = vCandidates[iC];
      const KeyPoint& kpR = KR[iR];
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        const int dist = desc_dist(DL + (size_t)iL * 32, DR + iR * 32);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = KR[bestIdxR].x;
      const float scaleFactor = EL.mvInvScaleFactor[kpL.octave];
      const float scaleduL = std::round(kpL.x * scaleFactor);
      const float scaledvL = std::round(kpL.y * scaleFactor);
      const float scaleduR0 = std::round(uR0 * scaleFactor);
      const int w = 5, L = 5;
      const Plane& PL = EL.pyr[kpL.octave];
      const Plane& PR = ER.pyr[kpL.octave];
      auto pix = [](const Plane& P, int y, int x) { return (float)P.roi()[(ptrdiff_t)y * P.stride + x]; };
      const int cy = (int)scaledvL, cxl = (int)scaleduL;
      const float ILc = pix(PL, cy, cxl);
      int bestDistS = INT32_MAX, bestincR = 0;
      float vDists[2 * 5 + 1];
      const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= PR.w) continue;
      for (int incR = -L; incR <= +L; incR++) {
        const int cxr = (int)(scaleduR0 + incR);
        const float IRc = pix(PR, cy, cxr);
        float dist = 0;
        for (int dy = -w; dy <= w; dy++)
          for (int dx = -w; dx <= w; dx++)
            dist += std::fabs((pix(PL, cy + dy, cxl + dx) - ILc) - (pix(PR, cy + dy, cxr + dx) - IRc));
        if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = EL.mvScaleFactor[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = 0.01f; bestuR = uL - 0.01f; }
        depth[iL] = mbf / disparity;
        u_right[iL] = bestuR;
        vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
      }
    }
  }
  if (vDistIdx.empty()) return 0;   // (the reference indexes an empty vector here)
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  int kept = (int)vDistIdx.size();
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if (vDistIdx[i].first < thDist) break;
    u_right[vDistIdx[i].second] = -1;
    depth[vDistIdx[i].second] = -1;
    kept--;
  }
  return kept;
}

int orc_stereo_match(void* hl, void* hr, float mb, float mbf, float* u_right, float* depth) {
  Extractor& EL = *(Extractor*)hl;
  Extractor& ER = *(Extractor*)hr;
  return stereo_match_keys(EL, ER, EL.out_kps.data(), EL.out_desc.data(), (int)EL.out_kps.size(), ER.out_kps.data(), ER.out_desc.data(),
                           (int)ER.out_kps.size(), mb, mbf, u_right, depth);
}

// Frame::ComputeObjStereoMatches (/root/reference/src/Frame.cc:2318-2503): the same algorithm on the object key sets
// (mvTempObjKeys / mvTempObjKeysRight with their descriptors) against the SAME two image pyramids.
int orc_stereo_match_keys(void* hl, void* hr, const void* kps_l, const uint8_t* desc_l, int nl, const void* kps_r, const uint8_t* desc_r,
                          int nr, float mb, float mbf, float* u_right, float* depth) {
  if (nl <= 0) return 0;                                   // Frame.cc:2320
  return stereo_match_keys(*(Extractor*)hl, *(Extractor*)hr, (const KeyPoint*)kps_l, desc_l, nl, (const KeyPoint*)kps_r, desc_r, nr, mb, mbf,
                           u_right, depth);
}

// standalone quadtree for unit tests: keys as int triples (x,y,response)
int orc_distribute(const int* keys, int n, int minX, int maxX, int minY, int maxY, int N, int* out_xyz) {
  std::vector<KeyPoint> v(n);
  for (int i = 0; i < n; i++)
    v[i] = KeyPoint{(float)keys[i * 3], (float)keys[i * 3 + 1], 7.f, -1.f, (float)keys[i * 3 + 2], 0, -1};
  std::vector<KeyPoint> r = distribute_octtree(v, minX, maxX, minY, maxY, N);
  for (size_t i = 0; i < r.size(); i++) {
    out_xyz[i * 3] = (int)r[i].x; out_xyz[i * 3 + 1] = (int)r[i].y; out_xyz[i * 3 + 2] = (int)r[i].response;
  }
  return (int)r.size();
}

}  // extern "C"
