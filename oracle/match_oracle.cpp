// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/orb_oracle.cpp header; the product never links this).
//
// CPU restatement of the reference's descriptor matching:
//   ORBmatcher::DescriptorDistance      /root/reference/src/ORBmatcher.cc:2704-2720
//   ORBmatcher::SearchByBruceMatching   :2043-2155
//   ORBmatcher::ComputeThreeMaxima      :2658-2699
//   ORBmatcher::SearchByProjection x3   :68-155, :157-248, :1613-1756 with
//   Frame::GetFeaturesInArea / PosInGrid /root/reference/src/Frame.cc:1808-1861,2027-2037
// PARITY UNPINNED by the reference (no tests there); pinned here by algebraic known answers
// (Hamming(a,a)=0, Hamming(a,~a)=256, popcount identities) in tests/test_oracle_match.py.
// Inputs are the plain arrays a caller extracts from Frame / MapPoint objects (SoA), outputs are index
// arrays standing in for the pointer vectors the reference fills.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;

int descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int32_t pa[8], pb[8];
  std::memcpy(pa, a, 32);
  std::memcpy(pb, b, 32);
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    unsigned int v = pa[i] ^ pb[i];
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

void three_maxima(std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

}  // namespace

extern "C" {

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

void orc_hamming_matrix(const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out) {
  for (int i = 0; i < nq; i++)
    for (int j = 0; j < nt; j++) out[(size_t)i * nt + j] = (uint16_t)descriptor_distance(q + 32 * i, t + 32 * j);
}

// SearchByBruceMatching.  q = last frame's object points (valid[i] = has a live MapObjectPoint that is not
// bad and not an outlier), t = current frame's object features.  query_of_train[j] = index of the query whose
// MapObjectPoint* the reference stores in vpMapObjectPointMatches[j], or -1.  Returns nmatches.
int orc_search_bruteforce(const uint8_t* qd, const float* qang, const uint8_t* qvalid, int nq, const uint8_t* td,
                          const float* tang, int nt, float nnratio, int check_ori, int* query_of_train) {
  for (int j = 0; j < nt; j++) query_of_train[j] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = HISTO_LENGTH / 360.0f;
  int nmatches = 0;
  for (int i = 0; i < nq; i++) {
    if (!qvalid[i]) continue;
    int bestDist1 = 256, bestIdx = -1, bestDist2 = 256;
    for (int j = 0; j < nt; j++) {
      if (query_of_train[j] >= 0) continue;
      const int dist = descriptor_distance(qd + 32 * i, td + 32 * j);
      if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = j; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist1 <= TH_LOW) {
      if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
        query_of_train[bestIdx] = i;
        if (check_ori) {
          float rot = qang[i] - tang[bestIdx];
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(bestIdx);
        }
        nmatches++;
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { query_of_train[idx] = -1; nmatches--; }
    }
  }
  return nmatches;
}


// ---- grid helpers (Frame.cc:1808-1861 GetFeaturesInArea; cells hold indices in insertion order) ----
struct OrcTrain {
  int n; const float* x; const float* y; const int* octave; const float* angle; const float* u_right;
  const uint8_t* desc; const uint8_t* occupied; const uint8_t* in_bbox; const int* cell_off; const int* cell_idx;
  float min_x, min_y, gw_inv, gh_inv;
};
static std::vector<int> features_in_area(const OrcTrain& F, float x, float y, float r, int minLevel, int maxLevel) {
  std::vector<int> v;
  const int COLS = 64, ROWS = 48;
  const int nMinCellX = std::max(0, (int)std::floor((x - F.min_x - r) * F.gw_inv));
  if (nMinCellX >= COLS) return v;
  const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - F.min_x + r) * F.gw_inv));
  if (nMaxCellX < 0) return v;
  const int nMinCellY = std::max(0, (int)std::floor((y - F.min_y - r) * F.gh_inv));
  if (nMinCellY >= ROWS) return v;
  const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - F.min_y + r) * F.gh_inv));
  if (nMaxCellY < 0) return v;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const int c = ix * ROWS + iy;
      for (int k = F.cell_off[c]; k < F.cell_off[c + 1]; k++) {
        const int j = F.cell_idx[k];
        if (bCheckLevels) {
          if (F.octave[j] < minLevel) continue;
          if (maxLevel >= 0 && F.octave[j] > maxLevel) continue;
        }
        const float distx = F.x[j] - x, disty = F.y[j] - y;
        if (std::fabs(distx) < r && std::fabs(disty) < r) v.push_back(j);
      }
    }
  return v;
}

// SearchByProjection(CurrentFrame, LastFrame, th, bMono), ORBmatcher.cc:1613-1756.
//   last-frame side: xw [m][3] (MapPoint::GetWorldPos), valid[i] = mvpMapPoints[i] && !mvbOutlier[i],
//   l_octave = mvKeys[i].octave, l_angle = mvKeysUn[i].angle, desc = pMP->GetDescriptor(), observed[i] = Observations()>0
//   tcw / tlw: float 4x4 row-major poses of the current / last frame; K = fx,fy,cx,cy,mbf,mb; bounds mnMinX..mnMaxY
// match_of_train[j] = index i of the last-frame point assigned to current keypoint j, -1 = unchanged/NULL.
int orc_search_projection_frame(const OrcTrain* F, int m, const float* xw, const uint8_t* valid, const int* l_octave,
                                const float* l_angle, const uint8_t* desc, const uint8_t* observed, const float* tcw,
                                const float* tlw, const float* K6, const float* bounds4, const float* scale_factors,
                                float th, int bMono, int check_ori, int* match_of_train) {
  for (int j = 0; j < F->n; j++) match_of_train[j] = -1;
  std::vector<uint8_t> blocked(F->occupied, F->occupied + F->n);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = HISTO_LENGTH / 360.0f;
  const float fx = K6[0], fy = K6[1], cx = K6[2], cy = K6[3], mbf = K6[4], mb = K6[5];
  // cv::Mat float algebra: products accumulate in double (cv::gemm), results are float
  auto mul3 = [](const float* R, const float* v, float* o) {   // R: rows of a 4x4
    for (int r = 0; r < 3; r++) o[r] = (float)((double)R[r * 4] * v[0] + (double)R[r * 4 + 1] * v[1] + (double)R[r * 4 + 2] * v[2]);
  };
  float twc[3], tlc[3];
  {  // twc = -Rcw.t()*tcw ; tlc = Rlw*twc + tlw
    float t[3] = {tcw[3], tcw[7], tcw[11]};
    for (int r = 0; r < 3; r++) twc[r] = (float)(-((double)tcw[r] * t[0] + (double)tcw[4 + r] * t[1] + (double)tcw[8 + r] * t[2]));
    float o[3];
    mul3(tlw, twc, o);
    tlc[0] = o[0] + tlw[3]; tlc[1] = o[1] + tlw[7]; tlc[2] = o[2] + tlw[11];
  }
  const bool bForward = tlc[2] > mb && !bMono;
  const bool bBackward = -tlc[2] > mb && !bMono;
  for (int i = 0; i < m; i++) {
    if (!valid[i]) continue;
    float x3Dc[3];
    mul3(tcw, xw + 3 * i, x3Dc);
    x3Dc[0] += tcw[3]; x3Dc[1] += tcw[7]; x3Dc[2] += tcw[11];
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = (float)(1.0 / x3Dc[2]);
    if (invzc < 0) continue;
    float u = fx * xc * invzc + cx, v = fy * yc * invzc + cy;
    if (u < bounds4[0] || u > bounds4[1]) continue;
    if (v < bounds4[2] || v > bounds4[3]) continue;
    const int nLastOctave = l_octave[i];
    const float radius = th * scale_factors[nLastOctave];
    std::vector<int> vIndices2;
    if (bForward) vIndices2 = features_in_area(*F, u, v, radius, nLastOctave, -1);
    else if (bBackward) vIndices2 = features_in_area(*F, u, v, radius, 0, nLastOctave);
    else vIndices2 = features_in_area(*F, u, v, radius, nLastOctave - 1, nLastOctave + 1);
    if (vIndices2.empty()) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (blocked[i2]) continue;
      if (F->u_right[i2] > 0) {
        const float ur = u - mbf * invzc;
        const float er = std::fabs(ur - F->u_right[i2]);
        if (er > radius) continue;
      }
      const int dist = descriptor_distance(desc + 32 * i, F->desc + 32 * i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      match_of_train[bestIdx2] = i;
      if (observed[i]) blocked[bestIdx2] = 1;
      nmatches++;
      if (check_ori) {
        float rot = l_angle[i] - F->angle[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) { match_of_train[idx] = -2; nmatches--; }   // -2: assigned in this call, then set to NULL (:1742-1750)
  }
  return nmatches;
}

// SearchByProjection(F, vpMapPoints, th) (ORBmatcher.cc:68-155) and SearchByProjection(F, nOrder, MOPs, th)
// (:157-248, object = 1).  Query side = the fields Frame::isInFrustum leaves in each point:
//   valid = mbTrackInView && !isBad(), proj_x/y/xr = mTrackProjX/Y/XR, level = mnTrackScaleLevel, view_cos
int orc_search_projection_points(const OrcTrain* F, int m, const uint8_t* valid, const float* proj_x, const float* proj_y,
                                 const float* proj_xr, const int* level, const float* view_cos, const uint8_t* desc,
                                 const uint8_t* observed, const float* scale_factors, float th, float nnratio,
                                 int object, int* match_of_train) {
  for (int j = 0; j < F->n; j++) match_of_train[j] = -1;
  std::vector<uint8_t> blocked(F->occupied, F->occupied + F->n);
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int i = 0; i < m; i++) {
    if (!valid[i]) continue;
    const int nPredictedLevel = level[i];
    float r = view_cos[i] > 0.998 ? 2.5f : 4.0f;   // RadiusByViewingCos
    if (bFactor) r *= th;
    const std::vector<int> vIndices = object
        ? features_in_area(*F, proj_x[i], proj_y[i], 5, nPredictedLevel - 1, nPredictedLevel + 1)
        : features_in_area(*F, proj_x[i], proj_y[i], r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
    if (vIndices.empty()) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : vIndices) {
      if (object && !F->in_bbox[idx]) continue;
      if (blocked[idx]) continue;
      if (F->u_right[idx] > 0) {
        const float er = std::fabs(proj_xr[i] - F->u_right[idx]);
        if (er > r * scale_factors[nPredictedLevel]) continue;
      }
      const int dist = descriptor_distance(desc + 32 * i, F->desc + 32 * idx);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F->octave[idx]; bestIdx = idx; }
      else if (dist < bestDist2) { bestLevel2 = F->octave[idx]; bestDist2 = dist; }
    }
    if (bestDist <= (object ? 130 : TH_HIGH)) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      match_of_train[bestIdx] = i;
      if (observed[i]) blocked[bestIdx] = 1;
      nmatches++;
    }
  }
  return nmatches;
}


// MapPoint / MapObjectPoint::ComputeDistinctiveDescriptors (/root/reference/src/MapObjectPoint.cc:379-436, src/MapPoint.cc:366):
// for every point, the index of the observation whose descriptor has the least median Hamming distance to the others.
// desc: concatenated 32-byte rows, point p owns rows [off[p], off[p+1]).  best[p] = -1 for a point without observations.
void orc_distinctive_descriptors(const uint8_t* desc, const int* off, int npoints, int* best) {
  for (int p = 0; p < npoints; p++) {
    const int N = off[p + 1] - off[p];
    if (N <= 0) { best[p] = -1; continue; }
    const uint8_t* D = desc + (size_t)off[p] * 32;
    std::vector<std::vector<int>> Distances(N, std::vector<int>(N, 0));
    for (int i = 0; i < N; i++)
      for (int j = i + 1; j < N; j++) {
        const int d = descriptor_distance(D + 32 * i, D + 32 * j);
        Distances[i][j] = d;
        Distances[j][i] = d;
      }
    int BestMedian = 2147483647, BestIdx = 0;
    for (int i = 0; i < N; i++) {
      std::vector<int> vDists(Distances[i]);
      std::sort(vDists.begin(), vDists.end());
      const int median = vDists[(size_t)(0.5 * (N - 1))];
      if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    best[p] = BestIdx;
  }
}


// The search half of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (/root/reference/src/ORBmatcher.cc:982-1136) and
// Fuse(ObjectKeyFrame*, vpMapObjectPoints, th) (:1138-1260): per candidate point the projection, the frustum / scale /
// viewing-angle gates, KeyFrame::GetFeaturesInArea and the chi-square-gated best Hamming match.  What the reference does
// with (bestIdx, bestDist) afterwards - Replace / AddObservation on the map - is pointer surgery and stays with the caller.
//   valid[i] = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF);  pos = GetWorldPos() / GetInObjFramePosition();
//   min_dist / max_dist = mfMinDistance / mfMaxDistance (the 0.8f / 1.2f of Get*DistanceInvariance are applied here);
//   R (row-major 3x3), t, ow = GetRotation(), GetTranslation(), GetCameraCenter();  bounds = {minX, maxX, minY, maxY} as doubles
//   (IsInImage: mnMinX.. ; IsInBBox: the detection box);  K5 = fx, fy, cx, cy, mbf.
// cv::Mat float products accumulate in double (cv::gemm, Mat::dot, cv::norm), everything else is float as written.
void orc_fuse_search(const OrcTrain* F, int m, const uint8_t* valid, const float* pos, const float* normal, const float* min_dist,
                     const float* max_dist, const uint8_t* desc, const float* R, const float* t, const float* ow, const float* K5,
                     const double* bounds, const float* scale_factors, const float* inv_level_sigma2, float log_scale_factor, int n_levels,
                     float th, int* best_idx, int* best_dist) {
  const float fx = K5[0], fy = K5[1], cx = K5[2], cy = K5[3], bf = K5[4];
  for (int i = 0; i < m; i++) {
    best_idx[i] = -1; best_dist[i] = 256;
    if (!valid[i]) continue;
    const float* P = pos + 3 * i;
    float Pc[3];
    for (int r = 0; r < 3; r++)
      Pc[r] = (float)((double)R[3 * r] * P[0] + (double)R[3 * r + 1] * P[1] + (double)R[3 * r + 2] * P[2]) + t[r];
    if (Pc[2] < 0.0f) continue;
    const float invz = 1 / Pc[2];
    const float x = Pc[0] * invz, y = Pc[1] * invz;
    const float u = fx * x + cx, v = fy * y + cy;
    const float ur = u - bf * invz;
    if (!(u >= bounds[0] && u < bounds[1] && v >= bounds[2] && v < bounds[3])) continue;
    const float PO[3] = {P[0] - ow[0], P[1] - ow[1], P[2] - ow[2]};
    const float dist3D = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const float* Pn = normal + 3 * i;
    const double dotn = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
    if (dotn < 0.5 * dist3D) continue;
    // PredictScale (MapPoint.cc:515-530): ceil(log(ratio) / mfLogScaleFactor), ::log(double)
    const float ratio = max_dist[i] / dist3D;
    int nPredictedLevel = (int)std::ceil(std::log((double)ratio) / log_scale_factor);
    if (nPredictedLevel < 0) nPredictedLevel = 0;
    else if (nPredictedLevel >= n_levels) nPredictedLevel = n_levels - 1;
    const float radius = th * scale_factors[nPredictedLevel];
    const std::vector<int> vIndices = features_in_area(*F, u, v, radius, -1, -1);
    if (vIndices.empty()) continue;
    int bestDist = 256, bestIdx = -1;
    for (int idx : vIndices) {
      const int kpLevel = F->octave[idx];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const float ex = u - F->x[idx], ey = v - F->y[idx];
      if (F->u_right[idx] >= 0) {
        const float er = ur - F->u_right[idx];
        const float e2 = ex * ex + ey * ey + er * er;
        if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
      } else {
        const float e2 = ex * ex + ey * ey;
        if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
      }
      const int dist = descriptor_distance(desc + 32 * i, F->desc + 32 * idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    best_dist[i] = bestDist;
    if (bestDist <= TH_LOW) best_idx[i] = bestIdx;
  }
}

}  // extern "C"
