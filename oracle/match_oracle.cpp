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

}  // extern "C"
