// Host check of pointslot_amd/csrc/retain_best.h against the library it restates: for many lists (random, tie-heavy small-integer
// responses like FAST scores, sorted, reversed, organ-pipe, median-of-three killers that reach the heap-select fallback) the
// order after KeyPointsFilter::retainBest must be element for element what std::nth_element + std::partition leave.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../pointslot_amd/csrc/retain_best.h"

struct KP { float response; int id; };

static void std_retain_best(std::vector<KP>& kp, int n_points) {   // keypoint.cpp, as oracle/orb_oracle.cpp: cv_retain_best
  if (n_points >= 0 && kp.size() > (size_t)n_points) {
    if (n_points == 0) { kp.clear(); return; }
    std::nth_element(kp.begin(), kp.begin() + n_points - 1, kp.end(), [](const KP& a, const KP& b) { return a.response > b.response; });
    const float amb = kp[n_points - 1].response;
    auto e = std::partition(kp.begin() + n_points, kp.end(), [amb](const KP& k) { return k.response >= amb; });
    kp.resize(e - kp.begin());
  }
}

static int check(const std::vector<float>& resp, int n_points, const char* what) {
  const int n = (int)resp.size();
  std::vector<KP> a(n);
  std::vector<float> r(resp);
  std::vector<int32_t> v(n);
  for (int i = 0; i < n; i++) { a[i] = KP{resp[i], i}; v[i] = i; }
  std_retain_best(a, n_points);
  RbList L{r.data(), v.data()};
  const int m = rb_retain_best(L, n, n_points);
  if (m != (int)a.size()) { printf("FAIL %s n=%d k=%d: size %d vs %zu\n", what, n, n_points, m, a.size()); return 1; }
  for (int i = 0; i < m; i++)
    if (v[i] != a[i].id || r[i] != a[i].response) { printf("FAIL %s n=%d k=%d: element %d is %d, library %d\n", what, n, n_points, i, v[i], a[i].id); return 1; }
  return 0;
}

int main() {
  std::mt19937 g(12345);
  int bad = 0, cases = 0;
  for (int it = 0; it < 4000; it++) {
    const int n = 1 + g() % (it < 3000 ? 700 : 5000);
    std::vector<float> r(n);
    const int kind = it % 8;
    for (int i = 0; i < n; i++) {
      switch (kind) {
        case 0: r[i] = (float)(g() % 1000000) * 1e-3f; break;                         // distinct-ish
        case 1: r[i] = (float)(20 + g() % 60); break;                                  // FAST scores: small integers, many ties
        case 2: r[i] = (float)(g() % 4); break;                                        // almost all ties
        case 3: r[i] = (float)i; break;                                                // ascending
        case 4: r[i] = (float)(n - i); break;                                          // descending
        case 5: r[i] = (float)(i < n / 2 ? i : n - i); break;                          // organ pipe
        case 6: r[i] = 7.f; break;                                                     // constant
        default: r[i] = (float)((i * 7919) % 1013) * ((g() & 1) ? 1.f : 1e-6f); break;
      }
    }
    const int k = (int)(g() % (unsigned)(n + 3));
    bad += check(r, k, "random");
    cases++;
  }
  // median-of-three killer (Musser): drives introselect to its depth limit, i.e. into heap_select
  for (int n = 64; n <= 8192; n *= 2) {
    std::vector<float> r(n);
    const int k2 = n / 2;
    for (int i = 1; i <= k2; i++) {
      if (i % 2) { r[i - 1] = (float)i; r[i] = (float)(k2 + i); }
      r[k2 + i - 1] = (float)(2 * i);
    }
    for (int k : {1, n / 4, n / 2, n - 2}) { bad += check(r, k, "killer"); cases++; }
    std::vector<float> neg(r);
    for (float& x : neg) x = -x;
    for (int k : {1, n / 4, n / 2, n - 2}) { bad += check(neg, k, "killer-neg"); cases++; }
  }
  printf("%d cases, %d failures\n", cases, bad);
  return bad ? 1 : 0;
}
