// cv::KeyPointsFilter::retainBest (OpenCV 3.4 features2d/src/keypoint.cpp) for the device-resident object detector.
// The survivors of retainBest come out in whatever order std::nth_element / std::partition leave them, and the reference's
// object keypoints keep that order (cv::ORB -> Frame::ExtractObjORB, /root/reference/src/Frame.cc:2623-2665): the only way
// to have it on the device is to execute libstdc++'s own algorithms there.  This header restates them for a (response, payload)
// pair of arrays - std::__introselect with its median-of-three pivot, unguarded partition, final insertion sort and the
// heap-select fallback (bits/stl_algo.h, bits/stl_heap.h of GCC 11), and the bidirectional std::__partition - one statement
// per statement, so that the element order after every step is the library's.  Comparator: KeypointResponseGreater
// (a.response > b.response).  Sequential by nature: one lane runs it per (image, level) list, and only when a level holds
// more keypoints than its quota.  tests/cpp/retain_best_check.cpp compares it with the library on the host, element by element.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define RB_HD __host__ __device__ __forceinline__
#else
#define RB_HD inline
#endif

struct RbList { float* r; int32_t* v; };   // response and payload of element i: r[i], v[i]

RB_HD bool rb_gt(const RbList& L, int a, int b) { return L.r[a] > L.r[b]; }
RB_HD void rb_swap(const RbList& L, int a, int b) {
  const float tr = L.r[a]; L.r[a] = L.r[b]; L.r[b] = tr;
  const int32_t tv = L.v[a]; L.v[a] = L.v[b]; L.v[b] = tv;
}
RB_HD void rb_move(const RbList& L, int dst, int src) { L.r[dst] = L.r[src]; L.v[dst] = L.v[src]; }

// std::__move_median_to_first(result, a, b, c, comp)
RB_HD void rb_move_median_to_first(const RbList& L, int result, int a, int b, int c) {
  if (rb_gt(L, a, b)) {
    if (rb_gt(L, b, c)) rb_swap(L, result, b);
    else if (rb_gt(L, a, c)) rb_swap(L, result, c);
    else rb_swap(L, result, a);
  } else if (rb_gt(L, a, c)) rb_swap(L, result, a);
  else if (rb_gt(L, b, c)) rb_swap(L, result, c);
  else rb_swap(L, result, b);
}
// std::__unguarded_partition(first, last, pivot, comp)
RB_HD int rb_unguarded_partition(const RbList& L, int first, int last, int pivot) {
  while (true) {
    while (rb_gt(L, first, pivot)) ++first;
    --last;
    while (rb_gt(L, pivot, last)) --last;
    if (!(first < last)) return first;
    rb_swap(L, first, last);
    ++first;
  }
}
// std::__insertion_sort(first, last, comp) with std::__unguarded_linear_insert
RB_HD void rb_insertion_sort(const RbList& L, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    const float vr = L.r[i]; const int32_t vv = L.v[i];
    if (vr > L.r[first]) {
      for (int k = i; k > first; --k) rb_move(L, k, k - 1);     // std::move_backward(first, i, i + 1)
      L.r[first] = vr; L.v[first] = vv;
    } else {
      int lastp = i, next = i - 1;
      while (vr > L.r[next]) { rb_move(L, lastp, next); lastp = next; --next; }
      L.r[lastp] = vr; L.v[lastp] = vv;
    }
  }
}
// std::__push_heap / __adjust_heap / __make_heap / __pop_heap / __heap_select on [first, ...)
RB_HD void rb_push_heap(const RbList& L, int first, int hole, int top, float vr, int32_t vv) {
  int parent = (hole - 1) / 2;
  while (hole > top && L.r[first + parent] > vr) { rb_move(L, first + hole, first + parent); hole = parent; parent = (hole - 1) / 2; }
  L.r[first + hole] = vr; L.v[first + hole] = vv;
}
RB_HD void rb_adjust_heap(const RbList& L, int first, int hole, int len, float vr, int32_t vv) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (rb_gt(L, first + second, first + (second - 1))) second--;
    rb_move(L, first + hole, first + second);
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    rb_move(L, first + hole, first + (second - 1));
    hole = second - 1;
  }
  rb_push_heap(L, first, hole, top, vr, vv);
}
RB_HD void rb_heap_select(const RbList& L, int first, int middle, int last) {
  const int len = middle - first;
  if (len >= 2) {                                               // std::__make_heap
    int parent = (len - 2) / 2;
    while (true) {
      const float vr = L.r[first + parent]; const int32_t vv = L.v[first + parent];
      rb_adjust_heap(L, first, parent, len, vr, vv);
      if (parent == 0) break;
      parent--;
    }
  }
  for (int i = middle; i < last; ++i)
    if (rb_gt(L, i, first)) {                                   // std::__pop_heap(first, middle, i, comp)
      const float vr = L.r[i]; const int32_t vv = L.v[i];
      rb_move(L, i, first);
      rb_adjust_heap(L, first, 0, len, vr, vv);
    }
}
// std::nth_element(first, nth, last, comp) = std::__introselect(first, nth, last, std::__lg(last - first) * 2, comp)
RB_HD void rb_nth_element(const RbList& L, int first, int nth, int last) {
  if (first == last || nth == last) return;
  int depth_limit = 0;
  for (int n = last - first; n > 1; n >>= 1) depth_limit++;    // std::__lg
  depth_limit *= 2;
  while (last - first > 3) {
    if (depth_limit == 0) {
      rb_heap_select(L, first, nth + 1, last);
      rb_swap(L, first, nth);
      return;
    }
    --depth_limit;
    const int mid = first + (last - first) / 2;                 // std::__unguarded_partition_pivot
    rb_move_median_to_first(L, first, first + 1, mid, last - 1);
    const int cut = rb_unguarded_partition(L, first + 1, last, first);
    if (cut <= nth) first = cut; else last = cut;
  }
  rb_insertion_sort(L, first, last);
}
// std::partition (bidirectional iterators) with the predicate response >= threshold; returns the new end
RB_HD int rb_partition_ge(const RbList& L, int first, int last, float threshold) {
  while (true) {
    while (true) {
      if (first == last) return first;
      else if (L.r[first] >= threshold) ++first;
      else break;
    }
    --last;
    while (true) {
      if (first == last) return first;
      else if (!(L.r[last] >= threshold)) --last;
      else break;
    }
    rb_swap(L, first, last);
    ++first;
  }
}
// KeyPointsFilter::retainBest(keypoints, n_points) on the list [0, n); returns the new size
RB_HD int rb_retain_best(const RbList& L, int n, int n_points) {
  if (n_points >= 0 && n > n_points) {
    if (n_points == 0) return 0;
    rb_nth_element(L, 0, n_points - 1, n);
    const float ambiguous_response = L.r[n_points - 1];
    return rb_partition_ge(L, n_points, n, ambiguous_response);
  }
  return n;
}

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------------------------------------------
// The same algorithms with the two partition loops executed by a whole wave (64 lanes, all of them call these functions together, the
// lists live in LDS).  The element order after every step is still the library's: a Hoare-style partition is a sequence of swaps
// whose partners are fixed by the ORIGINAL contents of the range - the k-th element (from the left) at which the left scan stops is
// swapped with the k-th element (from the right) at which the right scan stops, as long as the former lies left of the latter; the
// scans never look at a position again once a pointer has passed it.  So: every lane classifies its share of the positions, two prefix
// sums number the stops from the left and from the right, the pairs with "left stop < right stop" are swapped all at once.
// (One lane walking a 500-element list through LDS was 125 us; r04.)  lpos / rpos: LDS scratch for one int per element of the range.
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int rbw_scan_inclusive(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v += o; }
  return v;
}
__device__ __forceinline__ void rbw_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// swaps of the pairs (lpos[k], rpos[nR - 1 - k]) for k < npairs; rpos holds the right stops in ascending position order
__device__ __forceinline__ void rbw_swap_pairs(const RbList& L, const int* lpos, const int* rpos, int nR, int npairs, int lane) {
  for (int k = lane; k < npairs; k += 64) rb_swap(L, lpos[k], rpos[nR - 1 - k]);
  rbw_fence();
}
// std::__unguarded_partition(first, last, pivot) with pivot = first - 1 (std::__unguarded_partition_pivot calls it so): returns the cut
__device__ inline int rbw_unguarded_partition(const RbList& L, int first, int last, int* lpos, int* rpos, int lane) {
  const int lo = first - 1, len = last - lo, E = (len + 63) >> 6;        // the pivot's own position is the right scan's last stop
  const float rp = L.r[lo];
  const int p0 = lo + lane * E;
  int cl = 0, cr = 0;
  for (int e = 0; e < E; e++) {
    const int p = p0 + e;
    if (p < last) { const float v = L.r[p]; cl += (p >= first && !(v > rp)) ? 1 : 0; cr += !(rp > v) ? 1 : 0; }
  }
  const int il = rbw_scan_inclusive(cl, lane), ir = rbw_scan_inclusive(cr, lane);
  const int nL = __shfl(il, 63), nR = __shfl(ir, 63);
  int kl = il - cl, kr = ir - cr;
  for (int e = 0; e < E; e++) {
    const int p = p0 + e;
    if (p < last) {
      const float v = L.r[p];
      if (p >= first && !(v > rp)) lpos[kl++] = p;
      if (!(rp > v)) rpos[kr++] = p;
    }
  }
  rbw_fence();
  // the number of swaps: the pairs k with (k-th left stop) < (k-th right stop from the right); monotone in k
  int cnt = 0;
  const int nk = nL < nR ? nL : nR;
  for (int k = lane; k < nk; k += 64) cnt += lpos[k] < rpos[nR - 1 - k] ? 1 : 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
  // Where the left scan ends after the last swap: at its next stop in the original contents - or, if that lies further right, at the
  // position of the last swap's right partner, which by then holds a value the left scan stops at (the only place where a scan meets
  // a position that was already swapped)
  int cut = cnt < nL ? lpos[cnt] : 0x7fffffff;
  if (cnt >= 1) { const int rprev = rpos[nR - cnt]; cut = cut < rprev ? cut : rprev; }
  rbw_fence();
  rbw_swap_pairs(L, lpos, rpos, nR, cnt, lane);
  return cut;
}
// std::partition (bidirectional) with the predicate response >= threshold on [first, last): returns the new end
__device__ inline int rbw_partition_ge(const RbList& L, int first, int last, float threshold, int* lpos, int* rpos, int lane) {
  const int len = last - first, E = (len + 63) >> 6;
  const int p0 = first + lane * E;
  int cl = 0, cr = 0;
  for (int e = 0; e < E; e++) {
    const int p = p0 + e;
    if (p < last) { const bool keep = L.r[p] >= threshold; cl += keep ? 0 : 1; cr += keep ? 1 : 0; }
  }
  const int il = rbw_scan_inclusive(cl, lane), ir = rbw_scan_inclusive(cr, lane);
  const int nL = __shfl(il, 63), nR = __shfl(ir, 63);
  int kl = il - cl, kr = ir - cr;
  for (int e = 0; e < E; e++) {
    const int p = p0 + e;
    if (p < last) { if (L.r[p] >= threshold) rpos[kr++] = p; else lpos[kl++] = p; }
  }
  rbw_fence();
  int cnt = 0;
  const int nk = nL < nR ? nL : nR;
  for (int k = lane; k < nk; k += 64) cnt += lpos[k] < rpos[nR - 1 - k] ? 1 : 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
  rbw_fence();
  rbw_swap_pairs(L, lpos, rpos, nR, cnt, lane);
  return first + nR;                   // the elements that satisfy the predicate end up in front, whatever the order of the swaps
}
// KeyPointsFilter::retainBest by a wave: the introselect loop's pivot choice, the final insertion sort of at most three elements and
// the heap-select fallback stay with lane 0
__device__ inline int rbw_retain_best(const RbList& L, int n, int n_points, int* lpos, int* rpos, int lane) {
  if (!(n_points >= 0 && n > n_points)) return n;
  if (n_points == 0) return 0;
  {
    int first = 0, last = n;
    const int nth = n_points - 1;
    int depth_limit = 0;
    for (int m = last - first; m > 1; m >>= 1) depth_limit++;
    depth_limit *= 2;
    bool done = false;
    while (last - first > 3) {
      if (depth_limit == 0) {
        if (lane == 0) { rb_heap_select(L, first, nth + 1, last); rb_swap(L, first, nth); }
        rbw_fence();
        done = true;
        break;
      }
      --depth_limit;
      const int mid = first + (last - first) / 2;
      if (lane == 0) rb_move_median_to_first(L, first, first + 1, mid, last - 1);
      rbw_fence();
      const int cut = rbw_unguarded_partition(L, first + 1, last, lpos, rpos, lane);
      if (cut <= nth) first = cut; else last = cut;
    }
    if (!done) {
      if (lane == 0) rb_insertion_sort(L, first, last);
      rbw_fence();
    }
  }
  const float ambiguous_response = L.r[n_points - 1];
  return rbw_partition_ge(L, n_points, n, ambiguous_response, lpos, rpos, lane);
}
#endif
