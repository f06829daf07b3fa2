// Shared host-side helpers of libpointslot_hip.so: error reporting for the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../../include/pointslot_hip.h"

// Records a formatted message (thread-local) and returns `code`.
int ps_set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define PS_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return ps_set_error(PS_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)

// Host-side staging helper: fn(i) for i in [0, n), spread over short-lived threads when the job moves enough bytes to pay
// for them (packing a batch of problems into the pinned upload buffer is a memcpy-bound loop over independent slices).
template <typename Fn>
inline void ps_parallel_for(int n, size_t total_bytes, Fn fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  static const size_t cap = getenv("PS_PACK_THREADS") ? (size_t)atoi(getenv("PS_PACK_THREADS")) : 16;   // developer switch
  const size_t want = std::min<size_t>(std::min<size_t>((size_t)n, hw ? hw : 1), std::min<size_t>(cap, total_bytes / ((size_t)1 << 20) + 1));
  const int nt = (int)want;
  if (nt <= 1) {
    for (int i = 0; i < n; i++) fn(i);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(nt);
  for (int t = 0; t < nt; t++)
    pool.emplace_back([&fn, t, n, nt]() { for (int i = t; i < n; i += nt) fn(i); });
  for (std::thread& th : pool) th.join();
}
