// Host side of ps_object_ba_batch (include/pointslot_hip.h): packs the collected graphs of a batch of objects
// into one device arena (SoA + CSR edge lists per pose and per point), then drives the device-resident LM state
// machine of ba_kernels.hip: the same kernel sequence is enqueued until every problem reports DONE; the only
// host<->device traffic inside the loop is one 4-byte counter per global step.
// Replaces Optimizer::ObjectLocalBundleAdjustment (/root/reference/src/Optimizer.cc:820-1075; the graph
// collection of :755-818 stays on the host, in the caller).
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "ba_plan.h"
#include "ps_common.h"


extern "C" void psk_ba_global_step(const BaArrays*, int, int, int, int, int, int, int, int, hipStream_t);

struct ps_optimizer;   // defined in opt_host.hip; BA keeps its own arena inside this small side struct
struct BaCtx {
  uint8_t* d_buf = nullptr; size_t d_bytes = 0;
  uint8_t* h_buf = nullptr; size_t h_bytes = 0;
};

// accessors implemented in opt_host.hip
extern "C" int psi_optimizer_device(ps_optimizer* m);
extern "C" hipStream_t psi_optimizer_stream(ps_optimizer* m);
extern "C" BaCtx* psi_optimizer_ba_ctx(ps_optimizer* m);
extern "C" void psi_optimizer_set_ms(ps_optimizer* m, float ms);

namespace {
inline size_t al(size_t v) { return (v + 255) / 256 * 256; }
int ensure(BaCtx* c, size_t dbytes, size_t hbytes) {
  if (dbytes > c->d_bytes) {
    if (c->d_buf) hipFree(c->d_buf);
    c->d_buf = nullptr;
    PS_HIP(hipMalloc(&c->d_buf, dbytes));
    if (const char* fill = getenv("PS_DEBUG_FILL")) {   // diagnostic: poison fresh device memory.  hipMemset on device memory returns before the fill has run, and it runs
      PS_HIP(hipMemset(c->d_buf, atoi(fill), dbytes));                 // on the null stream, which the handle's non-blocking stream does not wait for: without the wait the fill
      PS_HIP(hipDeviceSynchronize());        // landed on top of the call's uploads now and then (r06: 2 of 50 runs of the poisoned test slice died of it)
    }
    c->d_bytes = dbytes;
  }
  if (hbytes > c->h_bytes) {
    if (c->h_buf) hipHostFree(c->h_buf);
    c->h_buf = nullptr;
    PS_HIP(hipHostMalloc(&c->h_buf, hbytes, hipHostMallocDefault));
    c->h_bytes = hbytes;
  }
  return PS_OK;
}
struct Layout {   // byte offsets in the arena; [0, host_end) is mirrored in pinned host memory
  size_t prob, state, poses, flags, points, e_pose, e_point, e_obs, e_is2, e_state, erase, csr_off, csr_edges, trace, ndone, host_end;
  size_t poses_bak, pidx, pact, points_bak, lact, chi2c, Hpp, bp, Hll, bl, Dinv, bs, xp, xl, part, W, S, Wd, end;
};
}  // namespace

extern "C" std::mutex* psi_optimizer_mutex(ps_optimizer* m);
extern "C" int ps_object_ba_batch(ps_optimizer* m, ps_ba_problem* probs, int nprob) {
  if (!m || !probs || nprob < 1) return ps_set_error(PS_ERR_INVALID, "ps_object_ba_batch: bad argument");
  std::lock_guard<std::mutex> lock(*psi_optimizer_mutex(m));   // a handle may be shared by the tracking and the object-mapping thread
  PS_HIP(hipSetDevice(psi_optimizer_device(m)));
  hipStream_t st = psi_optimizer_stream(m);
  BaCtx* ctx = psi_optimizer_ba_ctx(m);
  size_t NP = 0, NL = 0, NE = 0, NW = 0, NS = 0, NPART = 0;
  int max_np = 0, max_nl = 0, max_ne = 0, max_free = 0;
  // Speculative twins (ba_kernels.hip, ba_decide): every problem is in the batch twice - the twin runs the damping trial g2o would run
  // next if the current one is rejected, so that "first trial rejected, second accepted" costs one round of kernels.  PS_BA_TWINS=0: off.
  // r06: up to four members per problem (member s runs the s-th trial ahead).  PS_BA_TWINS=N (2 .. 4) caps the group size.
  static const int twins_max = getenv("PS_BA_TWINS") ? atoi(getenv("PS_BA_TWINS")) : 4;
  static const bool twins_on = twins_max >= 2;
  // (a batch of more than 24 problems fills the chip's CUs with solver workgroups by itself: measured 16 / 32 / 64 objects 0.63 / 0.81 / 1.18 ms per
  // iteration without twins, 0.55 / 0.86 / 1.32 with)
  const int user_nprob = nprob;
  for (int p = 0; p < user_nprob; p++) {
    const ps_ba_problem& P = probs[p];
    if (P.np < 1 || P.nl < 0 || P.ne < 0 || !P.poses7 || !P.pose_flags || (P.nl > 0 && !P.points) ||
        (P.ne > 0 && (!P.e_pose || !P.e_point || !P.e_obs || !P.e_inv_sigma2 || !P.erase)))
      return ps_set_error(PS_ERR_INVALID, "BA problem %d: bad sizes or null pointers", p);
    int nfree = 0;
    for (int i = 0; i < P.np; i++) nfree += (P.pose_flags[i] & 1) ? 0 : 1;
    if (nfree > PS_BA_MAX_POSES) return ps_set_error(PS_ERR_CAPACITY, "BA problem %d: %d free poses (max %d)", p, nfree, PS_BA_MAX_POSES);
    for (int e = 0; e < P.ne; e++)
      if (P.e_pose[e] < 0 || P.e_pose[e] >= P.np || P.e_point[e] < 0 || P.e_point[e] >= P.nl)
        return ps_set_error(PS_ERR_INVALID, "BA problem %d: edge %d references a missing vertex", p, e);
    max_np = P.np > max_np ? P.np : max_np; max_nl = P.nl > max_nl ? P.nl : max_nl; max_ne = P.ne > max_ne ? P.ne : max_ne;
    max_free = nfree > max_free ? nfree : max_free;
    NP += P.np; NL += P.nl; NE += P.ne;
    NW += (size_t)P.np * P.nl * 18;
    NS += (size_t)36 * P.np * P.np;
  }
  // ... and a twin doubles every arena (W: 144 bytes per pose x point, S: 288 per pose x pose) and every grid: only while the doubled W + S
  // stay under a cap (PS_BA_TWINS_MAX_MB, default 1024: BASELINE config 4's 8 objects take 38 MB) - one huge local-BA problem runs single
  static const size_t twins_cap = (size_t)(getenv("PS_BA_TWINS_MAX_MB") ? atoi(getenv("PS_BA_TWINS_MAX_MB")) : 1024) << 20;
  // members per problem: as many as keep the batch at <= PS_BA_GROUP_INSTANCES solver workgroups and the arenas under the cap - but never fewer than
  // the twin for batches <= 24.  Measured: four members pay up to 4 problems in every scene (- 12 % on SURVEY's config 4, where a quarter
  // of the iterations needs three or four trials; + 0 .. 1 % where nearly all need two); at 8 problems they are - 10 % on the first and + 6 % on the
  // second scene (every member adds its Schur complement to a launch that already fills the chip): the default stops at 16 instances.
  static const int inst_cap = getenv("PS_BA_GROUP_INSTANCES") ? atoi(getenv("PS_BA_GROUP_INSTANCES")) : 16;
  int G = 1;
  if (twins_on) {
    for (int g = std::min(twins_max, 4); g >= 2; g--)
      if (nprob * g <= std::max(inst_cap, 2 * std::min(nprob, 24)) && (nprob <= 24) && (size_t)g * (NW + NS) * sizeof(double) <= twins_cap) { G = g; break; }
  }
  const int twins = G > 1 ? 1 : 0;
  if (twins) { NP *= G; NL *= G; NE *= G; NW *= G; NS *= G; nprob *= G; }
  auto user = [&](int p) -> ps_ba_problem& { return probs[p / G]; };
  const int nbp = (max_np + 255) / 256, nbe = (max_ne + 255) / 256;
  const int part_cap = max_np + (max_nl + PS_BA_UPD_PPB - 1) / PS_BA_UPD_PPB + nbp + nbe + 8;
  NPART = (size_t)part_cap * nprob;
  const int nt = (max_free + PS_BA_TILE - 1) / PS_BA_TILE, max_tilepairs = nt * (nt + 1) / 2 > 0 ? nt * (nt + 1) / 2 : 1;

  Layout L;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += al(bytes + 64); return r; };
  L.prob = take(sizeof(BaProb) * nprob); L.state = take(sizeof(BaState) * nprob);
  L.poses = take(NP * 56); L.flags = take(NP); L.points = take(NL * 24);
  L.e_pose = take(NE * 4); L.e_point = take(NE * 4); L.e_obs = take(NE * 12); L.e_is2 = take(NE * 4);
  L.e_state = take(NE); L.erase = take(NE);
  L.csr_off = take((NP + NL + 2 * nprob) * 4); L.csr_edges = take(2 * NE * 4);
  L.trace = take((size_t)nprob * PS_BA_TRACE * 24); L.ndone = take(64);
  L.host_end = o;
  L.poses_bak = take(NP * 56); L.pidx = take(NP * 4); L.pact = take(NP * 4);
  L.points_bak = take(NL * 24); L.lact = take(NL); L.chi2c = take(NE * 8);
  L.Hpp = take(NP * 288); L.bp = take(NP * 48); L.Hll = take(NL * 72); L.bl = take(NL * 24); L.Dinv = take(NL * 72);
  L.bs = take(NP * 48); L.xp = take(NP * 48); L.xl = take(NL * 24); L.part = take(NPART * 8);
  L.W = take(NW * 8); L.S = take(NS * 8); L.Wd = take(NW * 8);
  L.end = o;
  int rc = ensure(ctx, L.end, L.host_end);
  if (rc != PS_OK) return rc;
  uint8_t* H = ctx->h_buf;
  uint8_t* D = ctx->d_buf;
  memset(H, 0, L.host_end);
  BaProb* hp = (BaProb*)(H + L.prob);
  BaState* hs = (BaState*)(H + L.state);
  int32_t* csr_off = (int32_t*)(H + L.csr_off);
  int32_t* csr_edges = (int32_t*)(H + L.csr_edges);
  size_t pb = 0, lb = 0, eb = 0, cb = 0, ceb = 0, wb = 0, sb = 0;
  for (int p = 0; p < nprob; p++) {
    const ps_ba_problem& P = user(p);
    BaProb& d = hp[p];
    d.np = P.np; d.nl = P.nl; d.ne = P.ne;
    d.pose_base = (int32_t)pb; d.point_base = (int32_t)lb; d.edge_base = (int32_t)eb;
    d.csr_pose_base = (int32_t)cb; d.csr_point_base = (int32_t)(cb + P.np + 1);
    d.csr_pose_edges_base = (int32_t)ceb; d.csr_point_edges_base = (int32_t)(ceb + P.ne);
    d.W_base = (int64_t)wb; d.S_base = (int64_t)sb;
    d.part_base = p * part_cap; d.part_cap = part_cap;
    if (twins && (p % G)) { const BaProb& q = hp[p - p % G]; d.lin_pose_base = q.pose_base; d.lin_point_base = q.point_base; d.lin_part_base = q.part_base; d.lin_W_base = q.W_base; }
    else { d.lin_pose_base = d.pose_base; d.lin_point_base = d.point_base; d.lin_part_base = d.part_base; d.lin_W_base = d.W_base; }
    d.fx = P.fx; d.fy = P.fy; d.cx = P.cx; d.cy = P.cy; d.bf = P.bf;
    memcpy(H + L.poses + pb * 56, P.poses7, (size_t)P.np * 56);
    memcpy(H + L.flags + pb, P.pose_flags, P.np);
    if (P.nl) memcpy(H + L.points + lb * 24, P.points, (size_t)P.nl * 24);
    if (P.ne) {
      memcpy(H + L.e_pose + eb * 4, P.e_pose, (size_t)P.ne * 4);
      memcpy(H + L.e_point + eb * 4, P.e_point, (size_t)P.ne * 4);
      memcpy(H + L.e_obs + eb * 12, P.e_obs, (size_t)P.ne * 12);
      memcpy(H + L.e_is2 + eb * 4, P.e_inv_sigma2, (size_t)P.ne * 4);
    }
    uint8_t* est = H + L.e_state + eb;
    for (int e = 0; e < P.ne; e++) est[e] = P.e_obs[3 * e + 2] < 0.f ? 2 : 0;   // ES_MONO
    // CSR lists: edges of every pose / point in edge-insertion order
    int32_t* po = csr_off + cb;
    int32_t* lo = po + P.np + 1;
    for (int e = 0; e < P.ne; e++) { po[P.e_pose[e] + 1]++; lo[P.e_point[e] + 1]++; }
    for (int i = 0; i < P.np; i++) po[i + 1] += po[i];
    for (int l = 0; l < P.nl; l++) lo[l + 1] += lo[l];
    std::vector<int32_t> cur_p(po, po + P.np), cur_l(lo, lo + P.nl);
    for (int e = 0; e < P.ne; e++) {
      csr_edges[ceb + cur_p[P.e_pose[e]]++] = e;
      csr_edges[ceb + P.ne + cur_l[P.e_point[e]]++] = e;
    }
    hs[p].stage = 0; hs[p].phase = BA_PH_BEGIN; hs[p].spec = p % G; hs[p].depth = G - 1;
    pb += P.np; lb += P.nl; eb += P.ne; cb += P.np + P.nl + 2; ceb += 2 * (size_t)P.ne;
    wb += (size_t)P.np * P.nl * 18; sb += (size_t)36 * P.np * P.np;
  }
  PS_HIP(hipMemcpyAsync(D, H, L.host_end, hipMemcpyHostToDevice, st));
  if (const char* fill = getenv("PS_BA_FILL")) PS_HIP(hipMemsetAsync(D + L.host_end, atoi(fill), L.end - L.host_end, st));   // diagnostic: poison the work arrays
  // W must start as zeros: (pose, point) pairs without an edge are never written (see ba_lin_pose)
  PS_HIP(hipMemsetAsync(D + L.W, 0, NW * 8, st));
  PS_HIP(hipMemsetAsync(D + L.Wd, 0, NW * 8, st));   // (the rows of fixed poses are never written and never read)
  PS_HIP(hipMemsetAsync(D + L.chi2c, 0, NE * 8 + 64, st));
  BaArrays A;
  A.prob = (const BaProb*)(D + L.prob); A.state = (BaState*)(D + L.state);
  A.poses = (double*)(D + L.poses); A.poses_bak = (double*)(D + L.poses_bak); A.pose_flags = D + L.flags;
  A.pidx = (int32_t*)(D + L.pidx); A.pact = (int32_t*)(D + L.pact);
  A.points = (double*)(D + L.points); A.points_bak = (double*)(D + L.points_bak); A.lact = D + L.lact;
  A.e_pose = (const int32_t*)(D + L.e_pose); A.e_point = (const int32_t*)(D + L.e_point);
  A.e_obs = (const float*)(D + L.e_obs); A.e_is2 = (const float*)(D + L.e_is2);
  A.e_state = D + L.e_state; A.chi2c = (double*)(D + L.chi2c); A.erase = D + L.erase;
  A.csr_off = (const int32_t*)(D + L.csr_off); A.csr_edges = (const int32_t*)(D + L.csr_edges);
  A.Hpp = (double*)(D + L.Hpp); A.bp = (double*)(D + L.bp); A.Hll = (double*)(D + L.Hll); A.bl = (double*)(D + L.bl);
  A.Dinv = (double*)(D + L.Dinv); A.bs = (double*)(D + L.bs); A.xp = (double*)(D + L.xp); A.xl = (double*)(D + L.xl);
  A.W = (double*)(D + L.W); A.S = (double*)(D + L.S); A.part = (double*)(D + L.part); A.trace = (double*)(D + L.trace); A.Wd = (double*)(D + L.Wd);
  A.ndone = (int32_t*)(D + L.ndone);

  struct EventPair {   // every early return below goes through PS_HIP: the events must not outlive the call
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
  } ev;
  PS_HIP(hipEventCreate(&ev.a));
  PS_HIP(hipEventCreate(&ev.b));
  hipEvent_t e0 = ev.a, e1 = ev.b;
  PS_HIP(hipEventRecord(e0, st));
  int32_t* h_done = (int32_t*)(H + L.ndone);
  int steps = 0;
  static const char* dbg_steps = getenv("PS_BA_DEBUG_MAX_STEPS");   // test knob: provoke the "did not terminate" error return
  const int max_steps = dbg_steps ? atoi(dbg_steps) : 15 * 10 + 8;   // 15 iterations x 10 trials + stage transitions
  // Several global steps are enqueued per host readback: finished problems make their kernels exit immediately, so the
  // only cost of over-enqueueing is a few empty launches at the very end, while every avoided readback saves a
  // stream drain + PCIe round trip.
  const int steps_per_sync = 3;
  for (;;) {
    for (int k = 0; k < steps_per_sync; k++) psk_ba_global_step(&A, nprob, max_np, max_nl, max_ne, max_tilepairs, max_free, steps + k == 0, G, st);
    PS_HIP(hipGetLastError());
    PS_HIP(hipMemcpyAsync(h_done, A.ndone, 4, hipMemcpyDeviceToHost, st));
    PS_HIP(hipStreamSynchronize(st));
    steps += steps_per_sync;
    if (*h_done >= nprob) break;
    if (steps > max_steps) return ps_set_error(PS_ERR_HIP, "object BA did not terminate after %d global steps", steps);
  }
  PS_HIP(hipEventRecord(e1, st));
  PS_HIP(hipMemcpyAsync(H, D, L.host_end, hipMemcpyDeviceToHost, st));
  PS_HIP(hipStreamSynchronize(st));
  float ms = 0;
  PS_HIP(hipEventElapsedTime(&ms, e0, e1));
  psi_optimizer_set_ms(m, ms);
  pb = lb = eb = 0;
  for (int p = 0; p < nprob; p++) {
    ps_ba_problem& P = user(p);
    if (p % G) { pb += P.np; lb += P.nl; eb += P.ne; continue; }   // the results are the primaries'
    memcpy(P.poses7, H + L.poses + pb * 56, (size_t)P.np * 56);
    if (P.nl) memcpy(P.points, H + L.points + lb * 24, (size_t)P.nl * 24);
    int ner = 0;
    if (P.ne) {
      memcpy(P.erase, H + L.erase + eb, P.ne);
      for (int e = 0; e < P.ne; e++) ner += P.erase[e];
    }
    P.n_erased = ner;
    P.iterations = hs[p].iters_done;
    P.trials = hs[p].trials_done;
    if (P.trace) {
      const double* t = (const double*)(H + L.trace) + (size_t)p * PS_BA_TRACE * 3;
      const int nt2 = hs[p].ntrace < PS_BA_TRACE ? hs[p].ntrace : PS_BA_TRACE;
      memcpy(P.trace, t, (size_t)nt2 * 24);
    }
    P.n_trace = hs[p].ntrace < PS_BA_TRACE ? hs[p].ntrace : PS_BA_TRACE;
    pb += P.np; lb += P.nl; eb += P.ne;
  }
  return PS_OK;
}

// Optimizer::LocalBundleAdjustment (/root/reference/src/Optimizer.cc:1077-1417): the static-map BA runs the same
// graph (EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ, marginalised points, LM 5 + outlier pass + 10) with plain
// VertexSE3Expmap keyframes (pose_flags bit 1 clear) and world-frame points, so it is the same solver.
extern "C" int ps_local_ba_batch(ps_optimizer* m, ps_ba_problem* probs, int nprob) { return ps_object_ba_batch(m, probs, nprob); }
