// C ABI of libmgacbam.so (declared in include/mgacbam.h): argument checking, ctx/scratch layout, launch geometry
// and kernel dispatch.  No allocation, no host<->device copy, no synchronisation: every entry point only enqueues
// kernels on the caller's stream, so calls are re-entrant and graph-capturable.
#include "../../include/mgacbam.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include "args.cuh"
#include "bwd.cuh"
#include "common.cuh"
#include "eca.cuh"
#include "segloss.cuh"
#include "gater.cuh"
#include "head.cuh"
#include "fwd.cuh"

using namespace mgacbam;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
// Every launch goes through hipLaunchKernel and its OWN return value is checked: the process-wide sticky error state (which
// may hold an asynchronous error of the framework's kernels on this thread) is neither read nor cleared.
static thread_local hipError_t g_launch_err = hipSuccess;
template <typename K, typename A>
static void launch(K kernel, unsigned grid, size_t smem, hipStream_t st, const A& args) {
  void* p[] = {const_cast<A*>(&args)};
  g_launch_err = hipLaunchKernel(reinterpret_cast<const void*>(kernel), dim3(grid), dim3(kBlock), p, smem, st);
}
static int launch_status(const char* what) {
  const hipError_t e = g_launch_err;
  g_launch_err = hipSuccess;
  if (e != hipSuccess) return fail(static_cast<int>(e), "%s: %s", what, hipGetErrorString(e));
  return 0;
}

extern "C" int mgacbam_abi_version(void) { return MGACBAM_ABI_VERSION; }
extern "C" const char* mgacbam_last_error(void) { return g_err; }
extern "C" const char* mgacbam_build_info(void) {
  return "libmgacbam gfx950 (CDNA4) hip " __VERSION__ " built " __DATE__;
}

// ------------------------------------------------------------------------------------------------
// layouts
// ------------------------------------------------------------------------------------------------
static int check_shape(int B, int C, int H, int W, int hidden, int k) {
  if (B < 1 || C < 1 || H < 1 || W < 1 || hidden < 1 || hidden > 4096 || C > 65536)
    return fail(MGACBAM_E_SHAPE, "bad shape B=%d C=%d H=%d W=%d hidden=%d", B, C, H, W, hidden);
  if (static_cast<long long>(H) * W > (1ll << 30) || static_cast<long long>(B) * C * H * W > (1ll << 40))
    return fail(MGACBAM_E_SHAPE, "tensor too large B=%d C=%d H=%d W=%d", B, C, H, W);
  if (k < 1 || k > 15 || (k & 1) == 0) return fail(MGACBAM_E_SHAPE, "spatial kernel k=%d must be odd and in 1..15", k);
  return 0;
}

// hand-off flags per sample: one per tile of >= kSyncPx pixels, whatever tile size the launch geometry picks
static size_t sync_flags(size_t HW) { return (HW + kSyncPx - 1) / kSyncPx + 1; }

static void ctx_layout(int B, int C, int H, int W, int hidden, mgacbam_ctx_layout_t* L) {
  const size_t HW = static_cast<size_t>(H) * W;
  size_t o = 0;
  auto take = [&](size_t n_elems) { size_t at = o; o = align16(o + n_elems * 4); return static_cast<int64_t>(at); };
  L->S = take(B); L->use = take(B); L->den = take(B);
  L->avg = take(static_cast<size_t>(B) * C); L->mx = take(static_cast<size_t>(B) * C); L->mavg = take(static_cast<size_t>(B) * C);
  L->valid = take(static_cast<size_t>(B) * C); L->amax = take(static_cast<size_t>(B) * C);
  L->h_avg = take(static_cast<size_t>(B) * hidden); L->h_mx = take(static_cast<size_t>(B) * hidden);
  L->ca = take(static_cast<size_t>(B) * C);
  L->planes = take(static_cast<size_t>(B) * 3 * HW);
  L->cidx = take(static_cast<size_t>(B) * HW);
  L->sa = take(static_cast<size_t>(B) * HW);
  L->proj = take(hidden <= MGACBAM_PROJ_MAX_HIDDEN ? static_cast<size_t>(B) * hidden * HW : 0);
  L->sync = take(3 * static_cast<size_t>(B) * sync_flags(HW) + 4 + B);
  L->status = L->sync + static_cast<int64_t>(4 * static_cast<size_t>(B) * sync_flags(HW));   // status word 0 (time-out) follows the k_gate tile flags
  L->total = static_cast<int64_t>(o);
}

static CtxPtrs ctx_ptrs(void* base, int B, int C, int H, int W, int hidden) {
  mgacbam_ctx_layout_t L;
  ctx_layout(B, C, H, W, hidden, &L);
  char* p = static_cast<char*>(base);
  CtxPtrs c;
  c.S = reinterpret_cast<float*>(p + L.S); c.use = reinterpret_cast<float*>(p + L.use); c.den = reinterpret_cast<float*>(p + L.den);
  c.avg = reinterpret_cast<float*>(p + L.avg); c.mx = reinterpret_cast<float*>(p + L.mx); c.mavg = reinterpret_cast<float*>(p + L.mavg);
  c.valid = reinterpret_cast<int*>(p + L.valid); c.amax = reinterpret_cast<int*>(p + L.amax);
  c.h_avg = reinterpret_cast<float*>(p + L.h_avg); c.h_mx = reinterpret_cast<float*>(p + L.h_mx);
  c.ca = reinterpret_cast<float*>(p + L.ca);
  c.planes = reinterpret_cast<float*>(p + L.planes); c.cidx = reinterpret_cast<int*>(p + L.cidx); c.sa = reinterpret_cast<float*>(p + L.sa);
  c.proj = reinterpret_cast<float*>(p + L.proj);
  c.sync = reinterpret_cast<int*>(p + L.sync);
  return c;
}

// ------------------------------------------------------------------------------------------------
// launch geometry
// ------------------------------------------------------------------------------------------------
static int pow2_floor(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }
static int pow2_ceil(int v) { int p = 1; while (p < v) p *= 2; return p; }
static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}
// Tuning / test knobs come from the environment ONCE (first call) -- not per call: the eager path makes ~50 look-ups per step
// otherwise.  mgacbam_reload_env() re-reads them (tests and tuning sweeps that change the environment in-process).
struct Knobs {
  int gate, chan_mintx, pool_tx, pool_cpt, r2_cpt, wsa_fat, chan_tx, chanf_tx, split_mlp, nt, half_vec, gate_h8, level_order, bwd_fold;
  int resident_wgs;        // MGACBAM_RESIDENT_WGS: override of the co-resident workgroup budget the hand-off eligibility is sized from
  int fault;               // MGACBAM_FAULT: fault injection for tests (args.cuh)
  unsigned spin_limit;     // MGACBAM_SPIN_LIMIT
  long long* trace;        // MGACBAM_TRACE_PTR (-DMGACBAM_TRACE builds, tools/trace_gate.py)
};
static Knobs read_knobs() {
  Knobs k;
  k.gate = env_int("MGACBAM_GATE", 1); k.chan_mintx = env_int("MGACBAM_CHAN_MINTX", 16);
  k.pool_tx = env_int("MGACBAM_POOL_TX", 0); k.pool_cpt = env_int("MGACBAM_POOL_CPT", 0); k.chan_tx = env_int("MGACBAM_CHAN_TX", 0);
  k.nt = env_int("MGACBAM_NT", 1); k.half_vec = env_int("MGACBAM_HALF_VEC", 4); k.gate_h8 = env_int("MGACBAM_GATE_H8", 1);
  k.level_order = env_int("MGACBAM_LEVEL_ORDER", 1); k.bwd_fold = env_int("MGACBAM_BWD_FOLD", 1);
  k.r2_cpt = env_int("MGACBAM_R2_CPT", 0); k.wsa_fat = env_int("MGACBAM_WSA_FAT", 0); k.chanf_tx = env_int("MGACBAM_CHANF_TX", 0); k.split_mlp = env_int("MGACBAM_SPLIT_MLP", -1);
  k.resident_wgs = env_int("MGACBAM_RESIDENT_WGS", 0); k.fault = env_int("MGACBAM_FAULT", 0);
  const int sl = env_int("MGACBAM_SPIN_LIMIT", 0);
  k.spin_limit = sl > 0 ? static_cast<unsigned>(sl) : (1u << 20);
  const char* tp = getenv("MGACBAM_TRACE_PTR");
  k.trace = (tp && *tp) ? reinterpret_cast<long long*>(strtoull(tp, nullptr, 0)) : nullptr;
  return k;
}
static std::mutex g_knob_mu;
static Knobs g_knobs;
static std::atomic<bool> g_knobs_ready{false};
static Knobs knobs() {
  if (!g_knobs_ready.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    if (!g_knobs_ready.load(std::memory_order_relaxed)) { g_knobs = read_knobs(); g_knobs_ready.store(true, std::memory_order_release); }
  }
  return g_knobs;   // (written once under the lock before the flag; mgacbam_reload_env is documented as not concurrent with calls)
}
extern "C" void mgacbam_reload_env(void) {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  g_knobs = read_knobs();
  g_knobs_ready.store(true, std::memory_order_release);
}

// Co-resident workgroups of a kernel on the current device = CUs x blocks per CU (occupancy API, for the chosen instantiation and
// its dynamic LDS).  This is what bounds the in-launch hand-off of k_gate: a tile waits for tiles up to 8*span ids AHEAD, and with
// in-order dispatch the lowest unfinished workgroup's producers are dispatched iff 8*span + 1 workgroups fit on the device together.
static int device_cus() {
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lk(mu);
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus[dev] = n > 0 ? n : -1;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}
template <typename K>
static int resident_workgroups(K kernel, size_t smem) {
  const int forced = knobs().resident_wgs;
  if (forced > 0) return forced;
  static std::mutex mu;
  static std::map<std::pair<const void*, size_t>, int> cache;
  const auto key = std::make_pair(reinterpret_cast<const void*>(kernel), smem);
  int per_cu = -1;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) per_cu = it->second;
  }
  if (per_cu < 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kBlock, smem) != hipSuccess) n = 0;
    per_cu = n;
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = per_cu;
  }
  return per_cu * device_cus();
}
static bool is_pow2_in(int v, int lo, int hi) { return v >= lo && v <= hi && (v & (v - 1)) == 0; }

// elements per lane per access: 4 (16 B fp32, 8 B fp16/bf16) when the row length allows, else scalar.  8-element (16 B)
// half vectors exist (MGACBAM_HALF_VEC=8) but halve the tile count, which starves the small levels (P5: 128 workgroups;
// k_chan 19 -> 33 us at YOLOv8n sizes), so they are opt-in for large feature maps only
static int vec_of(int H, int W, int dtype = MGACBAM_F32) {
  const long long hw = static_cast<long long>(H) * W;
  if (dtype != MGACBAM_F32 && hw % 8 == 0 && knobs().half_vec == 8) return 8;
  return hw % 4 == 0 ? 4 : 1;
}

// k_gate (x-resident chan+apply): every thread keeps kGateR channels, so TY = ceil(C / kGateR) slices (power of two) and the
// rest of the 256 threads go along H*W.  Eligible when a tile is >= kSyncPx pixels (ctx.sync has one flag per kSyncPx) and >= one
// image row, the tiles a k x k window reaches are few (their workgroups must be co-resident: 8 ids apart per tile, common.cuh)
// and the staged rows fit in LDS; otherwise the three-launch forward runs.
static void gate_geometry(int C, int H, int W, int k, int VEC, Tune& t) {
  t.gate_tx = 0; t.gate_rows = 0; t.gate_span = 0;
  const int gty = pow2_ceil((C + kGateR - 1) / kGateR);
  if (gty > kBlock || !knobs().gate) return;
  const int gtx = kBlock / gty, TP = gtx * VEC;
  int grows = (TP - 1) / W + 2;
  if (grows > H) grows = H;
  grows += k - 1;
  const int span = ((k / 2) * W + TP - 1) / TP + 1;           // tiles reached on either side
  const size_t lds = (3 * static_cast<size_t>(grows) * (W + k - 1) + TP + 3 * k * k + 3 * C + 64) * sizeof(float);
  // TP >= W: a tile narrower than an image row stages k+1 full rows for a fraction of a row of outputs (measured at
  // 1280-px inputs, C = 256/512: k_gate 220 us against 207 us for k_chan + k_apply; at >= 1.6 rows per tile it wins 18-20 %)
  // (whether the 8*span + 1 workgroups a tile's wait spans are co-resident on THIS device is checked at dispatch: forward_group)
  if (TP >= kSyncPx && TP >= W && lds <= 48 * 1024) { t.gate_tx = gtx; t.gate_rows = grows; t.gate_span = span; }
}

static Tune choose_tune(int B, int C, int H, int W, int k, int dtype = MGACBAM_F32) {
  const int HW = H * W, VEC = vec_of(H, W, dtype), nv = HW / VEC;
  const Knobs kn = knobs();
  Tune t;
  // rows of TX lanes sweep H*W: aim for >= 4 sweeps per lane, then shrink channels/row until the grid fills the chip
  int tx = pow2_floor(nv / 4 > 0 ? nv / 4 : 1);
  if (tx > 256) tx = 256;
  int cpt = 4;
  while (cpt > 1 && static_cast<long long>(B) * ((C + (256 / tx) * cpt - 1) / ((256 / tx) * cpt)) < 1024) cpt /= 2;
  t.pool_tx = tx; t.pool_cpt = cpt;
  // one H*W vector per lane, TY channel slices: TX <= 64 so row reductions are pure wave shuffles
  int ctx = pow2_ceil(nv) < 64 ? pow2_ceil(nv) : 64;
  const int min_tx = kn.chan_mintx;
  while (ctx > min_tx && static_cast<long long>(B) * ((nv + ctx - 1) / ctx) < 768) ctx /= 2;
  while (ctx < 64 && (256 / ctx) * 4 > C) ctx *= 2;       // keep >= 4 channels per row
  t.chan_tx = ctx;
  // conv tiles: full rows when W <= 128, otherwise equal column strips; 4 px per thread
  const int ntx = (W + 127) / 128;
  const int tw = (((W + ntx - 1) / ntx) + 3) / 4 * 4;
  t.conv_twq = tw / 4;
  int th = 256 / t.conv_twq;
  if (th > H) th = H;
  if (th > 64) th = 64;
  const int cap = 24 * 256 / (4 * (tw + k - 1)) - (k - 1);  // 4 staged planes (tile + halo) in <= 24 loads per thread
  if (th > cap) th = cap;
  if (th < 1) th = 1;
  t.conv_th = th;
  int wth = 3500 / (4 * (tw + k - 1)) - (k - 1);                // dWsa tiles: 4 staged planes (tile + halo) <= ~14 KB of LDS
  if (wth > H) wth = H;
  if (wth > 64) wth = 64;
  if (wth < 1) wth = 1;
  t.wsa_th = wth;
  // experiment hooks (tests / tuning sweeps); ignored when not a legal value
  int v;
  if (is_pow2_in(v = kn.pool_tx, 1, 256)) t.pool_tx = v;
  if ((v = kn.pool_cpt) == 1 || v == 2 || v == 4) t.pool_cpt = v;
  if (is_pow2_in(v = kn.chan_tx, 1, 64)) t.chan_tx = v;
  t.chanf_tx = t.chan_tx;
  if (is_pow2_in(v = kn.chanf_tx, 1, 64)) t.chanf_tx = v;
  // k_apply stages every image row its TX*VEC-pixel tile touches, plus the k-1 halo rows
  int rows = (t.chan_tx * VEC - 1) / W + 2;
  if (rows > H) rows = H;
  t.apply_rows = rows + k - 1;
  t.nt_stores = kn.nt ? 1 : 0;
  // k_gate (x-resident chan+apply): every thread keeps kGateR channels, so TY = ceil(C / kGateR) slices (power of two) and the
  // rest of the 256 threads go along H*W.  Eligible when a tile is >= kSyncPx pixels (ctx.sync has one flag per kSyncPx) and >= one image row,
  // the tiles a k x k window reaches are few (their workgroups must be co-resident: 8 ids apart per tile, common.cuh) and
  // the staged rows fit in LDS; otherwise the three-launch forward runs.
  gate_geometry(C, H, W, k, VEC, t);
  return t;
}

static int conv_tiles(const Tune& t, int H, int W) {
  const int TW = t.conv_twq * 4;
  return ((W + TW - 1) / TW) * ((H + t.conv_th - 1) / t.conv_th);
}
static int wsa_tiles(const Tune& t, int H, int W) {
  const int TW = t.conv_twq * 4;
  return ((W + TW - 1) / TW) * ((H + t.wsa_th - 1) / t.wsa_th);
}
static int chan_tiles(const Tune& t, int H, int W, int vec) {
  const int nv = H * W / vec;
  return (nv + t.chan_tx - 1) / t.chan_tx;
}

struct ScratchLayout { size_t A_part, gpre, gplanes, gwsa_part, gz, gbq, gh_avg, gh_mx, pgh, total; };
static ScratchLayout scratch_layout(int B, int C, int H, int W, int hidden, int k) {
  const Tune t = choose_tune(B, C, H, W, k);
  const size_t HW = static_cast<size_t>(H) * W, BC = static_cast<size_t>(B) * C;
  const size_t nt = chan_tiles(t, H, W, vec_of(H, W)), nconv = static_cast<size_t>(B) * wsa_tiles(t, H, W);
  ScratchLayout L;
  size_t o = 0;
  auto take = [&](size_t n_elems) { size_t at = o; o = align16(o + n_elems * 4); return at; };
  L.A_part = take(2 * BC * nt);                          // tile partials of A and Q live together: (B, nt, 2, C)
  L.gpre = take(B * HW); L.gplanes = take(static_cast<size_t>(B) * 3 * HW);
  L.gwsa_part = take(nconv * 3 * k * k);
  L.gz = take(BC); L.gbq = take(BC);
  L.gh_avg = take(static_cast<size_t>(B) * hidden); L.gh_mx = take(static_cast<size_t>(B) * hidden);
  const size_t ty = kBlock / t.pool_tx;                               // channel groups per sample, worst case (1 channel per row)
  L.pgh = take(static_cast<size_t>(B) * ((C + ty - 1) / ty) * hidden);
  L.total = o;
  return L;
}

extern "C" size_t mgacbam_ctx_bytes(int B, int C, int H, int W, int hidden) {
  if (check_shape(B, C, H, W, hidden, 7)) return 0;
  mgacbam_ctx_layout_t L;
  ctx_layout(B, C, H, W, hidden, &L);
  return static_cast<size_t>(L.total);
}
extern "C" int mgacbam_ctx_layout(int B, int C, int H, int W, int hidden, mgacbam_ctx_layout_t* out) {
  if (!out) return fail(MGACBAM_E_NULL, "out is NULL");
  if (int e = check_shape(B, C, H, W, hidden, 7)) return e;
  ctx_layout(B, C, H, W, hidden, out);
  return 0;
}
extern "C" size_t mgacbam_bwd_scratch_bytes(int B, int C, int H, int W, int hidden, int k) {
  if (check_shape(B, C, H, W, hidden, k)) return 0;
  return scratch_layout(B, C, H, W, hidden, k).total;
}

// ------------------------------------------------------------------------------------------------
// dispatch helpers
// ------------------------------------------------------------------------------------------------
static size_t elem_size(int dtype) { return dtype == MGACBAM_F32 ? 4 : 2; }
static bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

#define LAUNCH(kernel, grid, smem, stream, args) launch(kernel, static_cast<unsigned>(grid), smem, stream, args)

// T x VEC
#define DISPATCH_T_VEC(dtype, VECV, CALL)                                                         \
  do {                                                                                            \
    if ((dtype) == MGACBAM_F32) { if ((VECV) == 4) { CALL(float, 4); } else { CALL(float, 1); } }  \
    else if ((dtype) == MGACBAM_F16) { if ((VECV) == 8) { CALL(__half, 8); } else if ((VECV) == 4) { CALL(__half, 4); } else { CALL(__half, 1); } } \
    else { if ((VECV) == 8) { CALL(bf16_t, 8); } else if ((VECV) == 4) { CALL(bf16_t, 4); } else { CALL(bf16_t, 1); } }          \
  } while (0)

#define DISPATCH_CPT(CPTV, CALL2)                                              \
  do { if ((CPTV) == 4) { CALL2(4); } else if ((CPTV) == 2) { CALL2(2); } else { CALL2(1); } } while (0)

static Geo make_geo(int B, int C, int H, int W, const mgacbam_params_t& p) {
  Geo g;
  g.B = B; g.C = C; g.H = H; g.W = W; g.HW = H * W; g.hidden = p.hidden; g.k = p.k;
  g.use_sigmoid = p.use_sigmoid_mask; g.thr = p.tiny_thr; g.eps = p.eps;
  g.proj_h = 0;
  return g;
}
static ParamPtrs make_params(const mgacbam_params_t& p) { return ParamPtrs{p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta}; }
static int check_params(const mgacbam_params_t& p) {
  if (!p.w1 || !p.b1 || !p.w2 || !p.b2 || !p.wsa || !p.beta) return fail(MGACBAM_E_NULL, "NULL parameter pointer");
  return 0;
}

// Levels that share every compile-time property of the kernels (element type, vector width, mask / no mask,
// conv size, dL/dmask wanted) are launched together: one grid per stage, the levels' grids concatenated.
struct Sig {
  int dtype, vec, has_mask, k, gmask, proj;
  bool operator==(const Sig& o) const {
    return dtype == o.dtype && vec == o.vec && has_mask == o.has_mask && k == o.k && gmask == o.gmask && proj == o.proj;
  }
};

// channels per thread for the row-sweep kernels, uniform over a group: 2 when that still gives the chip >= 6 workgroups
// per CU, else 1 (4 is instantiated and reachable through MGACBAM_POOL_CPT, but measured slower at every benchmark shape:
// k_pool 80 us vs 91 us at config 4, 22 vs 26 us at config 2)
template <typename Args>
static int group_cpt(const Args* lv, int n, int max_cpt = 2) {
  const int forced = knobs().pool_cpt;
  if (forced == 1 || forced == 2 || forced == 4) return forced;
  for (int cpt = max_cpt; cpt > 1; cpt /= 2) {
    long long blocks = 0;
    for (int l = 0; l < n; ++l) {
      const int tx = lv[l].t.pool_tx;
      const int cpb = (kBlock / tx) * cpt;
      blocks += static_cast<long long>(lv[l].g.B) * ((lv[l].g.C + cpb - 1) / cpb);   // real workgroups (padding ids exit at once)
    }
    if (blocks >= 1536) return cpt;
  }
  return 1;
}
// grids are XCD-aligned (common.cuh: xcd_sample_part): ceil(B/8)*8 sample slots x parts
static int xcd_grid(int B, int parts) { return ((B + 7) / 8) * 8 * parts; }
static int pad8(int n) { return (n + 7) & ~7; }
template <typename Args>
static int sweep_blocks(const Args& a, int tx, int cpt) {
  const int cpb = (kBlock / tx) * cpt;
  return xcd_grid(a.g.B, (a.g.C + cpb - 1) / cpb);
}
static size_t convT_smem(const Tune& t, int k) {
  return (((3 * k * k + 3) & ~3) + static_cast<size_t>(t.conv_th + k - 1) * (t.conv_twq * 4 + k - 1)) * sizeof(float);
}
static size_t wsa_smem(const Tune& t, int k) {
  return (4 * static_cast<size_t>(t.wsa_th + k - 1) * (t.conv_twq * 4 + k - 1) + static_cast<size_t>(3 * k * t.wsa_th) * k) * sizeof(float);
}
static size_t params_smem(const Geo& g) { return (3 * static_cast<size_t>(g.B) + 2 * kBlock) * sizeof(float); }
static int params_blocks(const Geo& g) { return g.hidden + (g.C + kBlock - 1) / kBlock + (3 * g.k * g.k + 3) / 4 + 1; }
static size_t chan_smem(const Geo& g, int vec, bool proj) {
  return (3 * static_cast<size_t>(g.C) + 2 * g.hidden + (proj ? static_cast<size_t>(g.C) * kProjMax : 0) + 4 * kBlock * vec) * sizeof(float);
}
static size_t apply_smem(const Geo& g, const Tune& t, int vec) {
  return (((3 * g.k * g.k + 3) & ~3) + 3 * static_cast<size_t>(t.apply_rows) * (g.W + g.k - 1) + t.chan_tx * vec + g.C) * sizeof(float);
}
static size_t gate_smem(const Geo& g, const Tune& t, int vec) {
  const size_t head = (g.C + 3) & ~3;
  const size_t role = 3 * static_cast<size_t>(g.C) + 2 * g.hidden;           // gate_role: the MLP's scratch (staged form: [2C][2h][C])
  const size_t conv = ((3 * g.k * g.k + 3) & ~3) + 3 * static_cast<size_t>(t.gate_rows) * (g.W + g.k - 1) + static_cast<size_t>(t.gate_tx) * vec;
  return std::max(role, head + std::max(conv, static_cast<size_t>(3) * kBlock * vec)) * sizeof(float);
}
static int gate_tiles(const Tune& t, int H, int W, int vec) {
  const int nv = H * W / vec;
  return (nv + t.gate_tx - 1) / t.gate_tx;
}
static size_t bwd_apply_smem(const Geo& g, int vec) { return (5 * static_cast<size_t>(g.C) + 2 * g.hidden + kBlock * vec) * sizeof(float); }
static size_t reduce1_smem(const Geo& g, int vec) { return (3 * static_cast<size_t>(g.C) + kBlock * vec) * sizeof(float); }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
static int forward_args(const mgacbam_fwd_level_t& L, FwdArgs& A, Sig& sig) {
  if (!L.x || !L.y || !L.ctx) return fail(MGACBAM_E_NULL, "forward: x / y / ctx is NULL");
  if (int e = check_params(L.p)) return e;
  if (int e = check_shape(L.B, L.C, L.H, L.W, L.p.hidden, L.p.k)) return e;
  if (L.dtype < MGACBAM_F32 || L.dtype > MGACBAM_BF16) return fail(MGACBAM_E_DTYPE, "forward: dtype %d", L.dtype);
  const int VEC = vec_of(L.H, L.W, L.dtype);
  const size_t need = VEC * elem_size(L.dtype);
  if (!aligned_to(L.x, need) || !aligned_to(L.y, need) || !aligned_to(L.ctx, 16) || (L.mask && !aligned_to(L.mask, 16)))
    return fail(MGACBAM_E_ALIGN, "forward: x/y must be %zu-byte aligned, ctx 16-byte, mask %d-byte", need, VEC * 4);
  A.x = L.x; A.mask = L.mask; A.y = L.y; A.fused = 0;
  { const Knobs kn = knobs(); A.trace = kn.trace; A.spin_limit = kn.spin_limit; A.fault = kn.fault; }
  A.nflag = static_cast<int>(sync_flags(static_cast<size_t>(L.H) * L.W));
  A.c = ctx_ptrs(L.ctx, L.B, L.C, L.H, L.W, L.p.hidden);
  A.p = make_params(L.p);
  A.g = make_geo(L.B, L.C, L.H, L.W, L.p);
  A.t = choose_tune(L.B, L.C, L.H, L.W, L.p.k, L.dtype);
  const int proj = (L.flags & MGACBAM_FWD_SAVE_PROJ) && L.mask != nullptr;
  A.g.proj_h = (proj && L.p.hidden <= MGACBAM_PROJ_MAX_HIDDEN) ? L.p.hidden : 0;
  sig = Sig{L.dtype, VEC, L.mask != nullptr, L.p.k, 0, proj};
  return 0;
}

static int forward_group(FwdArgs* lv, int n, const Sig& sig, int stages, hipStream_t st) {
  Group<FwdArgs> G;
  G.n = n;
  const int pool_cpt = group_cpt(lv, n);
  for (int l = 0; l < n; ++l) { lv[l].t.pool_cpt = pool_cpt; G.lv[l] = lv[l]; }
  auto fill = [&](auto blocks_of) { int tot = 0; for (int l = 0; l < n; ++l) { G.start[l] = tot; tot += blocks_of(lv[l]); } G.start[n] = tot; return tot; };

  // MGACBAM_FWD_FUSE: stages 2 + 3 become ONE x-resident launch (k_gate) when every level of the group is eligible
  bool gate = (stages & MGACBAM_FWD_FUSE) && (stages & MGACBAM_FWD_CHAN) && (stages & MGACBAM_FWD_APPLY) && !sig.proj && sig.vec <= 4;
  // fp16 / bf16: k_gate reads 16 bytes per lane (8 elements, kept packed in the registers) whatever vector width the other kernels
  // use -- twice the pixels per tile, half the workgroups: at YOLOv8n sizes the grid then runs as ONE resident round
  int gvec = sig.vec;
  if (gate && sig.dtype != MGACBAM_F32 && sig.vec == 4 && knobs().gate_h8) {
    bool ok8 = true;
    for (int l = 0; l < n && ok8; ++l) {
      Tune t8 = lv[l].t;
      gate_geometry(lv[l].g.C, lv[l].g.H, lv[l].g.W, lv[l].g.k, 8, t8);
      ok8 = (lv[l].g.HW % 8 == 0) && t8.gate_tx > 0;
    }
    if (ok8) {
      gvec = 8;
      for (int l = 0; l < n; ++l) { gate_geometry(lv[l].g.C, lv[l].g.H, lv[l].g.W, lv[l].g.k, 8, lv[l].t); G.lv[l].t = lv[l].t; }
    }
  }
  for (int l = 0; l < n && gate; ++l) gate = lv[l].t.gate_tx > 0;
  size_t gsmem = 0;
  if (gate) {
    // residency precondition of the in-launch hand-off, from the DEVICE (CU count x occupancy of the chosen instantiation): the
    // 8*span + 1 workgroups a tile's wait spans must be co-resident; half of the budget is left to whatever else runs on the chip
    int span = 0;
    for (int l = 0; l < n; ++l) { gsmem = std::max(gsmem, gate_smem(lv[l].g, lv[l].t, gvec)); span = std::max(span, lv[l].t.gate_span); }
    int resident = 0;
#define RES_GATE(Tt, Vv) resident = (sig.k == 7) ? resident_workgroups(k_gate<Tt, Vv, 7>, gsmem) : resident_workgroups(k_gate<Tt, Vv, 0>, gsmem)
    DISPATCH_T_VEC(sig.dtype, gvec, RES_GATE);
#undef RES_GATE
    gate = 2 * (8 * span + 1) <= resident;
  }
  if (gate) for (int l = 0; l < n; ++l) { lv[l].fused = 1; G.lv[l].fused = 1; }

  if (stages & MGACBAM_FWD_POOL) {  // 1. pooling
    const int grid = fill([&](const FwdArgs& a) { return sweep_blocks(a, a.t.pool_tx, pool_cpt); });
#define CALL_POOL2(CPTV) if (sig.has_mask) LAUNCH((k_pool<TT, VV, CPTV, true>), grid, 0, st, G); else LAUNCH((k_pool<TT, VV, CPTV, false>), grid, 0, st, G)
#define CALL_POOL(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(pool_cpt, CALL_POOL2); }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_POOL);
#undef CALL_POOL
#undef CALL_POOL2
    if (int e = launch_status("k_pool")) return e;
  }
  if (gate) {
    const size_t smem = gsmem;
    GateGroup GG;
    const int tiles = fill([&](const FwdArgs& a) { return xcd_grid(a.g.B, gate_tiles(a.t, a.g.H, a.g.W, gvec)); });
    GG.g = G;
    GG.nrole = 0;
    for (int l = 0; l < n; ++l) { GG.rstart[l] = GG.nrole; GG.nrole += lv[l].g.B; }
    GG.rstart[n] = GG.nrole;
    const int grid = GG.nrole + tiles;
#define CALL_GATE(Tt, Vv) if (sig.k == 7) LAUNCH((k_gate<Tt, Vv, 7>), grid, smem, st, GG); else LAUNCH((k_gate<Tt, Vv, 0>), grid, smem, st, GG)
    DISPATCH_T_VEC(sig.dtype, gvec, CALL_GATE);
#undef CALL_GATE
    return launch_status("k_gate");
  }
  if (stages & MGACBAM_FWD_CHAN) {  // 2. shared MLP + channel gate (prologue, or a launch of its own), channel max / mean planes
    // With C*hidden large the MLP prologue keeps every k_chan workgroup from streaming for 15-20 us; one tiny launch per step is cheaper
    bool split_mlp = false;
    for (int l = 0; l < n; ++l) split_mlp |= static_cast<long long>(lv[l].g.C) * lv[l].g.hidden >= 8192;
    { const int f = knobs().split_mlp; if (f == 0) split_mlp = false; else if (f == 1) split_mlp = true; }
    if (split_mlp) {
      size_t msmem = 0;
      for (int l = 0; l < n; ++l) msmem = std::max(msmem, (3 * static_cast<size_t>(lv[l].g.C) + 2 * lv[l].g.hidden) * sizeof(float));
      const int mgrid = fill([&](const FwdArgs& a) { return a.g.B; });
      LAUNCH(k_mlp, mgrid, msmem, st, G);
      if (int e = launch_status("k_mlp")) return e;
    }
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, chan_smem(lv[l].g, sig.vec, sig.proj));
    const int grid = fill([&](const FwdArgs& a) { return xcd_grid(a.g.B, (a.g.HW / sig.vec + a.t.chanf_tx - 1) / a.t.chanf_tx); });
#define CALL_CHAN(Tt, Vv)                                                                                         \
    if (split_mlp) { if (sig.proj) LAUNCH((k_chan<Tt, Vv, true, true>), grid, smem, st, G); else LAUNCH((k_chan<Tt, Vv, false, true>), grid, smem, st, G); } \
    else { if (sig.proj) LAUNCH((k_chan<Tt, Vv, true>), grid, smem, st, G); else LAUNCH((k_chan<Tt, Vv, false>), grid, smem, st, G); }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_CHAN);
#undef CALL_CHAN
    if (int e = launch_status("k_chan")) return e;
  }
  if (stages & MGACBAM_FWD_APPLY) {  // 3. k x k conv + spatial gate (prologue), both gates + alpha residual
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, apply_smem(lv[l].g, lv[l].t, sig.vec));
    const int grid = fill([&](const FwdArgs& a) { return xcd_grid(a.g.B, chan_tiles(a.t, a.g.H, a.g.W, sig.vec)); });
#define CALL_APPLY(Tt, Vv)                                                    \
    switch (sig.k) {                                                          \
      case 3: LAUNCH((k_apply<Tt, Vv, 3>), grid, smem, st, G); break;         \
      case 5: LAUNCH((k_apply<Tt, Vv, 5>), grid, smem, st, G); break;         \
      case 7: LAUNCH((k_apply<Tt, Vv, 7>), grid, smem, st, G); break;         \
      default: LAUNCH((k_apply<Tt, Vv, 0>), grid, smem, st, G); break;        \
    }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_APPLY);
#undef CALL_APPLY
    if (int e = launch_status("k_apply")) return e;
  }
  return 0;
}

// ~ how long a workgroup of the level runs (channels per thread of the tile kernels); levels are launched longest first
template <typename A> static auto level_weight(const A& a, int) -> decltype(a.t.chan_tx, 0) { return a.g.C * a.t.chan_tx; }
template <typename A> static int level_weight(const A& a, long) { return a.g.C; }

// partition the levels into launch groups (same signature, at most kGroupMax levels) and run `run` on each
template <typename Args, typename Run>
static int for_each_group(Args* args, const Sig* sigs, int n, Run run) {
  bool done[MGACBAM_MAX_LEVELS] = {false};
  for (int l = 0; l < n; ++l) {
    if (done[l]) continue;
    Args grp[kGroupMax];
    int m = 0;
    for (int j = l; j < n && m < kGroupMax; ++j)
      if (!done[j] && sigs[j] == sigs[l]) { grp[m++] = args[j]; done[j] = true; }
    if (knobs().level_order)
      std::stable_sort(grp, grp + m, [](const Args& a, const Args& b) {
        return level_weight(a, 0) > level_weight(b, 0);
      });
    if (int e = run(grp, m, sigs[l])) return e;
  }
  return 0;
}

extern "C" int mgacbam_forward_stages(const mgacbam_fwd_level_t* levels, int n_levels, int stages, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  hipStream_t st = static_cast<hipStream_t>(stream);
  FwdArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l)
    if (int e = forward_args(levels[l], args[l], sigs[l])) return e;
  if (int e = for_each_group(args, sigs, n_levels, [&](FwdArgs* g, int m, const Sig& s) { return forward_group(g, m, s, stages, st); })) return e;
  g_err[0] = 0;
  return 0;
}
extern "C" int mgacbam_forward(const mgacbam_fwd_level_t* levels, int n_levels, void* stream) {
  return mgacbam_forward_stages(levels, n_levels, MGACBAM_FWD_ALL, stream);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
static int backward_args(const mgacbam_bwd_level_t& L, BwdArgs& A, Sig& sig) {
  if (!L.x || !L.gy || !L.ctx || !L.scratch || !L.gx) return fail(MGACBAM_E_NULL, "backward: x / gy / ctx / scratch / gx is NULL");
  if (!L.gw1 || !L.gb1 || !L.gw2 || !L.gb2 || !L.gwsa || !L.gbeta) return fail(MGACBAM_E_NULL, "backward: NULL parameter-gradient pointer");
  if (L.gmask && !L.mask) return fail(MGACBAM_E_NULL, "backward: gmask requested but mask is NULL");
  if (int e = check_params(L.p)) return e;
  if (int e = check_shape(L.B, L.C, L.H, L.W, L.p.hidden, L.p.k)) return e;
  if (L.dtype < MGACBAM_F32 || L.dtype > MGACBAM_BF16) return fail(MGACBAM_E_DTYPE, "backward: dtype %d", L.dtype);
  const int VEC = vec_of(L.H, L.W, L.dtype);
  const size_t need = VEC * elem_size(L.dtype);
  if (!aligned_to(L.x, need) || !aligned_to(L.gy, need) || !aligned_to(L.gx, need) || !aligned_to(L.ctx, 16) ||
      !aligned_to(L.scratch, 16) || (L.gmask && !aligned_to(L.gmask, 16)))
    return fail(MGACBAM_E_ALIGN, "backward: x/gy/gx must be %zu-byte aligned, ctx/scratch 16-byte", need);
  A.x = L.x; A.mask = L.mask; A.gy = L.gy; A.gx = L.gx; A.gmask = L.gmask;
  A.gw1 = L.gw1; A.gb1 = L.gb1; A.gw2 = L.gw2; A.gb2 = L.gb2; A.gwsa = L.gwsa; A.gbeta = L.gbeta;
  A.c = ctx_ptrs(const_cast<void*>(L.ctx), L.B, L.C, L.H, L.W, L.p.hidden);
  A.p = make_params(L.p);
  A.g = make_geo(L.B, L.C, L.H, L.W, L.p);
  A.t = choose_tune(L.B, L.C, L.H, L.W, L.p.k, L.dtype);
  const ScratchLayout SL = scratch_layout(L.B, L.C, L.H, L.W, L.p.hidden, L.p.k);
  char* sp = static_cast<char*>(L.scratch);
  A.s.A_part = reinterpret_cast<float*>(sp + SL.A_part);
  A.s.gpre = reinterpret_cast<float*>(sp + SL.gpre); A.s.gplanes = reinterpret_cast<float*>(sp + SL.gplanes);
  A.s.gwsa_part = reinterpret_cast<float*>(sp + SL.gwsa_part);
  A.s.gz = reinterpret_cast<float*>(sp + SL.gz); A.s.gbq = reinterpret_cast<float*>(sp + SL.gbq);
  A.s.gh_avg = reinterpret_cast<float*>(sp + SL.gh_avg); A.s.gh_mx = reinterpret_cast<float*>(sp + SL.gh_mx);
  A.s.pgh = reinterpret_cast<float*>(sp + SL.pgh);
  A.nt = chan_tiles(A.t, A.g.H, A.g.W, VEC);
  A.nconv = A.g.B * conv_tiles(A.t, A.g.H, A.g.W);
  A.nwsa = A.g.B * wsa_tiles(A.t, A.g.H, A.g.W);
  A.nrole = A.nwsa;
  A.npg = params_blocks(A.g);
  A.ncg = 0;
  A.nflag = static_cast<int>(sync_flags(static_cast<size_t>(L.H) * L.W));
  A.bflag0 = L.B * A.nflag + 4 + L.B;
  A.cflag0 = A.bflag0 + L.B * A.nflag;
  A.vec = VEC;
  { const Knobs kn = knobs(); A.trace = kn.trace; A.spin_limit = kn.spin_limit; }
  const int proj = (L.flags & MGACBAM_BWD_HAVE_PROJ) && L.gmask != nullptr;
  A.g.proj_h = (proj && L.p.hidden <= MGACBAM_PROJ_MAX_HIDDEN) ? L.p.hidden : 0;
  sig = Sig{L.dtype, VEC, L.mask != nullptr, L.p.k, L.gmask != nullptr, proj};
  return 0;
}

static int backward_group(BwdArgs* lv, int n, const Sig& sig, int stages, hipStream_t st) {
  Group<BwdArgs> G;
  G.n = n;
  // k_bwd_reduce2 re-reads three planes (g_planes x 2, cidx) per channel group: 4 channels per row halve that share of its loads
  // (config 4: 88 -> 80 us) whenever the grid still fills the chip; k_pool (one mask plane per group) measured slower with 4
  const int r2max = knobs().r2_cpt == 1 || knobs().r2_cpt == 2 || knobs().r2_cpt == 4 ? knobs().r2_cpt : 4;
  const int cpt = group_cpt(lv, n, r2max);
  for (int l = 0; l < n; ++l) {
    lv[l].t.pool_cpt = cpt;
    const int cpb = (kBlock / lv[l].t.pool_tx) * cpt;
    lv[l].ncg = (lv[l].g.C + cpb - 1) / cpb;
    G.lv[l] = lv[l];
  }
  auto fill = [&](auto blocks_of) { int tot = 0; for (int l = 0; l < n; ++l) { G.start[l] = tot; tot += blocks_of(lv[l]); } G.start[n] = tot; return tot; };

  // MGACBAM_BWD_FOLD: transposed conv as trailing role workgroups of the k_bwd_reduce1 launch (whole backward in this call, a tile at
  // least one image row and at least kSyncPx pixels -- one flag per tile in ctx.sync -- and few tiles per conv window)
  // (the conv tiles are the LAST workgroups of the launch and wait only for lower-numbered producers, which never wait themselves:
  //  progress does not depend on residency; the span bound is a speed heuristic)
  bool fold = (stages & MGACBAM_BWD_FOLD) && (stages & MGACBAM_BWD_REDUCE1) && (stages & MGACBAM_BWD_CONVT) && knobs().bwd_fold;
  for (int l = 0; l < n && fold; ++l) {
    const int TP = lv[l].t.chan_tx * sig.vec;
    fold = TP >= kSyncPx && TP >= lv[l].g.W && 8 * (((lv[l].t.conv_th + lv[l].g.k) * lv[l].g.W + TP - 1) / TP + 1) <= 512 &&
           lv[l].nconv <= lv[l].g.B * lv[l].nflag;                 // one flag per conv tile fits the region reserved in ctx.sync
  }
  if (fold) {
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max({smem, reduce1_smem(lv[l].g, sig.vec), convT_smem(lv[l].t, sig.k)});
    const int grid = fill([&](const BwdArgs& a) { return xcd_grid(a.g.B, a.nt) + pad8(a.nconv); });
#define CALL_R1F(Tt, Vv) if (sig.k == 7) LAUNCH((k_bwd_reduce1_fold<Tt, Vv, 7>), grid, smem, st, G); else LAUNCH((k_bwd_reduce1_fold<Tt, Vv, 0>), grid, smem, st, G)
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_R1F);
#undef CALL_R1F
    if (int e = launch_status("k_bwd_reduce1_fold")) return e;
  }
  if ((stages & MGACBAM_BWD_REDUCE1) && !fold) {  // 1. per-(b,c) and per-pixel reductions of gy*x
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, reduce1_smem(lv[l].g, sig.vec));
    const int grid = fill([&](const BwdArgs& a) { return xcd_grid(a.g.B, a.nt); });
#define CALL_R1(Tt, Vv) LAUNCH((k_bwd_reduce1<Tt, Vv>), grid, smem, st, G)
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_R1);
#undef CALL_R1
    if (int e = launch_status("k_bwd_reduce1")) return e;
  }
  if ((stages & MGACBAM_BWD_CONVT) && !fold) {  // 2. transposed conv
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, convT_smem(lv[l].t, sig.k));
    const int grid = fill([&](const BwdArgs& a) { return a.nconv; });
    switch (sig.k) {
      case 3: LAUNCH(k_bwd_convT<3>, grid, smem, st, G); break;
      case 5: LAUNCH(k_bwd_convT<5>, grid, smem, st, G); break;
      case 7: LAUNCH(k_bwd_convT<7>, grid, smem, st, G); break;
      default: LAUNCH(k_bwd_convT<0>, grid, smem, st, G); break;
    }
    if (int e = launch_status("k_bwd_convT")) return e;
  }
  const bool fuse = (stages & MGACBAM_BWD_FUSE) != 0;
  const bool fuse_wsa = fuse && (stages & MGACBAM_BWD_REDUCE2) && (stages & MGACBAM_BWD_WSA) && sig.k == 7;
  const bool fuse_pg = fuse && (stages & MGACBAM_BWD_APPLY) && (stages & MGACBAM_BWD_PARAMGRAD);
  if (stages & MGACBAM_BWD_REDUCE2) {  // 3. rest of g_ca (needs g_planes), g_z [+ dWsa partials as role workgroups]
    size_t smem = 0;
    for (int l = 0; l < n; ++l) {
      smem = std::max(smem, (64 + static_cast<size_t>(std::max(kPghLds, kBlock / lv[l].t.pool_tx))) * sizeof(float));
      if (fuse_wsa) smem = std::max(smem, wsa_smem(lv[l].t, sig.k));
    }
    if (fuse_wsa && knobs().wsa_fat) {
      // experiment (MGACBAM_WSA_FAT=1, off by default): as many dWsa roles as the streaming workgroups leave slots idle (config 2: 1792 of
      // 2048), each working through several tiles, instead of one thin role per tile.  Measured at configs 2 and 3: no change (26.9-27.2 us
      // either way) -- what the roles add to the launch (5.5 us over the role-free kernel) is not slot displacement
      int slots = 0;
#define RES_R22(CPTV) slots = resident_workgroups(k_bwd_reduce2<TT, VV, CPTV, true>, smem)
#define RES_R2(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(cpt, RES_R22); }
      DISPATCH_T_VEC(sig.dtype, sig.vec, RES_R2);
#undef RES_R2
#undef RES_R22
      long long streaming = 0, tiles = 0;
      for (int l = 0; l < n; ++l) {
        const int cpb = (kBlock / lv[l].t.pool_tx) * cpt;
        streaming += static_cast<long long>(lv[l].g.B) * ((lv[l].g.C + cpb - 1) / cpb);
        tiles += lv[l].nwsa;
      }
      const long long idle = slots - streaming;
      if (idle >= 32 && idle < tiles) {
        for (int l = 0; l < n; ++l) {
          lv[l].nrole = static_cast<int>(std::max(1ll, std::min<long long>(lv[l].nwsa, idle * lv[l].nwsa / tiles)));
          G.lv[l].nrole = lv[l].nrole;
        }
      }
    }
    const int grid = fill([&](const BwdArgs& a) { return (fuse_wsa ? pad8(a.nrole) : 0) + sweep_blocks(a, a.t.pool_tx, cpt); });
#define CALL_R22(CPTV) if (fuse_wsa) LAUNCH((k_bwd_reduce2<TT, VV, CPTV, true>), grid, smem, st, G); else LAUNCH((k_bwd_reduce2<TT, VV, CPTV, false>), grid, smem, st, G)
#define CALL_R2(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(cpt, CALL_R22); }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_R2);
#undef CALL_R2
#undef CALL_R22
    if (int e = launch_status("k_bwd_reduce2")) return e;
  }
  if ((stages & MGACBAM_BWD_WSA) && !fuse_wsa) {  // 4. dWsa tile partials (depends on stage 1 only)
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, wsa_smem(lv[l].t, sig.k));
    const int grid = fill([&](const BwdArgs& a) { return a.nwsa; });
    switch (sig.k) {
      case 3: LAUNCH(k_bwd_wsa<3>, grid, smem, st, G); break;
      case 5: LAUNCH(k_bwd_wsa<5>, grid, smem, st, G); break;
      case 7: LAUNCH(k_bwd_wsa<7>, grid, smem, st, G); break;
      default: LAUNCH(k_bwd_wsa<0>, grid, smem, st, G); break;
    }
    if (int e = launch_status("k_bwd_wsa")) return e;
  }
  if ((stages & MGACBAM_BWD_PARAMGRAD) && !fuse_pg) {  // 5. every parameter gradient
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, params_smem(lv[l].g));
    const int grid = fill([&](const BwdArgs& a) { return a.npg; });
    LAUNCH(k_bwd_params, grid, smem, st, G);
    if (int e = launch_status("k_bwd_params")) return e;
  }
  if (stages & MGACBAM_BWD_APPLY) {  // 6. gx (+ gmask) [+ parameter gradients as role workgroups]
    size_t smem = 0;
    for (int l = 0; l < n; ++l) {
      smem = std::max(smem, bwd_apply_smem(lv[l].g, sig.vec));
      if (fuse_pg) smem = std::max(smem, params_smem(lv[l].g));
    }
    const int grid = fill([&](const BwdArgs& a) { return (fuse_pg ? pad8(a.npg) : 0) + xcd_grid(a.g.B, a.nt); });
#define CALL_AP2(GM) if (fuse_pg) LAUNCH((k_bwd_apply<TT, VV, GM, true>), grid, smem, st, G); else LAUNCH((k_bwd_apply<TT, VV, GM, false>), grid, smem, st, G)
#define CALL_AP(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; if (sig.gmask) { CALL_AP2(true); } else { CALL_AP2(false); } }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_AP);
#undef CALL_AP
#undef CALL_AP2
    if (int e = launch_status("k_bwd_apply")) return e;
  }
  return 0;
}

extern "C" int mgacbam_backward_stages(const mgacbam_bwd_level_t* levels, int n_levels, int stages, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  hipStream_t st = static_cast<hipStream_t>(stream);
  BwdArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l)
    if (int e = backward_args(levels[l], args[l], sigs[l])) return e;
  if (int e = for_each_group(args, sigs, n_levels, [&](BwdArgs* g, int m, const Sig& s) { return backward_group(g, m, s, stages, st); })) return e;
  g_err[0] = 0;
  return 0;
}
extern "C" int mgacbam_backward(const mgacbam_bwd_level_t* levels, int n_levels, void* stream) {
  return mgacbam_backward_stages(levels, n_levels, MGACBAM_BWD_ALL, stream);
}

// ------------------------------------------------------------------------------------------------
// MaskECA
// ------------------------------------------------------------------------------------------------
struct EcaCtxLayout { size_t S, use, den, avg, mavg, w, splane, total; };
static EcaCtxLayout eca_ctx_layout(int B, int C, int H, int W) {
  const size_t HW = static_cast<size_t>(H) * W, BC = static_cast<size_t>(B) * C;
  EcaCtxLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o = align16(o + n * 4); return at; };
  L.S = take(B); L.use = take(B); L.den = take(B);
  L.avg = take(BC); L.mavg = take(BC); L.w = take(BC);
  L.splane = take(B * HW);
  L.total = o;
  return L;
}
static EcaCtx eca_ctx_ptrs(void* base, int B, int C, int H, int W) {
  const EcaCtxLayout L = eca_ctx_layout(B, C, H, W);
  char* p = static_cast<char*>(base);
  auto f = [&](size_t off) { return reinterpret_cast<float*>(p + off); };
  return EcaCtx{f(L.S), f(L.use), f(L.den), f(L.avg), f(L.mavg), f(L.w), f(L.splane)};
}
extern "C" size_t mgacbam_eca_ctx_bytes(int B, int C, int H, int W) {
  if (check_shape(B, C, H, W, 1, 3)) return 0;
  return eca_ctx_layout(B, C, H, W).total;
}
extern "C" size_t mgacbam_eca_scratch_bytes(int B, int C, int H, int W) {
  if (check_shape(B, C, H, W, 1, 3)) return 0;
  return align16(static_cast<size_t>(B) * C * 4);
}
static Geo eca_geo(int B, int C, int H, int W, const mgacbam_eca_params_t& p) {
  Geo g;
  g.B = B; g.C = C; g.H = H; g.W = W; g.HW = H * W; g.hidden = 1; g.k = p.k;
  g.use_sigmoid = p.use_sigmoid_mask; g.thr = p.tiny_thr; g.eps = p.eps; g.proj_h = 0;
  return g;
}

static int eca_forward_group(EcaFwdArgs* lv, int n, const Sig& sig, hipStream_t st) {
  Group<EcaFwdArgs> G;
  G.n = n;
  const int cpt = group_cpt(lv, n);
  for (int l = 0; l < n; ++l) { lv[l].t.pool_cpt = cpt; G.lv[l] = lv[l]; }
  auto fill = [&](auto blocks_of) { int tot = 0; for (int l = 0; l < n; ++l) { G.start[l] = tot; tot += blocks_of(lv[l]); } G.start[n] = tot; return tot; };
  const int grid = fill([&](const EcaFwdArgs& a) { return sweep_blocks(a, a.t.pool_tx, cpt); });
#define CALL_EP2(CPTV) if (sig.has_mask) LAUNCH((k_eca_pool<TT, VV, CPTV, true>), grid, 0, st, G); else LAUNCH((k_eca_pool<TT, VV, CPTV, false>), grid, 0, st, G)
#define CALL_EP(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(cpt, CALL_EP2); }
  DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_EP);
#undef CALL_EP
#undef CALL_EP2
  if (int e = launch_status("k_eca_pool")) return e;
#define CALL_EA2(CPTV) LAUNCH((k_eca_apply<TT, VV, CPTV>), grid, 0, st, G)
#define CALL_EA(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(cpt, CALL_EA2); }
  DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_EA);
#undef CALL_EA
#undef CALL_EA2
  return launch_status("k_eca_apply");
}

static int eca_check_params(const mgacbam_eca_params_t& p) {
  if (!p.w || !p.beta) return fail(MGACBAM_E_NULL, "eca: NULL parameter pointer");
  if (p.k < 1 || p.k > 15 || (p.k & 1) == 0) return fail(MGACBAM_E_SHAPE, "eca: conv1d kernel k=%d must be odd and in 1..15", p.k);
  return 0;
}

extern "C" int mgacbam_eca_forward(const mgacbam_eca_fwd_level_t* levels, int n_levels, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  EcaFwdArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    const mgacbam_eca_fwd_level_t& L = levels[l];
    if (!L.x || !L.y || !L.ctx) return fail(MGACBAM_E_NULL, "eca forward: x / y / ctx is NULL");
    if (int e = eca_check_params(L.p)) return e;
    if (int e = check_shape(L.B, L.C, L.H, L.W, 1, L.p.k)) return e;
    if (L.dtype < MGACBAM_F32 || L.dtype > MGACBAM_BF16) return fail(MGACBAM_E_DTYPE, "eca forward: dtype %d", L.dtype);
    const int VEC = vec_of(L.H, L.W, L.dtype);
    const size_t need = VEC * elem_size(L.dtype);
    if (!aligned_to(L.x, need) || !aligned_to(L.y, need) || !aligned_to(L.ctx, 16) || (L.mask && !aligned_to(L.mask, 16)))
      return fail(MGACBAM_E_ALIGN, "eca forward: x/y must be %zu-byte aligned, ctx and mask 16-byte", need);
    EcaFwdArgs& A = args[l];
    A.x = L.x; A.mask = L.mask; A.y = L.y;
    A.c = eca_ctx_ptrs(L.ctx, L.B, L.C, L.H, L.W);
    A.w1d = L.p.w; A.beta = L.p.beta;
    A.g = eca_geo(L.B, L.C, L.H, L.W, L.p);
    A.t = choose_tune(L.B, L.C, L.H, L.W, 7, L.dtype);
    sigs[l] = Sig{L.dtype, VEC, L.mask != nullptr, 0, 0, 0};
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int e = for_each_group(args, sigs, n_levels, [&](EcaFwdArgs* g, int m, const Sig& s) { return eca_forward_group(g, m, s, st); })) return e;
  g_err[0] = 0;
  return 0;
}

static int eca_backward_group(EcaBwdArgs* lv, int n, const Sig& sig, hipStream_t st) {
  Group<EcaBwdArgs> G;
  G.n = n;
  const int cpt = group_cpt(lv, n);
  for (int l = 0; l < n; ++l) { lv[l].t.pool_cpt = cpt; G.lv[l] = lv[l]; }
  auto fill = [&](auto blocks_of) { int tot = 0; for (int l = 0; l < n; ++l) { G.start[l] = tot; tot += blocks_of(lv[l]); } G.start[n] = tot; return tot; };
  {
    const int grid = fill([&](const EcaBwdArgs& a) { return sweep_blocks(a, a.t.pool_tx, cpt); });
#define CALL_ER2(CPTV) LAUNCH((k_eca_reduce<TT, VV, CPTV>), grid, 0, st, G)
#define CALL_ER(Tt, Vv) { using TT = Tt; constexpr int VV = Vv; DISPATCH_CPT(cpt, CALL_ER2); }
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_ER);
#undef CALL_ER
#undef CALL_ER2
    if (int e = launch_status("k_eca_reduce")) return e;
  }
  {
    size_t smem = 0;
    for (int l = 0; l < n; ++l) smem = std::max(smem, (3 * static_cast<size_t>(lv[l].g.C) + kBlock * sig.vec) * sizeof(float));
    const int grid = fill([&](const EcaBwdArgs& a) { return kEcaRoles + xcd_grid(a.g.B, a.nt); });
#define CALL_EB(Tt, Vv) if (sig.gmask) LAUNCH((k_eca_bwd<Tt, Vv, true>), grid, smem, st, G); else LAUNCH((k_eca_bwd<Tt, Vv, false>), grid, smem, st, G)
    DISPATCH_T_VEC(sig.dtype, sig.vec, CALL_EB);
#undef CALL_EB
    if (int e = launch_status("k_eca_bwd")) return e;
  }
  return 0;
}

extern "C" int mgacbam_eca_backward(const mgacbam_eca_bwd_level_t* levels, int n_levels, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  EcaBwdArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    const mgacbam_eca_bwd_level_t& L = levels[l];
    if (!L.x || !L.gy || !L.ctx || !L.scratch || !L.gx || !L.gw || !L.gbeta) return fail(MGACBAM_E_NULL, "eca backward: NULL pointer");
    if (L.gmask && !L.mask) return fail(MGACBAM_E_NULL, "eca backward: gmask requested but mask is NULL");
    if (int e = eca_check_params(L.p)) return e;
    if (int e = check_shape(L.B, L.C, L.H, L.W, 1, L.p.k)) return e;
    if (L.dtype < MGACBAM_F32 || L.dtype > MGACBAM_BF16) return fail(MGACBAM_E_DTYPE, "eca backward: dtype %d", L.dtype);
    const int VEC = vec_of(L.H, L.W, L.dtype);
    const size_t need = VEC * elem_size(L.dtype);
    if (!aligned_to(L.x, need) || !aligned_to(L.gy, need) || !aligned_to(L.gx, need) || !aligned_to(L.ctx, 16) ||
        !aligned_to(L.scratch, 16) || (L.gmask && !aligned_to(L.gmask, 16)))
      return fail(MGACBAM_E_ALIGN, "eca backward: x/gy/gx must be %zu-byte aligned, ctx/scratch/gmask 16-byte", need);
    EcaBwdArgs& A = args[l];
    A.x = L.x; A.mask = L.mask; A.gy = L.gy; A.gx = L.gx; A.gmask = L.gmask; A.gw = L.gw; A.gbeta = L.gbeta;
    A.c = eca_ctx_ptrs(const_cast<void*>(L.ctx), L.B, L.C, L.H, L.W);
    A.w1d = L.p.w; A.beta = L.p.beta;
    A.s.gg = static_cast<float*>(L.scratch);
    A.g = eca_geo(L.B, L.C, L.H, L.W, L.p);
    A.t = choose_tune(L.B, L.C, L.H, L.W, 7, L.dtype);
    A.nt = chan_tiles(A.t, L.H, L.W, VEC);
    sigs[l] = Sig{L.dtype, VEC, L.mask != nullptr, 0, L.gmask != nullptr, 0};
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int e = for_each_group(args, sigs, n_levels, [&](EcaBwdArgs* g, int m, const Sig& s) { return eca_backward_group(g, m, s, st); })) return e;
  g_err[0] = 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// nearest-neighbour resize (integer index path)
// ------------------------------------------------------------------------------------------------
extern "C" int mgacbam_resize_nearest(const float* src, float* dst, int n_planes, int in_h, int in_w, int out_h, int out_w,
                                      void* stream) {
  if (!src || !dst) return fail(MGACBAM_E_NULL, "resize: NULL pointer");
  if (n_planes < 1 || in_h < 1 || in_w < 1 || out_h < 1 || out_w < 1) return fail(MGACBAM_E_SHAPE, "resize: bad shape");
  const size_t total = static_cast<size_t>(n_planes) * out_h * out_w;
  size_t grid = (total + kBlock - 1) / kBlock;
  if (grid > 4096) grid = 4096;
  void* kargs[] = {&src, &dst, &n_planes, &in_h, &in_w, &out_h, &out_w};
  g_launch_err = hipLaunchKernel(reinterpret_cast<const void*>(k_resize_nearest), dim3(static_cast<unsigned>(grid)), dim3(kBlock), kargs, 0,
                                 static_cast<hipStream_t>(stream));
  if (int e = launch_status("k_resize_nearest")) return e;
  g_err[0] = 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// segmentation loss
// ------------------------------------------------------------------------------------------------
static int seg_check(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, bool bwd) {
  if (!levels || !cfg) return fail(MGACBAM_E_NULL, "segloss: NULL argument");
  if (n < 1 || n > MGASEG_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "segloss: n_levels=%d", n);
  for (int l = 0; l < n; ++l) {
    const mgaseg_level_t& L = levels[l];
    if (!L.logits || !L.target || (bwd && !L.glogits)) return fail(MGACBAM_E_NULL, "segloss: level %d has a NULL pointer", l);
    if (L.B < 1 || L.H < 1 || L.W < 1 || L.Ht < 1 || L.Wt < 1 || static_cast<long long>(L.H) * L.W > (1ll << 30))
      return fail(MGACBAM_E_SHAPE, "segloss: level %d bad shape B=%d H=%d W=%d Ht=%d Wt=%d", l, L.B, L.H, L.W, L.Ht, L.Wt);
    if (L.dtype != levels[0].dtype || L.dtype < MGACBAM_F32 || L.dtype > MGACBAM_BF16) return fail(MGACBAM_E_DTYPE, "segloss: dtype %d", L.dtype);
    if (L.resize != MGASEG_NEAREST && L.resize != MGASEG_BILINEAR) return fail(MGACBAM_E_SHAPE, "segloss: level %d resize mode %d", l, L.resize);
  }
  return 0;
}
static size_t seg_ws_level(int B) { return align16(static_cast<size_t>(B) * (kSegParts + 1) * 4 * sizeof(float)); }
extern "C" size_t mgaseg_ws_bytes(const mgaseg_level_t* levels, int n) {
  if (!levels || n < 1 || n > MGASEG_MAX_LEVELS) { fail(MGACBAM_E_LEVELS, "segloss: n_levels=%d", n); return 0; }
  size_t tot = 0;
  for (int l = 0; l < n; ++l) tot += seg_ws_level(levels[l].B);
  return tot;
}
static int seg_args(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, void* ws, SegArgs& A) {
  A.n = n;
  char* w = static_cast<char*>(ws);
  int tot = 0;
  for (int l = 0; l < n; ++l) {
    const mgaseg_level_t& L = levels[l];
    SegLevel& S = A.lv[l];
    S.logits = L.logits; S.target = L.target; S.glogits = L.glogits;
    S.part = reinterpret_cast<float*>(w);
    S.sums = S.part + static_cast<size_t>(L.B) * kSegParts * 4;
    w += seg_ws_level(L.B);
    S.B = L.B; S.H = L.H; S.W = L.W; S.Ht = L.Ht; S.Wt = L.Wt; S.w_scale = L.scale_weight;
    S.bilinear = L.resize == MGASEG_BILINEAR ? 1 : 0;
    A.start[l] = tot;
    tot += L.B * kSegParts;
  }
  A.start[n] = tot;
  A.w_bce = cfg->bce_weight; A.w_dice = cfg->dice_weight; A.smooth = cfg->smooth; A.lambda = cfg->loss_lambda;
  A.ufl = cfg->use_unified_focal ? 1 : 0; A.u_lambda = cfg->ufl_lambda; A.u_delta = cfg->ufl_delta; A.u_gamma = cfg->ufl_gamma;
  A.out = nullptr; A.gout = nullptr; A.has_kd = 0; memset(&A.kd, 0, sizeof(A.kd));
  return tot;
}
static int seg_forward_impl(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, void* ws, float* out, const KendallArgs* kd, void* stream) {
  if (int e = seg_check(levels, n, cfg, false)) return e;
  if (!ws || !out) return fail(MGACBAM_E_NULL, "segloss: ws / out is NULL");
  SegArgs A;
  const int grid = seg_args(levels, n, cfg, ws, A);
  A.out = out;
  if (kd) { A.has_kd = 1; A.kd = *kd; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (levels[0].dtype) {
    case MGACBAM_F32: LAUNCH(k_seg_partial<float>, grid, 0, st, A); break;
    case MGACBAM_F16: LAUNCH(k_seg_partial<__half>, grid, 0, st, A); break;
    default: LAUNCH(k_seg_partial<bf16_t>, grid, 0, st, A); break;
  }
  if (int e = launch_status("k_seg_partial")) return e;
  LAUNCH(k_seg_final, 1, 0, st, A);
  if (int e = launch_status("k_seg_final")) return e;
  g_err[0] = 0;
  return 0;
}
static int seg_backward_impl(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, const void* ws, const float* gout, const KendallArgs* kd, void* stream) {
  if (int e = seg_check(levels, n, cfg, true)) return e;
  if (!ws || (!gout && !kd)) return fail(MGACBAM_E_NULL, "segloss: ws / gout is NULL");
  SegArgs A;
  const int grid = seg_args(levels, n, cfg, const_cast<void*>(ws), A);
  A.gout = gout;
  if (kd) { A.has_kd = 1; A.kd = *kd; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (levels[0].dtype) {
    case MGACBAM_F32: LAUNCH(k_seg_bwd<float>, grid, 0, st, A); break;
    case MGACBAM_F16: LAUNCH(k_seg_bwd<__half>, grid, 0, st, A); break;
    default: LAUNCH(k_seg_bwd<bf16_t>, grid, 0, st, A); break;
  }
  if (int e = launch_status("k_seg_bwd")) return e;
  g_err[0] = 0;
  return 0;
}
extern "C" int mgaseg_forward(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, void* ws, float* out, void* stream) {
  return seg_forward_impl(levels, n, cfg, ws, out, nullptr, stream);
}
extern "C" int mgaseg_backward(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, const void* ws, const float* gout, void* stream) {
  return seg_backward_impl(levels, n, cfg, ws, gout, nullptr, stream);
}
static int kendall_check(const float* det, int n_det, const float* log_vars) {
  if (!det || !log_vars) return fail(MGACBAM_E_NULL, "kendall: NULL pointer");
  if (n_det < 1 || n_det > 4096) return fail(MGACBAM_E_SHAPE, "kendall: n_det=%d", n_det);
  return 0;
}
extern "C" int mgaseg_kendall_forward(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, void* ws, float* out,
                                      const float* det, int n_det, const float* log_vars, float* total, void* stream) {
  if (int e = kendall_check(det, n_det, log_vars)) return e;
  if (!total) return fail(MGACBAM_E_NULL, "kendall: total is NULL");
  const KendallArgs kd{det, out, log_vars, nullptr, total, nullptr, nullptr, nullptr, n_det};
  return seg_forward_impl(levels, n, cfg, ws, out, &kd, stream);
}
extern "C" int mgaseg_kendall_backward(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, const void* ws, const float* out,
                                       const float* det, int n_det, const float* log_vars, const float* g_total,
                                       float* g_det, float* g_seg, float* g_log_vars, void* stream) {
  if (int e = kendall_check(det, n_det, log_vars)) return e;
  if (!out || !g_total || !g_det || !g_log_vars) return fail(MGACBAM_E_NULL, "kendall: NULL pointer");
  const KendallArgs kd{det, out, log_vars, g_total, nullptr, g_det, g_seg, g_log_vars, n_det};
  return seg_backward_impl(levels, n, cfg, ws, nullptr, &kd, stream);
}

// ------------------------------------------------------------------------------------------------
// MGAMaskHead
// ------------------------------------------------------------------------------------------------
static int head_check_shape(int B, int C, int H, int W, int hidden) {
  if (B < 1 || C < 1 || H < 1 || W < 1 || hidden < 1 || hidden > 1024 || C > 8192)
    return fail(MGACBAM_E_SHAPE, "mask head: bad shape B=%d C=%d H=%d W=%d hidden=%d", B, C, H, W, hidden);
  if (static_cast<long long>(H) * W > (1ll << 28) || static_cast<long long>(B) * std::max(C, hidden) * H * W > (1ll << 40))
    return fail(MGACBAM_E_SHAPE, "mask head: tensor too large B=%d C=%d H=%d W=%d", B, C, H, W);
  return 0;
}
struct HeadTiling { int vec, hidp, cp, tile_px, tps, nwg, gx_tile_px, gx_tps, fw_kw, gx_kw, fw_mtw, gx_mtw, nwg_out, nwg1, act_ppt, act_hl, ncb, nshare, gw2; };
// wave arrangement of k_head_gemm (head.cuh): MW waves along M for `mtiles` 16-output tiles, KW waves along K when K is long (a chain of
// K/4 dependent steps otherwise), the rest along pixels
static void head_waves(int mtiles, int mtw, int K, int& pw, int& kw) {
  int mw = std::min(4, (mtiles + mtw - 1) / mtw);
  if (mw == 3) mw = 4;
  const int rest = 4 / mw;
  kw = K >= 128 ? rest : 1;
  pw = rest / kw;
}
static HeadTiling head_tiling(int B, int C, int H, int W, int hidden) {
  HeadTiling t;
  const int HW = H * W;
  t.vec = (HW % 4 == 0) ? 4 : 1;
  t.hidp = (hidden + 15) & ~15; t.cp = (C + 15) & ~15;
  int pw;
  t.fw_mtw = t.hidp / 16 > 8 ? 4 : 2;                               // forward: two accumulator tiles per wave (120 VGPRs, 4 workgroups per CU, every level of a YOLOv8n/s call in ONE launch); hidden > 128: four
  t.gx_mtw = 2;                                                     // gx: its B operand (g_a, z: E/4 each) is cheap to re-read; light workgroups
  head_waves(t.hidp / 16, t.fw_mtw, C, pw, t.fw_kw);
  t.tile_px = pw * 16 * t.vec;
  t.tps = (HW + t.tile_px - 1) / t.tile_px;
  t.nwg = B * t.tps;
  head_waves(t.cp / 16, t.gx_mtw, hidden, pw, t.gx_kw);
  t.gx_tile_px = pw * 16 * t.vec;
  t.gx_tps = (HW + t.gx_tile_px - 1) / t.gx_tile_px;
  { int opx, ojo; head_out_shape(hidden, opx, ojo); t.nwg_out = B * ((HW + opx - 1) / opx); }
  t.act_ppt = HW >= 2048 ? 4 : (HW >= 512 ? 2 : 1);             // pixels per thread of k_head_bwd_act (amortises its per-channel reductions)
  t.nwg1 = B * ((HW + kBlock * t.act_ppt - 1) / (kBlock * t.act_ppt));
  t.act_hl = kBlock * t.act_ppt + 2 * (W + 1);
  t.ncb = (C + kHeadCB - 1) / kHeadCB;
  // pixel shares of k_head_bwd_gw: as many workgroups as ~4 MB of dW1 partials allow (32..256), and no more than there are pairs of pixel chunks
  const long long per_share = static_cast<long long>(t.ncb) * t.hidp * kHeadCB * 4;
  long long ns = (4ll << 20) / per_share;
  const long long chunks = static_cast<long long>(B) * ((HW + 4 * t.vec - 1) / (4 * t.vec));
  ns = std::min(ns, (chunks + 7) / 8);
  t.nshare = static_cast<int>(std::max(32ll, std::min(256ll, ns)));
  t.gw2 = (t.vec == 4 && t.hidp <= 64) ? 1 : 0;                     // k_head_bwd_gw2 (operands through LDS): a share = every nshare-th 64-pixel chunk
  if (t.gw2) {
    const long long chunks64 = static_cast<long long>(B) * ((HW + kHeadGwPx - 1) / kHeadGwPx);
    const long long cap = std::max(1ll, std::min(512ll, (4ll << 20) / per_share));
    t.nshare = static_cast<int>(std::max(1ll, std::min(cap, chunks64 / 4)));   // (measured: 2-4 chunks per workgroup and 512-2048 shares all within 1 %; 6+ chunks, one resident round: +12 %)
  }
  return t;
}
struct HeadCtxLayout { size_t z, mean, rstd, par, part, total; };
static HeadCtxLayout head_ctx_layout(int B, int C, int H, int W, int hidden) {
  const HeadTiling t = head_tiling(B, C, H, W, hidden);
  HeadCtxLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o = align16(o + n * 4); return at; };
  L.z = take(static_cast<size_t>(B) * hidden * H * W);
  L.mean = take(t.hidp); L.rstd = take(t.hidp);
  L.par = take(static_cast<size_t>(t.hidp) * kHeadPar);
  L.part = take(static_cast<size_t>(t.nwg) * 2 * t.hidp);
  L.total = o;
  return L;
}
struct HeadScratchLayout { size_t ga, part1, kst, gwpart, total; };
static HeadScratchLayout head_scratch_layout(int B, int C, int H, int W, int hidden) {
  const HeadTiling t = head_tiling(B, C, H, W, hidden);
  HeadScratchLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o = align16(o + n * 4); return at; };
  L.ga = take(static_cast<size_t>(B) * hidden * H * W);
  L.part1 = take(static_cast<size_t>(t.nwg1) * t.hidp * kHeadNStat);
  L.kst = take(5 * static_cast<size_t>(t.hidp));
  L.gwpart = take(static_cast<size_t>(t.ncb) * t.nshare * t.hidp * kHeadCB);
  L.total = o;
  return L;
}
extern "C" size_t mgahead_ctx_bytes(int B, int C, int H, int W, int hidden) {
  if (head_check_shape(B, C, H, W, hidden)) return 0;
  return head_ctx_layout(B, C, H, W, hidden).total;
}
extern "C" size_t mgahead_bwd_scratch_bytes(int B, int C, int H, int W, int hidden) {
  if (head_check_shape(B, C, H, W, hidden)) return 0;
  return head_scratch_layout(B, C, H, W, hidden).total;
}
static int head_common(const mgahead_params_t& P, int B, int C, int H, int W, int dtype, void* ctx, HeadArgs& A, Sig& sig) {
  if (!P.w1 || !P.bn_weight || !P.bn_bias || !P.running_mean || !P.running_var || !P.wh || !P.bh)
    return fail(MGACBAM_E_NULL, "mask head: NULL parameter pointer");
  if (int e = head_check_shape(B, C, H, W, P.hidden)) return e;
  if (dtype < MGACBAM_F32 || dtype > MGACBAM_BF16) return fail(MGACBAM_E_DTYPE, "mask head: dtype %d", dtype);
  if (!(P.eps > 0.f) || !(P.momentum >= 0.f && P.momentum <= 1.f)) return fail(MGACBAM_E_SHAPE, "mask head: eps=%g momentum=%g", P.eps, P.momentum);
  const HeadTiling t = head_tiling(B, C, H, W, P.hidden);
  memset(&A, 0, sizeof(A));
  A.p = HeadPtrs{P.w1, P.bn_weight, P.bn_bias, P.running_mean, P.running_var, reinterpret_cast<long long*>(P.num_batches_tracked), P.wh, P.bh};
  A.g.B = B; A.g.C = C; A.g.hid = P.hidden; A.g.H = H; A.g.W = W; A.g.HW = H * W; A.g.hidp = t.hidp; A.g.cp = t.cp;
  A.g.eps = P.eps; A.g.momentum = P.momentum; A.g.training = P.training ? 1 : 0;
  const HeadCtxLayout L = head_ctx_layout(B, C, H, W, P.hidden);
  char* cp = static_cast<char*>(ctx);
  A.c = HeadCtx{reinterpret_cast<float*>(cp + L.z), reinterpret_cast<float*>(cp + L.mean), reinterpret_cast<float*>(cp + L.rstd),
                reinterpret_cast<float*>(cp + L.par), reinterpret_cast<float*>(cp + L.part)};
  A.tile_px = t.tile_px; A.tiles_per_sample = t.tps; A.nwg = t.nwg;
  A.gx_tile_px = t.gx_tile_px; A.gx_tiles_per_sample = t.gx_tps; A.fw_kw = t.fw_kw; A.gx_kw = t.gx_kw; A.fw_mtw = t.fw_mtw; A.gx_mtw = t.gx_mtw;
  A.trace = knobs().trace; A.trace_base = 0;
  A.nwg_out = t.nwg_out; A.nwg1 = t.nwg1; A.act_ppt = t.act_ppt; A.act_hl_max = t.act_hl; A.ncb = t.ncb; A.nshare = t.nshare; A.gw2 = t.gw2;
  sig = Sig{dtype, t.vec, 0, 0, 0, 0};
  return 0;
}
static size_t head_gemm_smem(const HeadArgs* lv, int n) {
  size_t m = 0;
  for (int l = 0; l < n; ++l) m = std::max(m, (static_cast<size_t>(kHeadLdsX) + 8 * lv[l].g.hidp) * sizeof(float));
  return m;
}
template <typename Fn>
static int head_fill(Group<HeadArgs>& G, const HeadArgs* lv, int n, Fn blocks_of) {
  int tot = 0;
  for (int l = 0; l < n; ++l) { G.start[l] = tot; tot += blocks_of(lv[l]); }
  G.start[n] = tot;
  return tot;
}
static int head_forward_group(HeadArgs* lv, int n, const Sig& sig, hipStream_t st) {
  Group<HeadArgs> G;
  G.n = n;
  for (int l = 0; l < n; ++l) G.lv[l] = lv[l];
  {
    // one launch per accumulator template (two tiles per wave: hidden <= 128, i.e. every level of the n/s models; four beyond): each level
    // alone is latency-bound, so levels that share a launch overlap each other
    const size_t smem = head_gemm_smem(lv, n);
    for (int pass = 0; pass < 2; ++pass) {
      Group<HeadArgs> Gm;
      Gm.n = 0;
      int grid = 0, mtw = 1;
      for (int l = 0; l < n; ++l)
        if ((lv[l].fw_mtw <= 2) == (pass == 0)) {
          Gm.lv[Gm.n] = lv[l]; Gm.start[Gm.n] = grid; grid += lv[l].nwg; ++Gm.n;
          mtw = std::max(mtw, lv[l].fw_mtw);
        }
      if (!Gm.n) continue;
      Gm.start[Gm.n] = grid;
      for (int l = 0; l < Gm.n; ++l) Gm.lv[l].trace_base = pass * 8192;
      for (int l = 0; l < Gm.n; ++l) {                            // the wave arrangement follows the template the level runs under
        int pw;
        head_waves(Gm.lv[l].g.hidp / 16, mtw, Gm.lv[l].g.C, pw, Gm.lv[l].fw_kw);
        if (pw * 16 * sig.vec != Gm.lv[l].tile_px) return fail(MGACBAM_E_SHAPE, "mask head: inconsistent tiling");   // (cannot happen: see head_tiling)
      }
#define CALL_HP3(Tt, Vv, Mm) LAUNCH((k_head_gemm<Tt, Vv, false, Mm>), grid, smem, st, Gm)
#define CALL_HP(Tt, Vv) { if (mtw <= 2) { CALL_HP3(Tt, Vv, 2); } else { CALL_HP3(Tt, Vv, 4); } }
      if (sig.dtype == MGACBAM_F32) { if (sig.vec == 4) { CALL_HP(float, 4); } else { CALL_HP(float, 1); } }
      else if (sig.dtype == MGACBAM_F16) { if (sig.vec == 4) { CALL_HP(__half, 4); } else { CALL_HP(__half, 1); } }
      else { if (sig.vec == 4) { CALL_HP(bf16_t, 4); } else { CALL_HP(bf16_t, 1); } }
#undef CALL_HP
#undef CALL_HP3
      if (int e = launch_status("k_head_gemm<fwd>")) return e;
    }
  }
  {
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.g.hid; });
    LAUNCH(k_head_stats, grid, 0, st, G);
    if (int e = launch_status("k_head_stats")) return e;
  }
  {
    int ohl = 0;                                                  // LDS floats per wave: the level's rows and constants
    for (int l = 0; l < n; ++l) {
      int opx, ojo;
      head_out_shape(lv[l].g.hid, opx, ojo);
      ohl = std::max(ohl, ojo * (head_out_row(opx, lv[l].g.W) + kHeadOutCst));
    }
    for (int l = 0; l < n; ++l) { lv[l].out_hl_max = ohl; G.lv[l].out_hl_max = ohl; }
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.nwg_out; });
    const size_t smem = std::max(static_cast<size_t>(4) * ohl, static_cast<size_t>(16) * kWave) * sizeof(float);
#define CALL_HO(Tt) { if (sig.vec == 4) { LAUNCH((k_head_out<Tt, 4>), grid, smem, st, G); } else { LAUNCH((k_head_out<Tt, 1>), grid, smem, st, G); } }
    switch (sig.dtype) {
      case MGACBAM_F32: CALL_HO(float); break;
      case MGACBAM_F16: CALL_HO(__half); break;
      default: CALL_HO(bf16_t); break;
    }
#undef CALL_HO
    if (int e = launch_status("k_head_out")) return e;
  }
  return 0;
}
extern "C" int mgahead_forward(const mgahead_fwd_level_t* levels, int n_levels, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  HeadArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    const mgahead_fwd_level_t& L = levels[l];
    if (!L.x || !L.logits || !L.ctx) return fail(MGACBAM_E_NULL, "mask head forward: x / logits / ctx is NULL");
    if (int e = head_common(L.p, L.B, L.C, L.H, L.W, L.dtype, L.ctx, args[l], sigs[l])) return e;
    const size_t need = sigs[l].vec * elem_size(L.dtype);
    if (!aligned_to(L.x, need) || !aligned_to(L.ctx, 16)) return fail(MGACBAM_E_ALIGN, "mask head forward: x must be %zu-byte aligned, ctx 16-byte", need);
    args[l].x = L.x; args[l].logits = L.logits;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int e = for_each_group(args, sigs, n_levels, [&](HeadArgs* g, int m, const Sig& s) { return head_forward_group(g, m, s, st); })) return e;
  g_err[0] = 0;
  return 0;
}
static int head_backward_group(HeadArgs* lv, int n, const Sig& sig, hipStream_t st) {
  Group<HeadArgs> G;
  G.n = n;
  for (int l = 0; l < n; ++l) G.lv[l] = lv[l];
  {
    int hl = 0;
    for (int l = 0; l < n; ++l) hl = std::max(hl, lv[l].act_hl_max);
    for (int l = 0; l < n; ++l) { lv[l].act_hl_max = hl; G.lv[l].act_hl_max = hl; }      // the reduction scratch sits behind the launch's longest run
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.nwg1 * ((a.g.hid + kHeadJC - 1) / kHeadJC); });
    const size_t smem = (static_cast<size_t>(hl) + 16 * kHeadJC * kHeadNStat) * sizeof(float);
    switch (sig.dtype) {
      case MGACBAM_F32: LAUNCH(k_head_bwd_act<float>, grid, smem, st, G); break;
      case MGACBAM_F16: LAUNCH(k_head_bwd_act<__half>, grid, smem, st, G); break;
      default: LAUNCH(k_head_bwd_act<bf16_t>, grid, smem, st, G); break;
    }
    if (int e = launch_status("k_head_bwd_act")) return e;
  }
  {
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.g.hid; });
    LAUNCH(k_head_bwd_fin, grid, 0, st, G);
    if (int e = launch_status("k_head_bwd_fin")) return e;
  }
  {
    for (int l = 0; l < n; ++l) G.lv[l].trace_base = 16384;
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return a.g.B * a.gx_tiles_per_sample; });
    const size_t smem = head_gemm_smem(lv, n);
#define CALL_HX(Tt, Vv) LAUNCH((k_head_gemm<Tt, Vv, true, 2>), grid, smem, st, G)
    if (sig.dtype == MGACBAM_F32) { if (sig.vec == 4) { CALL_HX(float, 4); } else { CALL_HX(float, 1); } }
    else if (sig.dtype == MGACBAM_F16) { if (sig.vec == 4) { CALL_HX(__half, 4); } else { CALL_HX(__half, 1); } }
    else { if (sig.vec == 4) { CALL_HX(bf16_t, 4); } else { CALL_HX(bf16_t, 1); } }
#undef CALL_HX
    if (int e = launch_status("k_head_gemm<gx>")) return e;
  }
  {                                                               // dW1 partials
    for (int pass = 0; pass < 2; ++pass) {                         // pass 0: the levels that take the LDS-staged form, pass 1: the rest
      Group<HeadArgs> Gw;
      Gw.n = 0;
      int grid = 0;
      size_t smem = 0;
      for (int l = 0; l < n; ++l)
        if ((lv[l].gw2 != 0) == (pass == 0)) {
          Gw.lv[Gw.n] = G.lv[l]; Gw.start[Gw.n] = grid; grid += lv[l].ncb * lv[l].nshare; ++Gw.n;
          const size_t hp = static_cast<size_t>(lv[l].g.hidp);
          smem = std::max(smem, (pass == 0 ? 5 * hp + (kHeadCB + hp) * kHeadGwPitch : 5 * hp + 1024) * sizeof(float));
        }
      if (!Gw.n) continue;
      Gw.start[Gw.n] = grid;
      if (pass == 0) {
        switch (sig.dtype) {
          case MGACBAM_F32: LAUNCH(k_head_bwd_gw2<float>, grid, smem, st, Gw); break;
          case MGACBAM_F16: LAUNCH(k_head_bwd_gw2<__half>, grid, smem, st, Gw); break;
          default: LAUNCH(k_head_bwd_gw2<bf16_t>, grid, smem, st, Gw); break;
        }
        if (int e = launch_status("k_head_bwd_gw2")) return e;
        continue;
      }
#define CALL_HW(Tt, Vv) LAUNCH((k_head_bwd_gw<Tt, Vv>), grid, smem, st, Gw)
      if (sig.dtype == MGACBAM_F32) { if (sig.vec == 4) { CALL_HW(float, 4); } else { CALL_HW(float, 1); } }
      else if (sig.dtype == MGACBAM_F16) { if (sig.vec == 4) { CALL_HW(__half, 4); } else { CALL_HW(__half, 1); } }
      else { if (sig.vec == 4) { CALL_HW(bf16_t, 4); } else { CALL_HW(bf16_t, 1); } }
#undef CALL_HW
      if (int e = launch_status("k_head_bwd_gw")) return e;
    }
  }
  {
    const int grid = head_fill(G, lv, n, [](const HeadArgs& a) { return (a.g.hid * a.g.C + kHeadGwfOut - 1) / kHeadGwfOut; });
    LAUNCH(k_head_bwd_gwf, grid, 0, st, G);
    if (int e = launch_status("k_head_bwd_gwf")) return e;
  }
  return 0;
}
extern "C" int mgahead_backward(const mgahead_bwd_level_t* levels, int n_levels, void* stream) {
  if (!levels) return fail(MGACBAM_E_NULL, "levels is NULL");
  if (n_levels < 1 || n_levels > MGACBAM_MAX_LEVELS) return fail(MGACBAM_E_LEVELS, "n_levels=%d", n_levels);
  HeadArgs args[MGACBAM_MAX_LEVELS];
  Sig sigs[MGACBAM_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    const mgahead_bwd_level_t& L = levels[l];
    if (!L.x || !L.g_logits || !L.ctx || !L.scratch || !L.gx) return fail(MGACBAM_E_NULL, "mask head backward: x / g_logits / ctx / scratch / gx is NULL");
    if (!L.gw1 || !L.gbn_weight || !L.gbn_bias || !L.gwh || !L.gbh) return fail(MGACBAM_E_NULL, "mask head backward: NULL parameter-gradient pointer");
    if (int e = head_common(L.p, L.B, L.C, L.H, L.W, L.dtype, const_cast<void*>(L.ctx), args[l], sigs[l])) return e;
    const size_t need = sigs[l].vec * elem_size(L.dtype);
    if (!aligned_to(L.x, need) || !aligned_to(L.gx, need) || !aligned_to(L.ctx, 16) || !aligned_to(L.scratch, 16))
      return fail(MGACBAM_E_ALIGN, "mask head backward: x/gx must be %zu-byte aligned, ctx/scratch 16-byte", need);
    HeadArgs& A = args[l];
    A.x = L.x; A.gl = L.g_logits; A.gx = L.gx;
    A.gw1 = L.gw1; A.ggamma = L.gbn_weight; A.gbeta = L.gbn_bias; A.gwh = L.gwh; A.gbh = L.gbh;
    A.accum_gx = (L.flags & MGAHEAD_BWD_ACCUM_GX) ? 1 : 0;
    A.gl2 = L.g_logits2;
    const HeadScratchLayout SL = head_scratch_layout(L.B, L.C, L.H, L.W, L.p.hidden);
    char* sp = static_cast<char*>(L.scratch);
    A.s = HeadScratch{reinterpret_cast<float*>(sp + SL.ga), reinterpret_cast<float*>(sp + SL.part1), reinterpret_cast<float*>(sp + SL.kst),
                      reinterpret_cast<float*>(sp + SL.gwpart)};
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int e = for_each_group(args, sigs, n_levels, [&](HeadArgs* g, int m, const Sig& s) { return head_backward_group(g, m, s, st); })) return e;
  g_err[0] = 0;
  return 0;
}

extern "C" int mgakendall_forward(const float* det, int n_det, const float* seg, const float* log_vars, float* total, void* stream) {
  if (!det || !seg || !log_vars || !total) return fail(MGACBAM_E_NULL, "kendall: NULL pointer");
  if (n_det < 1 || n_det > 4096) return fail(MGACBAM_E_SHAPE, "kendall: n_det=%d", n_det);
  KendallArgs A{det, seg, log_vars, nullptr, total, nullptr, nullptr, nullptr, n_det};
  void* p[] = {&A};
  g_launch_err = hipLaunchKernel(reinterpret_cast<const void*>(k_kendall_fwd), dim3(1), dim3(kWave), p, 0, static_cast<hipStream_t>(stream));
  if (int e = launch_status("k_kendall_fwd")) return e;
  g_err[0] = 0;
  return 0;
}
extern "C" int mgakendall_backward(const float* det, int n_det, const float* seg, const float* log_vars, const float* g_total,
                                   float* g_det, float* g_seg, float* g_log_vars, void* stream) {
  if (!det || !seg || !log_vars || !g_total || !g_det || !g_seg || !g_log_vars) return fail(MGACBAM_E_NULL, "kendall: NULL pointer");
  if (n_det < 1 || n_det > 4096) return fail(MGACBAM_E_SHAPE, "kendall: n_det=%d", n_det);
  KendallArgs A{det, seg, log_vars, g_total, nullptr, g_det, g_seg, g_log_vars, n_det};
  void* p[] = {&A};
  g_launch_err = hipLaunchKernel(reinterpret_cast<const void*>(k_kendall_bwd), dim3(1), dim3(kWave), p, 0, static_cast<hipStream_t>(stream));
  if (int e = launch_status("k_kendall_bwd")) return e;
  g_err[0] = 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// ProbMaskGater
// ------------------------------------------------------------------------------------------------
static int pmg_args(size_t n, const mgapmg_cfg_t* cfg, GaterArgs& A) {
  if (!cfg) return fail(MGACBAM_E_NULL, "gater: cfg is NULL");
  if (n < 1 || !(cfg->tau > 0.f)) return fail(MGACBAM_E_SHAPE, "gater: n=%zu tau=%g", n, cfg->tau);
  A.n = n; A.inv_tau = 1.f / cfg->tau; A.p_min = cfg->p_min; A.threshold = cfg->threshold; A.hard = cfg->hard ? 1 : 0;
  A.p = A.u1 = A.u2 = A.gout = nullptr; A.out = A.msoft = A.gp = nullptr;
  return 0;
}
static unsigned pmg_grid(size_t n) { const size_t g = (n + kBlock - 1) / kBlock; return static_cast<unsigned>(g > 2048 ? 2048 : g); }
extern "C" int mgapmg_forward(const float* p, const float* u1, const float* u2, float* out, float* msoft, size_t n,
                              const mgapmg_cfg_t* cfg, void* stream) {
  if (!p || !u1 || !u2 || !out || !msoft) return fail(MGACBAM_E_NULL, "gater: NULL pointer");
  GaterArgs A;
  if (int e = pmg_args(n, cfg, A)) return e;
  A.p = p; A.u1 = u1; A.u2 = u2; A.out = out; A.msoft = msoft;
  LAUNCH(k_pmg_fwd, pmg_grid(n), 0, static_cast<hipStream_t>(stream), A);
  if (int e = launch_status("k_pmg_fwd")) return e;
  g_err[0] = 0;
  return 0;
}
extern "C" int mgapmg_backward(const float* p, const float* msoft, const float* gout, float* gp, size_t n, const mgapmg_cfg_t* cfg,
                               void* stream) {
  if (!p || !msoft || !gout || !gp) return fail(MGACBAM_E_NULL, "gater: NULL pointer");
  GaterArgs A;
  if (int e = pmg_args(n, cfg, A)) return e;
  A.p = p; A.msoft = const_cast<float*>(msoft); A.gout = gout; A.gp = gp;
  LAUNCH(k_pmg_bwd, pmg_grid(n), 0, static_cast<hipStream_t>(stream), A);
  if (int e = launch_status("k_pmg_bwd")) return e;
  g_err[0] = 0;
  return 0;
}
