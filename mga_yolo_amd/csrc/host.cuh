// Host side shared by every translation unit of libmgacbam.so (api_*.hip): error reporting, launch helper, knobs, ctx / scratch
// layouts, launch geometry and the level-grouping of the grouped launches.  The library is built from several translation units
// (one per kernel family) so that they compile in parallel and an edit rebuilds one family; the state they share (the thread-local
// error string, the knobs) is C++17 inline data: ONE instance per process.
// No allocation, no host<->device copy, no synchronisation anywhere: every entry point only enqueues kernels on the caller's
// stream, so calls are re-entrant and graph-capturable.
#pragma once
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
#include "common.cuh"

using namespace mgacbam;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
inline thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
// Every launch goes through hipLaunchKernel and its OWN return value is checked: the process-wide sticky error state (which
// may hold an asynchronous error of the framework's kernels on this thread) is neither read nor cleared.
inline thread_local hipError_t g_launch_err = hipSuccess;
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
  // hand-off state: [B][nflag] k_gate tile flags, 4 status words, [B] ca flags, [B][nflag] k_bwd_reduce1 tile flags, [B][nflag] folded
  // conv-tile flags; merged backward launch (k_bwd_r12): [B][nflag] tile, [B][nflag] conv-tile, [B][nflag] dWsa-tile and [B][C] sweep flags
  // (the merged launch has tile / conv-tile flags of its OWN beside its dWsa-tile and sweep flags: every class of generation counters is
  //  bumped exactly once per launch of its kind, so the two launch forms can alternate on one ctx without their counters drifting apart)
  L->sync = take(6 * static_cast<size_t>(B) * sync_flags(HW) + 4 + B + static_cast<size_t>(B) * C);
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
  int head_gw_div;         // MGAHEAD_GW_DIV (default 5): 64-pixel chunks per k_head_bwd_gw2 workgroup (pixel shares = chunks / this).  5: the YOLOv8n pyramid's 1,008
                           // workgroups are one resident round at the kernel's 108 VGPRs (4 per CU): slice 0.3886 -> 0.3852 ms; 3, 6, 8 and config 3: no difference
  int bwd_merge;           // MGACBAM_BWD_MERGE (default 1): k_bwd_reduce1 + conv + dWsa tiles + k_bwd_reduce2 as one launch (k_bwd_r12)
  int wsa_tail;            // MGACBAM_WSA_TAIL (default 0, opt-in): dWsa tile partials + sums as the last workgroups of the k_bwd_apply launch
  int gate_narrow;         // MGACBAM_GATE_NARROW (default 0): k_gate also for tiles narrower than an image row
  int gate, chan_mintx, pool_tx, pool_cpt, r2_cpt, wsa_fat, chan_tx, chanf_tx, split_mlp, nt, half_vec, gate_h8, level_order, bwd_fold;
  int resident_wgs;        // MGACBAM_RESIDENT_WGS: override of the co-resident workgroup budget the hand-off eligibility is sized from
  int fault;               // MGACBAM_FAULT: fault injection for tests (args.cuh)
  unsigned spin_limit;     // MGACBAM_SPIN_LIMIT
  long long* trace;        // MGACBAM_TRACE_PTR (-DMGACBAM_TRACE builds, tools/trace_gate.py)
};
static Knobs read_knobs() {
  Knobs k;
  k.bwd_merge = env_int("MGACBAM_BWD_MERGE", 1);
  k.head_gw_div = env_int("MGAHEAD_GW_DIV", 5);
  k.wsa_tail = env_int("MGACBAM_WSA_TAIL", 0);
  k.gate_narrow = env_int("MGACBAM_GATE_NARROW", 0);
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
inline std::mutex g_knob_mu;
inline Knobs g_knobs;
inline std::atomic<bool> g_knobs_ready{false};
static Knobs knobs() {
  if (!g_knobs_ready.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    if (!g_knobs_ready.load(std::memory_order_relaxed)) { g_knobs = read_knobs(); g_knobs_ready.store(true, std::memory_order_release); }
  }
  return g_knobs;   // (written once under the lock before the flag; mgacbam_reload_env is documented as not concurrent with calls)
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
  // TP < W (wide feature maps at C >= 256: a tile is a fraction of an image row).  Supported -- a tile inside one row stages and waits
  // for only the columns its windows reach (fwd.cuh: `narrow`), parity-tested -- but OPT-IN (MGACBAM_GATE_NARROW=1): at BASELINE
  // configs[3] it measures 224 us against 201 us for k_mlp + k_chan + k_apply.  tools/trace_gate.py fwd cfg4: a 64- / 32-pixel tile
  // needs rows of up to 3 below it, i.e. tiles 7-8 positions AHEAD in dispatch order, and sits 6.8 us (p90 11) waiting for them to be
  // scheduled and to reach their publish point; the TY = 16 / 32 channel slices combine through LDS in 5.2 us; the C = 512 role MLP
  // takes 20 us for the first round; 60 % of the resident workgroups are in the chain at any time and HBM runs at 3 TB/s
  // (whether the 8*span + 1 workgroups a tile's wait spans are co-resident on THIS device is checked at dispatch: forward_group)
  if (TP >= kSyncPx && (TP >= W || knobs().gate_narrow) && lds <= 48 * 1024) { t.gate_tx = gtx; t.gate_rows = grows; t.gate_span = span; }
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
static ScratchLayout scratch_layout(int B, int C, int H, int W, int hidden, int k, int dtype) {
  // the tile counts follow the launch geometry, which follows the element type (vector width) and the knobs: the layout is per dtype
  const Tune t = choose_tune(B, C, H, W, k, dtype);
  const size_t HW = static_cast<size_t>(H) * W, BC = static_cast<size_t>(B) * C;
  const size_t nt = chan_tiles(t, H, W, vec_of(H, W, dtype)), nconv = static_cast<size_t>(B) * wsa_tiles(t, H, W);
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

// ABI 14: every work buffer travels with its capacity; the requirement is recomputed under the CURRENT knobs at every call
static int check_capacity(const char* what, const char* buf, size_t want, size_t got) {
  if (got < want)
    return fail(MGACBAM_E_SIZE, "%s: %s holds %zu bytes, this shape needs %zu under the current knobs (query the size again after "
                "mgacbam_reload_env(); a size cached across a knob change or taken for another shape is stale)", what, buf, got, want);
  return 0;
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
  int lf32 = 0;   // mask head only: logits / g_logits are fp32 whatever dtype is (MGAHEAD_LOGITS_F32)
  int gvec = 0;   // forward only: elements per lane of k_gate for this level -- a function of the LEVEL alone (dtype, shape, k, knobs), never of
                  // the levels it happens to be called with: the hand-off flags in ctx.sync count calls per TILE, so a ctx must see the same
                  // tiling in every call whatever the group composition (levels of different gvec go to different launches)
  bool operator==(const Sig& o) const {
    return dtype == o.dtype && vec == o.vec && has_mask == o.has_mask && k == o.k && gmask == o.gmask && proj == o.proj && gvec == o.gvec && lf32 == o.lf32;
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
