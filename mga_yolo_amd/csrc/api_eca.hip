// libmgacbam.so, C ABI (include/mgacbam.h): MaskECA, the nearest resize and the ProbMaskGater gate
#include "host.cuh"
#include "eca.cuh"
#include "resize.cuh"
#include "gater.cuh"

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
    if (int e = check_capacity("eca forward", "ctx", eca_ctx_layout(L.B, L.C, L.H, L.W).total, L.ctx_bytes)) return e;
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
    if (int e = check_capacity("eca backward", "ctx", eca_ctx_layout(L.B, L.C, L.H, L.W).total, L.ctx_bytes)) return e;
    if (int e = check_capacity("eca backward", "scratch", align16(static_cast<size_t>(L.B) * L.C * 4), L.scratch_bytes)) return e;
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

