// libmgacbam.so, C ABI (include/mgacbam.h): segmentation loss and the Kendall combine
#include "host.cuh"
#include "segloss.cuh"

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
static int seg_forward_impl(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, void* ws, size_t ws_bytes, float* out, const KendallArgs* kd, void* stream) {
  if (int e = seg_check(levels, n, cfg, false)) return e;
  if (!ws || !out) return fail(MGACBAM_E_NULL, "segloss: ws / out is NULL");
  if (int e = check_capacity("segloss forward", "ws", mgaseg_ws_bytes(levels, n), ws_bytes)) return e;
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
static int seg_backward_impl(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, const void* ws, size_t ws_bytes, const float* gout, const KendallArgs* kd, void* stream) {
  if (int e = seg_check(levels, n, cfg, true)) return e;
  if (!ws || (!gout && !kd)) return fail(MGACBAM_E_NULL, "segloss: ws / gout is NULL");
  if (int e = check_capacity("segloss backward", "ws", mgaseg_ws_bytes(levels, n), ws_bytes)) return e;
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
extern "C" int mgaseg_forward(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, void* ws, size_t ws_bytes, float* out, void* stream) {
  return seg_forward_impl(levels, n, cfg, ws, ws_bytes, out, nullptr, stream);
}
extern "C" int mgaseg_backward(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, const void* ws, size_t ws_bytes, const float* gout,
                               void* stream) {
  return seg_backward_impl(levels, n, cfg, ws, ws_bytes, gout, nullptr, stream);
}
static int kendall_check(const float* det, int n_det, const float* log_vars) {
  if (!det || !log_vars) return fail(MGACBAM_E_NULL, "kendall: NULL pointer");
  if (n_det < 1 || n_det > 4096) return fail(MGACBAM_E_SHAPE, "kendall: n_det=%d", n_det);
  return 0;
}
extern "C" int mgaseg_kendall_forward(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, void* ws, size_t ws_bytes, float* out,
                                      const float* det, int n_det, const float* log_vars, float* total, void* stream) {
  if (int e = kendall_check(det, n_det, log_vars)) return e;
  if (!total) return fail(MGACBAM_E_NULL, "kendall: total is NULL");
  const KendallArgs kd{det, out, log_vars, nullptr, total, nullptr, nullptr, nullptr, n_det};
  return seg_forward_impl(levels, n, cfg, ws, ws_bytes, out, &kd, stream);
}
extern "C" int mgaseg_kendall_backward(const mgaseg_level_t* levels, int n, const mgaseg_cfg_t* cfg, const void* ws, size_t ws_bytes, const float* out,
                                       const float* det, int n_det, const float* log_vars, const float* g_total,
                                       float* g_det, float* g_seg, float* g_log_vars, void* stream) {
  if (int e = kendall_check(det, n_det, log_vars)) return e;
  if (!out || !g_total || !g_det || !g_log_vars) return fail(MGACBAM_E_NULL, "kendall: NULL pointer");
  const KendallArgs kd{det, out, log_vars, g_total, nullptr, g_det, g_seg, g_log_vars, n_det};
  return seg_backward_impl(levels, n, cfg, ws, ws_bytes, nullptr, &kd, stream);
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

