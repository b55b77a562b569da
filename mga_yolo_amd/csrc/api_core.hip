// libmgacbam.so, C ABI (include/mgacbam.h): version / error / knob entry points and the size queries of the MaskCBAM family
#include "host.cuh"

extern "C" int mgacbam_abi_version(void) { return MGACBAM_ABI_VERSION; }

extern "C" const char* mgacbam_last_error(void) { return g_err; }

extern "C" const char* mgacbam_build_info(void) {
  return "libmgacbam gfx950 (CDNA4) hip " __VERSION__ " built " __DATE__;
}

extern "C" void mgacbam_reload_env(void) {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  g_knobs = read_knobs();
  g_knobs_ready.store(true, std::memory_order_release);
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
  size_t m = 0;                                                  // one answer for every element type the backward may be called with
  for (int dt = MGACBAM_F32; dt <= MGACBAM_BF16; ++dt) m = std::max(m, scratch_layout(B, C, H, W, hidden, k, dt).total);
  return m;
}

