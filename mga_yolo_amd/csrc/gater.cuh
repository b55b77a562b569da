// ProbMaskGater on the device: the stochastic gate the reference applies to the mask before the block when
// MGA_PROB_MODE is set.                                                     mga_yolo/nn/modules/probmaskgater.py:58-98
//   p = max(clamp(p_in, 0, 1), p_min)                                        :77-79
//   gumbel : m = sigmoid((logit(clamp(p, 1e-6, 1-1e-6)) + g) / tau),  g = -log(-log U1) + log(-log U2)      :58-71, 85-87
//   hard_st: forward (m > threshold), backward that of m (straight-through)                                   :89-92
// The two uniform tensors are drawn by the caller with torch's generator exactly as the reference does (:53-56), so the kernel is
// a pure function of (p_in, U1, U2) and parity with the reference is element-wise, not statistical.  One launch instead of ~14.
#pragma once
#include "common.cuh"

namespace mgacbam {

constexpr float kGateEps = 1e-6f;
struct GaterArgs {
  const float* p; const float* u1; const float* u2;   // inputs
  float* out; float* msoft;                            // forward outputs (msoft: saved for backward)
  const float* gout; float* gp;                        // backward
  size_t n;
  float inv_tau, p_min, threshold;
  int hard;
};

__global__ __launch_bounds__(kBlock) void k_pmg_fwd(const GaterArgs A) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < A.n; i += static_cast<size_t>(gridDim.x) * kBlock) {
    float p = fminf(fmaxf(A.p[i], 0.f), 1.f);
    if (A.p_min > 0.f) p = fmaxf(p, A.p_min);
    const float a = fminf(fmaxf(A.u1[i], kGateEps), 1.f - kGateEps), b = fminf(fmaxf(A.u2[i], kGateEps), 1.f - kGateEps);
    const float g = -logf(-logf(a)) + logf(-logf(b));                       // logistic noise
    const float q = fminf(fmaxf(p, kGateEps), 1.f - kGateEps);
    const float z = (logf(q) - log1pf(-q) + g) * A.inv_tau;
    const float m = 1.f / (1.f + expf(-z));
    A.msoft[i] = m;
    A.out[i] = A.hard ? (m > A.threshold ? 1.f : 0.f) : m;
  }
}

__global__ __launch_bounds__(kBlock) void k_pmg_bwd(const GaterArgs A) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < A.n; i += static_cast<size_t>(gridDim.x) * kBlock) {
    const float pin = A.p[i];
    float p = fminf(fmaxf(pin, 0.f), 1.f);
    bool pass = pin >= 0.f && pin <= 1.f;                                   // clamp(0,1) passes gradient inside, bounds included
    if (A.p_min > 0.f) { pass = pass && p >= A.p_min; p = fmaxf(p, A.p_min); }
    pass = pass && p >= kGateEps && p <= 1.f - kGateEps;                    // the logit's own clamp
    const float m = A.msoft[i];
    const float dlogit = 1.f / p + 1.f / (1.f - p);
    A.gp[i] = pass ? A.gout[i] * m * (1.f - m) * A.inv_tau * dlogit : 0.f;
  }
}

}  // namespace mgacbam
