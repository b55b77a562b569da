// Multi-scale segmentation loss on the mask logits: BCE-with-logits (mean) + soft Dice per level, summed with the
// scale weights.                                                         mga_yolo/nn/losses/segmentation.py:38-42, 87-151
//   forward : k_seg_partial  (sample b, part) -> partial sums {bce, p*t, p, t}   ;  k_seg_final: per level bce = sum/(B*HW),
//             dice_b = 1 - (2 I_b + s)/(P_b + T_b + s), combined = w_l (w_bce bce + w_dice mean_b dice_b), total = lambda * sum_l
//   backward: k_seg_bwd      g_x = g_total * lambda * w_l * [ w_bce (p - t)/(B*HW) + w_dice/B * d dice_b/d p * p(1-p) ]
//             d dice_b / d p_i = -(2 t_i D_b - (2 I_b + s)) / D_b^2,  D_b = P_b + T_b + s
// The target may live at another resolution: it is gathered with F.interpolate(mode="nearest")'s index rule on the fly
// (segmentation.py:103-110; same integer path as k_resize_nearest).  All tensors are B x 1 x H x W: the whole loss is a few MB, so
// the kernels are launch/latency-bound; one launch covers every level, sums are two-stage with a fixed order (reproducible).
#pragma once
#include "common.cuh"

namespace mgacbam {

constexpr int kSegMaxLevels = 4;
constexpr int kSegParts = 8;       // workgroups per sample and level
constexpr float kSegEps = 1e-6f;   // eps of _lmf / _lmft (segmentation.py:44, 65)

struct SegLevel {
  const void* logits;   // (B,1,H,W) T
  const float* target;  // (B,1,Ht,Wt) fp32
  void* glogits;        // (B,1,H,W) T, backward only
  float* part;          // (B, kSegParts, 4) partial sums            (workspace)
  float* sums;          // (B, 4) per-sample {bce, I, P, T}           (workspace, saved for backward)
  int B, H, W, Ht, Wt;
  float w_scale;
  int bilinear;         // target resize rule when (Ht,Wt) != (H,W): 0 nearest (segmentation.py:110), 1 bilinear align_corners=False (:103-108)
};
// Kendall multi-task combine (mga_yolo/model/model.py:204-206): total[i] = e^{-s0} det[i] + s0 + e^{-s1} seg + s1
struct KendallArgs {
  const float* det; const float* seg; const float* log_vars; const float* g_total;
  float* total; float* g_det; float* g_seg; float* g_log_vars;
  int n;
};
struct SegArgs {
  int n;
  int start[kSegMaxLevels + 1];     // workgroup ids of level l: [start[l], start[l+1])
  SegLevel lv[kSegMaxLevels];
  float w_bce, w_dice, smooth, lambda;
  int ufl;                          // 1: Unified Focal mode (segmentation.py:44-85, 114-131): slot 0 of the sums = modified focal CE
  float u_lambda, u_delta, u_gamma;
  float* out;                       // [0] total, then per level {bce, dice, combined}
  const float* gout;                // backward: dL/d(total), device scalar (has_kd: unused, dL/d(seg total) comes from the combine)
  int has_kd;                       // mgaseg_kendall_*: the Kendall combine rides in k_seg_final / k_seg_bwd (two launches fewer per step)
  KendallArgs kd;
};

__device__ __forceinline__ float seg_target(const SegLevel& L, int b, int y, int x) {
  if (L.Ht == L.H && L.Wt == L.W) return L.target[(static_cast<size_t>(b) * L.H + y) * L.W + x];
  const float sh = static_cast<float>(L.Ht) / static_cast<float>(L.H), sw = static_cast<float>(L.Wt) / static_cast<float>(L.W);
  if (L.bilinear) {
    // F.interpolate(mode="bilinear", align_corners=False): src = max(scale * (dst + 0.5) - 0.5, 0), 4 taps, upper index clamped
    const float fy = fmaxf(sh * (static_cast<float>(y) + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * (static_cast<float>(x) + 0.5f) - 0.5f, 0.f);
    const int y0 = min(static_cast<int>(fy), L.Ht - 1), x0 = min(static_cast<int>(fx), L.Wt - 1);
    const int y1 = y0 + (y0 < L.Ht - 1 ? 1 : 0), x1 = x0 + (x0 < L.Wt - 1 ? 1 : 0);
    const float ly = fy - static_cast<float>(y0), lx = fx - static_cast<float>(x0);
    const float* tb = L.target + static_cast<size_t>(b) * L.Ht * L.Wt;
    const float top = (1.f - lx) * tb[static_cast<size_t>(y0) * L.Wt + x0] + lx * tb[static_cast<size_t>(y0) * L.Wt + x1];
    const float bot = (1.f - lx) * tb[static_cast<size_t>(y1) * L.Wt + x0] + lx * tb[static_cast<size_t>(y1) * L.Wt + x1];
    return (1.f - ly) * top + ly * bot;
  }
  const int sy = min(static_cast<int>(floorf(static_cast<float>(y) * sh)), L.Ht - 1);
  const int sx = min(static_cast<int>(floorf(static_cast<float>(x) * sw)), L.Wt - 1);
  return L.target[(static_cast<size_t>(b) * L.Ht + sy) * L.Wt + sx];
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_seg_partial(const SegArgs A) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < kSegMaxLevels; ++i)
    if (i < A.n && static_cast<int>(blockIdx.x) >= A.start[i]) l = i;
  const SegLevel& L = A.lv[l];
  const int local = blockIdx.x - A.start[l];
  const int b = local / kSegParts, part = local - b * kSegParts;
  const int HW = L.H * L.W;
  const T* xp = static_cast<const T*>(L.logits) + static_cast<size_t>(b) * HW;
  float s_bce = 0.f, s_i = 0.f, s_p = 0.f, s_t = 0.f;
  // batches of 4 positions per thread: their logits and targets are requested together (one position per loop trip is a chain of
  // dependent trips to memory, and this kernel is nothing but latency)
  constexpr int U = 4;
  for (int i0 = part * kBlock + threadIdx.x; i0 < HW; i0 += U * kSegParts * kBlock) {
    float xv[U], tv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * kSegParts * kBlock;
      xv[u] = 0.f; tv[u] = 0.f;
      if (i < HW) {
        xv[u] = to_f32<T>(xp[i]);
        const int y = i / L.W;
        tv[u] = seg_target(L, b, y, i - y * L.W);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * kSegParts * kBlock < HW) {
        const float x = xv[u], t = tv[u];
        const float e = expf(-fabsf(x));
        const float ce = fmaxf(x, 0.f) - x * t + log1pf(e);      // binary_cross_entropy_with_logits, elementwise
        const float p = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);   // sigmoid
        if (A.ufl) {                                             // _lmf, segmentation.py:44-52
          const bool pos = t > 0.5f;
          const float pt = fminf(fmaxf(pos ? p : 1.f - p, kSegEps), 1.f - kSegEps);
          const float base = fmaxf(1.f - pt, kSegEps);
          s_bce += powf(base, 1.f - A.u_gamma) * ce * (pos ? A.u_delta : 1.f - A.u_delta);
        } else {
          s_bce += ce;
        }
        s_i += p * t; s_p += p; s_t += t;
      }
    }
  }
  // the four sums together: wave sums (DPP), one trip through LDS, fixed order over the waves
  __shared__ float red4[kBlock / kWave][4];
  float v[4] = {s_bce, s_i, s_p, s_t};
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = wave_group_sum(v[q], kWave);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) red4[threadIdx.x >> 6][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float r = red4[0][threadIdx.x];
    for (int w = 1; w < kBlock / kWave; ++w) r += red4[w][threadIdx.x];
    L.part[(static_cast<size_t>(b) * kSegParts + part) * 4 + threadIdx.x] = r;
  }
}

// one workgroup: per-sample sums (fixed order), the level terms and the total.  Wave w owns level w (kSegMaxLevels = waves per
// workgroup): the levels' chains of dependent loads run side by side and the sums over the samples are wave reductions (no barrier);
// the operands of the Kendall epilogue are requested at the start.
__global__ __launch_bounds__(kBlock) void k_seg_final(const SegArgs A) {
  static_assert(kSegMaxLevels <= kBlock / kWave, "one wave per level");
  __shared__ float s_comb[kSegMaxLevels];
  __shared__ float s_total;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s0 = 0.f, s1 = 0.f;
  if (A.has_kd) { s0 = A.kd.log_vars[0]; s1 = A.kd.log_vars[1]; }
  if (wave < A.n) {
    const int l = wave;
    const SegLevel& L = A.lv[l];
    float bce = 0.f, dice = 0.f;
    for (int b = lane; b < L.B; b += kWave) {
      float pv[kSegParts][4];
#pragma unroll
      for (int p = 0; p < kSegParts; ++p) load_vec<float, 4>(L.part + (static_cast<size_t>(b) * kSegParts + p) * 4, pv[p]);
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < kSegParts; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += pv[p][q];
      store_vec<float, 4>(L.sums + b * 4, s);
      bce += s[0];
      if (A.ufl) {                                                             // _lmft, segmentation.py:65-76 (tp + fn = T, tp + fp = P)
        const float den = fmaxf(A.u_delta * s[3] + (1.f - A.u_delta) * s[2] + A.smooth, kSegEps);
        dice += powf(fmaxf(1.f - (s[1] + A.smooth) / den, kSegEps), A.u_gamma);
      } else {
        dice += 1.f - (2.f * s[1] + A.smooth) / (s[2] + s[3] + A.smooth);      // segmentation.py:38-42
      }
    }
    bce = wave_group_sum(bce, kWave);
    dice = wave_group_sum(dice, kWave);
    if (lane == 0) {
      bce /= static_cast<float>(L.B) * static_cast<float>(L.H * L.W);          // BCEWithLogitsLoss(reduction="mean")
      dice /= static_cast<float>(L.B);
      const float comb = A.ufl ? L.w_scale * (A.u_lambda * bce + (1.f - A.u_lambda) * dice)      // segmentation.py:120
                               : L.w_scale * (A.w_bce * bce + A.w_dice * dice);                  // segmentation.py:134-136
      A.out[1 + 3 * l] = bce; A.out[2 + 3 * l] = dice; A.out[3 + 3 * l] = comb;
      s_comb[l] = comb;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int l = 0; l < A.n; ++l) total += s_comb[l];                          // level order
    A.out[0] = total * A.lambda;                                               // segmentation.py:149
    s_total = total * A.lambda;
  }
  if (A.has_kd) {                                                              // model.py:204-206 on the value just formed
    __syncthreads();
    const float seg_term = expf(-s1) * s_total + s1, e0 = expf(-s0);
    for (int i = threadIdx.x; i < A.kd.n; i += kBlock) A.kd.total[i] = e0 * A.kd.det[i] + s0 + seg_term;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_seg_bwd(const SegArgs A) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < kSegMaxLevels; ++i)
    if (i < A.n && static_cast<int>(blockIdx.x) >= A.start[i]) l = i;
  const SegLevel& L = A.lv[l];
  const int local = blockIdx.x - A.start[l];
  const int b = local / kSegParts, part = local - b * kSegParts;
  const int HW = L.H * L.W;
  const T* xp = static_cast<const T*>(L.logits) + static_cast<size_t>(b) * HW;
  T* gp = static_cast<T*>(L.glogits) + static_cast<size_t>(b) * HW;
  float gout;
  if (A.has_kd) {                                               // dL/d(seg total) = e^{-s1} sum_i g_total[i]: every wave forms it for itself (no barrier);
    const int lane = threadIdx.x & 63;                          // workgroup 0 also leaves the combine's own gradients
    const float s0 = A.kd.log_vars[0], s1 = A.kd.log_vars[1], seg = *A.kd.seg;
    const float e0 = expf(-s0), e1 = expf(-s1);
    const bool writer = blockIdx.x == 0 && threadIdx.x < kWave;
    float gsum = 0.f, g0 = 0.f;
    for (int i = lane; i < A.kd.n; i += kWave) {
      const float gi = A.kd.g_total[i];
      if (writer) A.kd.g_det[i] = gi * e0;
      gsum += gi;
      g0 += gi * (1.f - e0 * A.kd.det[i]);
    }
    gsum = wave_group_sum(gsum, kWave);                          // (uniform: every lane holds the sum)
    g0 = wave_group_sum(g0, kWave);
    if (writer && lane == 0) {
      if (A.kd.g_seg) *A.kd.g_seg = gsum * e1;
      A.kd.g_log_vars[0] = g0;
      A.kd.g_log_vars[1] = gsum * (1.f - e1 * seg);
    }
    gout = gsum * e1;
  } else {
    gout = *A.gout;
  }
  const float g = gout * A.lambda * L.w_scale;
  const float I = L.sums[b * 4 + 1], D = L.sums[b * 4 + 2] + L.sums[b * 4 + 3] + A.smooth;
  const float kb = g * A.w_bce / (static_cast<float>(L.B) * static_cast<float>(HW));
  const float kd = g * A.w_dice / static_cast<float>(L.B);
  const float num = 2.f * I + A.smooth, invD2 = 1.f / (D * D);
  // Unified Focal: mti = (tp + s)/Du, Du = delta T + (1-delta) P + s;  M = max(1 - mti, eps)^gamma
  const float Du_raw = A.u_delta * L.sums[b * 4 + 3] + (1.f - A.u_delta) * L.sums[b * 4 + 2] + A.smooth;
  const float Du = fmaxf(Du_raw, kSegEps);
  const float om = 1.f - (I + A.smooth) / Du;
  const float dM = om > kSegEps ? -A.u_gamma * powf(om, A.u_gamma - 1.f) : 0.f;     // dM/dmti (clamp_min passes no gradient below eps)
  const float dDu = Du_raw > kSegEps ? (1.f - A.u_delta) : 0.f;
  const float ku = g * A.u_lambda / (static_cast<float>(L.B) * static_cast<float>(HW));
  const float kt = g * (1.f - A.u_lambda) / static_cast<float>(L.B);
  for (int i = part * kBlock + threadIdx.x; i < HW; i += kSegParts * kBlock) {
    const float x = to_f32<T>(xp[i]);
    const int y = i / L.W;
    const float t = seg_target(L, b, y, i - y * L.W);
    const float e = expf(-fabsf(x));
    const float p = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    const float dp = p * (1.f - p);
    if (A.ufl) {
      const bool pos = t > 0.5f;
      const float raw = pos ? p : 1.f - p;
      const float pt = fminf(fmaxf(raw, kSegEps), 1.f - kSegEps);
      const float dpt = (raw >= kSegEps && raw <= 1.f - kSegEps) ? (pos ? dp : -dp) : 0.f;
      const float om_pt = 1.f - pt;
      const float base = fmaxf(om_pt, kSegEps);
      const float dbase = om_pt >= kSegEps ? -dpt : 0.f;
      const float ce = fmaxf(x, 0.f) - x * t + log1pf(e);
      const float w = pos ? A.u_delta : 1.f - A.u_delta;
      const float a1 = 1.f - A.u_gamma;
      const float dlmf = w * (a1 * powf(base, a1 - 1.f) * dbase * ce + powf(base, a1) * (p - t));
      const float dmti = (t * Du - (I + A.smooth) * dDu) / (Du * Du);
      gp[i] = from_f32<T>(ku * dlmf + kt * dM * dmti * dp);
    } else {
      const float ddice = -(2.f * t * D - num) * invD2;
      gp[i] = from_f32<T>(kb * (p - t) + kd * ddice * dp);
    }
  }
}

// mgakendall_*: the combine alone, one wave
__global__ __launch_bounds__(kWave) void k_kendall_fwd(const KendallArgs A) {
  const float s0 = A.log_vars[0], s1 = A.log_vars[1];
  const float seg_term = expf(-s1) * *A.seg + s1;
  for (int i = threadIdx.x; i < A.n; i += kWave) A.total[i] = expf(-s0) * A.det[i] + s0 + seg_term;
}
__global__ __launch_bounds__(kWave) void k_kendall_bwd(const KendallArgs A) {
  const float s0 = A.log_vars[0], s1 = A.log_vars[1], e0 = expf(-s0), e1 = expf(-s1), seg = *A.seg;
  float gsum = 0.f, g0 = 0.f;
  for (int i = threadIdx.x; i < A.n; i += kWave) {
    const float g = A.g_total[i];
    A.g_det[i] = g * e0;
    gsum += g;
    g0 += g * (1.f - e0 * A.det[i]);
  }
  gsum = wave_group_sum(gsum, kWave);
  g0 = wave_group_sum(g0, kWave);
  if (threadIdx.x == 0) {
    *A.g_seg = gsum * e1;
    A.g_log_vars[0] = g0;
    A.g_log_vars[1] = gsum * (1.f - e1 * seg);
  }
}

}  // namespace mgacbam
