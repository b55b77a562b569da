// MaskECA kernels (reference mga_yolo/nn/modules/masked_eca.py:139-196): the second mask-guided attention
// block behind the same boundary.  Its pooled descriptor is k_pool minus the max branch, its gate is a k-tap 1-D conv over
// the channel axis instead of the MLP, and it has no spatial branch:
//
//   forward   k_eca_pool   x (1 read)        -> S/use/den, masked average (+GAP blend), sigma(mask) plane          [sweep]
//             k_eca_apply  x (1 read), avg   -> prologue: w = sigmoid(conv1d(avg)), g = 1 + a*(w-0.5) ; y = x*g    [sweep]
//   backward  k_eca_reduce x, gy (1 read)    -> gg[b,c] = sum_hw gy*x                                              [sweep]
//             k_eca_bwd    gy (+x for gmask) -> prologue: conv1d backward -> g_avg ; gx = gy*g + g_avg*wA ; gmask  [tile]
//                                               role workgroup: dW1d (k taps) and dbeta
//
// 3 E forward, 5 E backward (4 E when dL/dmask is not wanted), 4 launches per step for all pyramid levels together.
#pragma once
#include "args.cuh"
#include "common.cuh"

namespace mgacbam {

struct EcaCtx {            // saved forward statistics (mgacbam_eca_ctx_bytes)
  float* S; float* use; float* den;   // (B)
  float* avg; float* mavg; float* w;  // (B,C): pooled descriptor, masked average, sigmoid(conv1d(avg))
  float* splane;                      // (B,HW) sigma(mask) (zeros when there is no mask)
};
struct EcaScratch { float* gg; };     // (B,C) sum_hw gy*x

struct EcaFwdArgs {
  const void* x; const float* mask; void* y;
  EcaCtx c; const float* w1d; const float* beta; Geo g; Tune t;
};
struct EcaBwdArgs {
  const void* x; const float* mask; const void* gy; void* gx; float* gmask;
  float* gw; float* gbeta;
  EcaCtx c; const float* w1d; const float* beta; EcaScratch s; Geo g; Tune t;
  int nt;
};

// ---------------------------------------------------------------------------------------------
// k_eca_pool: masked average pooling with GAP fallback                             masked_eca.py:139-165
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int CPT, bool HAS_MASK>
__global__ __launch_bounds__(kBlock) void k_eca_pool(const Group<EcaFwdArgs> G) {
  __shared__ float red[64];
  int bid;
  const int l = find_level(G, blockIdx.x, bid);
  const EcaFwdArgs& A = G.lv[l];
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.pool_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int CPB = TY * CPT;
  const int ncg = (g.C + CPB - 1) / CPB;
  int b, cg;
  if (!xcd_sample_part(bid, g.B, ncg, b, cg)) return;
  const int c0 = cg * CPB + ty * CPT;
  const int nv = g.HW / VEC;
  const T* xr[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j)
    xr[j] = static_cast<const T*>(A.x) + (static_cast<size_t>(b) * g.C + min(c0 + j, g.C - 1)) * g.HW;
  const float* mb = HAS_MASK ? A.mask + static_cast<size_t>(b) * g.HW : nullptr;
  float* splane = A.c.splane + static_cast<size_t>(b) * g.HW;
  const bool writes_plane = (cg == 0 && ty == 0);
  float sx[CPT], sxs[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) { sx[j] = 0.f; sxs[j] = 0.f; }
  float ssum = 0.f;
  for (int i = tx; i < nv; i += TX) {
    float s[VEC];
    if (HAS_MASK) {
      float m[VEC];
      load_vec<float, VEC>(mb + static_cast<size_t>(i) * VEC, m);
#pragma unroll
      for (int e = 0; e < VEC; ++e) { s[e] = g.use_sigmoid ? sigmoid_fast(m[e]) : m[e]; ssum += s[e]; }   // masked_eca.py:146-147
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    }
    if (writes_plane) store_vec<float, VEC>(splane + static_cast<size_t>(i) * VEC, s);
    float xv[CPT][VEC];
#pragma unroll
    for (int j = 0; j < CPT; ++j) load_vec<T, VEC>(xr[j] + static_cast<size_t>(i) * VEC, xv[j]);
#pragma unroll
    for (int j = 0; j < CPT; ++j)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { sx[j] += xv[j][e]; if (HAS_MASK) sxs[j] += xv[j][e] * s[e]; }
  }
  float sums[2 * CPT + 1];
#pragma unroll
  for (int j = 0; j < CPT; ++j) { sums[j] = sx[j]; sums[CPT + j] = sxs[j]; }
  sums[2 * CPT] = ssum;
  row_sum<2 * CPT + 1>(sums, TX, tid, red);
  if (tx == 0) {
    const float N = static_cast<float>(g.HW);
    const float S = sums[2 * CPT];
    const float use = (S / N >= g.thr) ? 1.f : 0.f;            // masked_eca.py:152-153, 159
    const float den = fmaxf(S, g.eps);                         // :156, 164
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = c0 + j;
      if (c < g.C) {
        const size_t o = static_cast<size_t>(b) * g.C + c;
        const float gap = sums[j] / N;
        const float mavg = HAS_MASK ? sums[CPT + j] / den : gap;                 // :157, 165
        A.c.mavg[o] = mavg;
        A.c.avg[o] = HAS_MASK ? mavg * use + gap * (1.f - use) : gap;            // :161
      }
    }
    if (cg == 0 && ty == 0) {
      A.c.S[b] = HAS_MASK ? S : 0.f;
      A.c.use[b] = HAS_MASK ? use : 0.f;
      A.c.den[b] = HAS_MASK ? den : 1.f;
    }
  }
}

// sigmoid(conv1d(avg))[b,c] with zero padding over the channel axis                masked_eca.py:181-187
__device__ __forceinline__ float eca_gate_w(const float* __restrict__ avg_b, const float* __restrict__ w1d, int k, int C, int c) {
  const int pad = k / 2;
  float y = 0.f;
  for (int t = 0; t < k; ++t) {
    const int cc = c + t - pad;
    if (cc >= 0 && cc < C) y += w1d[t] * avg_b[cc];
  }
  return sigmoidf_(y);
}

// ---------------------------------------------------------------------------------------------
// k_eca_apply: y = x * (1 + softplus(beta) * (w - 0.5))                             masked_eca.py:187-193
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int CPT>
__global__ __launch_bounds__(kBlock) void k_eca_apply(const Group<EcaFwdArgs> G) {
  int bid;
  const int l = find_level(G, blockIdx.x, bid);
  const EcaFwdArgs& A = G.lv[l];
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.pool_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int CPB = TY * CPT;
  const int ncg = (g.C + CPB - 1) / CPB;
  int b, cg;
  if (!xcd_sample_part(bid, g.B, ncg, b, cg)) return;
  const int c0 = cg * CPB + ty * CPT;
  const int nv = g.HW / VEC;
  const float a = softplusf_(*A.beta);
  const T* xr[CPT];
  T* yr[CPT];
  float gate[CPT];
  bool live[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    live[j] = (c0 + j) < g.C;
    const int c = min(c0 + j, g.C - 1);
    const size_t o = static_cast<size_t>(b) * g.C + c;
    xr[j] = static_cast<const T*>(A.x) + o * g.HW;
    yr[j] = static_cast<T*>(A.y) + o * g.HW;
    const float w = eca_gate_w(A.c.avg + static_cast<size_t>(b) * g.C, A.w1d, g.k, g.C, c);
    gate[j] = 1.f + a * (w - 0.5f);                            // masked_eca.py:190
    if (tx == 0 && live[j]) A.c.w[o] = w;
  }
  for (int i = tx; i < nv; i += TX) {
    float xv[CPT][VEC];
#pragma unroll
    for (int j = 0; j < CPT; ++j) load_vec<T, VEC>(xr[j] + static_cast<size_t>(i) * VEC, xv[j]);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      float yv[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) yv[e] = xv[j][e] * gate[j];
      if (live[j]) store_vec_stream<T, VEC>(yr[j] + static_cast<size_t>(i) * VEC, yv, A.t.nt_stores);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_eca_reduce: gg[b,c] = sum_hw gy * x   (dL/dg)
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int CPT>
__global__ __launch_bounds__(kBlock) void k_eca_reduce(const Group<EcaBwdArgs> G) {
  __shared__ float red[64];
  int bid;
  const int l = find_level(G, blockIdx.x, bid);
  const EcaBwdArgs& A = G.lv[l];
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.pool_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int CPB = TY * CPT;
  const int ncg = (g.C + CPB - 1) / CPB;
  int b, cg;
  if (!xcd_sample_part(bid, g.B, ncg, b, cg)) return;
  const int c0 = cg * CPB + ty * CPT;
  const int nv = g.HW / VEC;
  const T* xr[CPT];
  const T* gr[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const size_t o = (static_cast<size_t>(b) * g.C + min(c0 + j, g.C - 1)) * g.HW;
    xr[j] = static_cast<const T*>(A.x) + o;
    gr[j] = static_cast<const T*>(A.gy) + o;
  }
  float acc[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
  for (int i = tx; i < nv; i += TX) {
    float xv[CPT][VEC], gv[CPT][VEC];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      load_vec<T, VEC>(xr[j] + static_cast<size_t>(i) * VEC, xv[j]);
      load_vec<T, VEC>(gr[j] + static_cast<size_t>(i) * VEC, gv[j]);
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[j] += xv[j][e] * gv[j][e];
  }
  row_sum<CPT>(acc, TX, tid, red);
  if (tx == 0) {
#pragma unroll
    for (int j = 0; j < CPT; ++j)
      if (c0 + j < g.C) A.s.gg[static_cast<size_t>(b) * g.C + c0 + j] = acc[j];
  }
}

// parameter-gradient role workgroups, one per output so each is a short reduction (a single workgroup doing all of them was
// the tail of the launch: 79 us):  r < k: dW1d[r] = sum_{b,c} gy1[b,c] * avg[b, c+r-pad]   r == 15: dbeta = sigmoid(beta) *
// sum_{b,c} gg*(w-0.5).  Fixed summation order => reproducible.
constexpr int kEcaRoles = 16;
__device__ __forceinline__ void eca_params_body(const EcaBwdArgs& A, const int r, float* red) {
  const Geo& g = A.g;
  const int tid = threadIdx.x, k = g.k, pad = k / 2, BC = g.B * g.C;
  if (r >= k && r != kEcaRoles - 1) return;
  const float a = softplusf_(*A.beta);
  float acc = 0.f;
#pragma unroll 4
  for (int o = tid; o < BC; o += kBlock) {
    const float w = A.c.w[o], gg = A.s.gg[o];
    if (r == kEcaRoles - 1) {
      acc += gg * (w - 0.5f);
    } else {
      const int b = o / g.C, c = o - b * g.C, cc = c + r - pad;
      if (cc >= 0 && cc < g.C) acc += a * gg * w * (1.f - w) * A.c.avg[static_cast<size_t>(b) * g.C + cc];
    }
  }
  acc = block_sum(acc, tid, red);
  if (tid == 0) { if (r == kEcaRoles - 1) *A.gbeta = sigmoidf_(*A.beta) * acc; else A.gw[r] = acc; }
}

// ---------------------------------------------------------------------------------------------
// k_eca_bwd: gx = gy*g + g_avg*wA ; gmask = (use/den) * (sum_c g_avg*x - K_b) * s(1-s)          (tile layout of k_bwd_apply)
//   prologue per workgroup (its sample): gy1 = a*gg*w(1-w) -> LDS; g_avg = conv1d^T(gy1) ; q[c] = {g, g_avg} ; K_b
//   the level's first 16 workgroups are the parameter-gradient roles (dW1d taps, dbeta)
//   LDS: [C gy1][2C q][256*VEC combine]
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, bool GMASK>
__global__ __launch_bounds__(kBlock) void k_eca_bwd(const Group<EcaBwdArgs> G) {
  extern __shared__ __align__(16) float smem[];
  __shared__ float red[8];
  int bid;
  const int l = find_level(G, blockIdx.x, bid);
  const EcaBwdArgs& A = G.lv[l];
  if (bid < kEcaRoles) { eca_params_body(A, bid, red); return; }   // 16 role ids keep the streaming ids XCD-aligned
  bid -= kEcaRoles;
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.chan_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int nv = g.HW / VEC;
  const int ntile = A.nt;
  int b, tile;
  if (!xcd_sample_part(bid, g.B, ntile, b, tile)) return;
  const int i = tile * TX + tx;
  const bool active = i < nv;
  const int ii = active ? i : nv - 1;
  const size_t base = static_cast<size_t>(b) * g.C * g.HW + static_cast<size_t>(ii) * VEC;
  const T* xp = static_cast<const T*>(A.x) + base;
  const T* gp = static_cast<const T*>(A.gy) + base;
  T* op = static_cast<T*>(A.gx) + base;
  const float a = softplusf_(*A.beta);
  const float N = static_cast<float>(g.HW);
  const bool has_mask = A.mask != nullptr;
  const int k = g.k, pad = k / 2;

  constexpr int UN = 2;                                        // first feature vectors are requested before the prologue
  float g0v[UN][VEC], x0v[UN][VEC];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const size_t co = static_cast<size_t>(min(ty + u * TY, g.C - 1)) * g.HW;
    load_vec<T, VEC>(gp + co, g0v[u]);
    if (GMASK) load_vec<T, VEC>(xp + co, x0v[u]);
  }
  float* s_gy1 = smem;
  float2* s_q = reinterpret_cast<float2*>(smem + g.C);
  float* sm = smem + 3 * g.C;
  for (int c = tid; c < g.C; c += kBlock) {
    const size_t o = static_cast<size_t>(b) * g.C + c;
    const float w = A.c.w[o];
    s_gy1[c] = a * A.s.gg[o] * w * (1.f - w);
  }
  __syncthreads();
  const float live = (has_mask && A.c.S[b] >= g.eps) ? 1.f : 0.f;
  float kpart = 0.f;
  for (int c = tid; c < g.C; c += kBlock) {
    float ga = 0.f;
    for (int t = 0; t < k; ++t) {                              // conv1d backward w.r.t. its input
      const int cc = c - t + pad;
      if (cc >= 0 && cc < g.C) ga += A.w1d[t] * s_gy1[cc];
    }
    const size_t o = static_cast<size_t>(b) * g.C + c;
    s_q[c] = make_float2(1.f + a * (A.c.w[o] - 0.5f), ga);
    kpart += ga * A.c.mavg[o] * live;
  }
  kpart = block_sum(kpart, tid, red);
  if (tid == 0) red[7] = kpart;
  __syncthreads();
  const float kb = red[7];

  float sv[VEC], wA[VEC], accp[VEC];
  load_vec<float, VEC>(A.c.splane + static_cast<size_t>(b) * g.HW + static_cast<size_t>(ii) * VEC, sv);
  const float use = A.c.use[b], den = A.c.den[b];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { wA[e] = has_mask ? (use * sv[e] / den + (1.f - use) / N) : 1.f / N; accp[e] = 0.f; }
  auto emit = [&](const float (&gv)[VEC], const float (&xv)[VEC], int c) {
    float ov[VEC];
    const float2 q = s_q[c];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      ov[e] = gv[e] * q.x + q.y * wA[e];
      if (GMASK) accp[e] += q.y * xv[e];
    }
    if (active) store_vec_stream<T, VEC>(op + static_cast<size_t>(c) * g.HW, ov, A.t.nt_stores);
  };
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const int c = ty + u * TY;
    if (c < g.C) emit(g0v[u], x0v[u], c);
  }
#pragma unroll 4
  for (int c = ty + UN * TY; c < g.C; c += TY) {
    float gv[VEC], xv[VEC];
    load_vec<T, VEC>(gp + static_cast<size_t>(c) * g.HW, gv);
    if (GMASK) load_vec<T, VEC>(xp + static_cast<size_t>(c) * g.HW, xv);
    emit(gv, xv, c);
  }
  if (GMASK) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) sm[tid * VEC + e] = accp[e];
    __syncthreads();
    if (ty == 0 && active) {
      for (int r = 1; r < TY; ++r)
#pragma unroll
        for (int e = 0; e < VEC; ++e) accp[e] += sm[(r * TX + tx) * VEC + e];
      float gm[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float gs = (use / den) * (accp[e] - kb);
        gm[e] = g.use_sigmoid ? gs * sv[e] * (1.f - sv[e]) : gs;
      }
      store_vec<float, VEC>(A.gmask + static_cast<size_t>(b) * g.HW + static_cast<size_t>(i) * VEC, gm);
    }
  }
}

}  // namespace mgacbam
