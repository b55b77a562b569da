// Forward kernels of the mask-guided CBAM block (reference mga_yolo/nn/modules/masked_cbam.py:87-171).
//
//   k_pool      x (1 read)            -> avg, mx (+ arg-max), S/use/den, sigma(mask) plane     [HBM-bound]
//   k_mlp_fwd   avg, mx               -> h_avg, h_mx, ca                                        [tiny]
//   k_chan      x (1 read), ca        -> planes[max_c, mean_c], cidx                            [HBM-bound]
//   k_conv_fwd  planes                -> sa                                                     [tiny]
//   k_apply     x (1 read), ca, sa    -> y (1 write)                                            [HBM-bound]
//
// No kernel materialises cam_out / sam_out / the expanded mask (the reference makes ~15 full-size
// temporaries).  Thread layout everywhere: 256 threads = TY rows x TX lanes, TX lanes run along H*W
// (contiguous, VEC elements per lane per access => 16-byte coalesced accesses for fp32).
#pragma once
#include "args.cuh"
#include "common.cuh"

namespace mgacbam {

// ---------------------------------------------------------------------------------------------
// k_pool: masked average / masked max pooling over H*W for every (b, c)      masked_cbam.py:87-121
//   workgroup = (sample b, CPB = TY*CPT channels); each row of TX lanes sweeps the whole H*W extent of its
//   CPT channels, so sigma(mask) is evaluated once per position and reused from registers for CPT
//   channels; the per-channel sums and the (max, first arg-max) pair are then reduced with wave
//   shuffles (+ one LDS step when a row spans several waves).
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int CPT, bool HAS_MASK>
__device__ __forceinline__ void pool_body(const FwdArgs& A, const int bid, float* red) {
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.pool_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int CPB = TY * CPT;
  const int ncg = (g.C + CPB - 1) / CPB;
  const int b = bid / ncg, cg = bid - b * ncg;
  const int c0 = cg * CPB + ty * CPT;
  const int nv = g.HW / VEC;

  const T* xr[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = min(c0 + j, g.C - 1);                      // clamp: loads stay unconditional, stores are guarded
    xr[j] = static_cast<const T*>(A.x) + (static_cast<size_t>(b) * g.C + c) * g.HW;
  }
  const float* mb = HAS_MASK ? A.mask + static_cast<size_t>(b) * g.HW : nullptr;
  float* splane = A.c.planes + (static_cast<size_t>(b) * 3 + 2) * g.HW;
  const bool writes_plane = (cg == 0 && ty == 0);

  float sx[CPT], sxs[CPT], vmax[CPT];
  int imax[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) { sx[j] = 0.f; sxs[j] = 0.f; vmax[j] = -FLT_MAX; imax[j] = 0; }
  float ssum = 0.f;

  for (int i = tx; i < nv; i += TX) {
    float s[VEC];
    bool sel[VEC];
    if (HAS_MASK) {
      float m[VEC];
      load_vec<float, VEC>(mb + static_cast<size_t>(i) * VEC, m);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        s[e] = g.use_sigmoid ? sigmoid_fast(m[e]) : m[e];   // masked_cbam.py:93-94
        sel[e] = s[e] > 0.5f;                               // masked_cbam.py:116
        ssum += s[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { s[e] = 0.f; sel[e] = true; }   // masked_cbam.py:138 (zero plane)
    }
    if (writes_plane) store_vec<float, VEC>(splane + static_cast<size_t>(i) * VEC, s);
    float xv[CPT][VEC];
#pragma unroll
    for (int j = 0; j < CPT; ++j) load_vec<T, VEC>(xr[j] + static_cast<size_t>(i) * VEC, xv[j]);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float v = xv[j][e];
        sx[j] += v;
        if (HAS_MASK) sxs[j] += v * s[e];
        if (sel[e] && v > vmax[j]) { vmax[j] = v; imax[j] = i * VEC + e; }   // strict > : first max wins
      }
    }
  }

  float sums[2 * CPT + 1];
#pragma unroll
  for (int j = 0; j < CPT; ++j) { sums[j] = sx[j]; sums[CPT + j] = sxs[j]; }
  sums[2 * CPT] = ssum;
  row_sum<2 * CPT + 1>(sums, TX, tid, red);
  row_argmax<CPT>(vmax, imax, TX, tid, red);

  if (tx == 0) {
    const float N = static_cast<float>(g.HW);
    const float S = sums[2 * CPT];
    const float use = (S / N >= g.thr) ? 1.f : 0.f;          // masked_cbam.py:97-98
    const float den = fmaxf(S, g.eps);                       // masked_cbam.py:99
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = c0 + j;
      if (c < g.C) {
        const size_t o = static_cast<size_t>(b) * g.C + c;
        const float gap = sums[j] / N;                       // masked_cbam.py:101
        float avg, mavg, mxo;
        int valid;
        if (HAS_MASK) {
          mavg = sums[CPT + j] / den;                        // masked_cbam.py:100
          avg = mavg * use + gap * (1.f - use);              // masked_cbam.py:102
          valid = isclosef_(vmax[j], -FLT_MAX) ? 0 : 1;      // masked_cbam.py:120
          mxo = valid ? vmax[j] : gap;                       // masked_cbam.py:121
        } else {
          mavg = gap; avg = gap; valid = 1; mxo = vmax[j];   // masked_cbam.py:90-91, 107-108
        }
        A.c.avg[o] = avg; A.c.mavg[o] = mavg; A.c.mx[o] = mxo;
        A.c.valid[o] = valid; A.c.amax[o] = imax[j];
      }
    }
    if (cg == 0 && ty == 0) {
      A.c.S[b] = HAS_MASK ? S : 0.f;
      A.c.use[b] = HAS_MASK ? use : 0.f;
      A.c.den[b] = HAS_MASK ? den : 1.f;
    }
  }
}

template <typename T, int VEC, int CPT, bool HAS_MASK>
__global__ __launch_bounds__(kBlock) void k_pool(const Group<FwdArgs> G) {
  __shared__ float red[64];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  pool_body<T, VEC, CPT, HAS_MASK>(G.lv[l], local, red);
}

// ---------------------------------------------------------------------------------------------
// k_mlp_fwd: shared MLP on both descriptors, channel gate                     masked_cbam.py:54-58, 128-129
//   one workgroup per sample; <= 74k MAC per sample -- far too small for MFMA
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_mlp_fwd(const Group<FwdArgs> G) {
  extern __shared__ float sm[];
  int local;
  const int lvl = find_level(G, blockIdx.x, local);
  const FwdArgs& A = G.lv[lvl];
  const Geo& g = A.g;
  const int b = local, tid = threadIdx.x, C = g.C, h = g.hidden;
  float* s_avg = sm;
  float* s_mx = sm + C;
  float* s_ha = sm + 2 * C;
  float* s_hm = s_ha + h;
  for (int c = tid; c < C; c += kBlock) {
    s_avg[c] = A.c.avg[static_cast<size_t>(b) * C + c];
    s_mx[c] = A.c.mx[static_cast<size_t>(b) * C + c];
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int j = wave; j < h; j += kBlock / kWave) {
    const float* w = A.p.w1 + static_cast<size_t>(j) * C;
    float da = 0.f, dm = 0.f;
    for (int c = lane; c < C; c += kWave) { const float wv = w[c]; da += wv * s_avg[c]; dm += wv * s_mx[c]; }
    da = wave_group_sum(da, kWave);
    dm = wave_group_sum(dm, kWave);
    if (lane == 0) {
      const float bj = A.p.b1[j];
      const float ha = fmaxf(da + bj, 0.f), hm = fmaxf(dm + bj, 0.f);
      s_ha[j] = ha; s_hm[j] = hm;
      A.c.h_avg[static_cast<size_t>(b) * h + j] = ha;
      A.c.h_mx[static_cast<size_t>(b) * h + j] = hm;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += kBlock) {
    const float* w = A.p.w2 + static_cast<size_t>(c) * h;
    float za = 0.f, zm = 0.f;
    for (int j = 0; j < h; ++j) { const float wv = w[j]; za += wv * s_ha[j]; zm += wv * s_hm[j]; }
    const float bc = A.p.b2[c];
    const float z = (za + bc) + (zm + bc);                   // masked_cbam.py:128 (bias enters twice)
    A.c.ca[static_cast<size_t>(b) * C + c] = sigmoidf_(z);   // masked_cbam.py:129
  }
}

// ---------------------------------------------------------------------------------------------
// k_chan: u = x * ca ; per pixel max_c u (+ first arg-max channel) and mean_c u        masked_cbam.py:130,135-136
//   workgroup = (sample b, tile of TX vectors along H*W); row ty handles channels ty, ty+TY, ...;
//   the TY partial (max, idx, sum) triples are combined through LDS.
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__device__ __forceinline__ void chan_body(const FwdArgs& A, const int bid, float* sm) {
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.chan_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int nv = g.HW / VEC;
  const int ntile = (nv + TX - 1) / TX;
  const int b = bid / ntile, tile = bid - b * ntile;
  const int i = tile * TX + tx;
  const bool active = i < nv;
  const int ii = active ? i : nv - 1;
  const T* xp = static_cast<const T*>(A.x) + static_cast<size_t>(b) * g.C * g.HW + static_cast<size_t>(ii) * VEC;
  const float* cab = A.c.ca + static_cast<size_t>(b) * g.C;

  float vmax[VEC], vsum[VEC];
  int vidx[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { vmax[e] = -INFINITY; vsum[e] = 0.f; vidx[e] = ty; }
#pragma unroll 4
  for (int c = ty; c < g.C; c += TY) {
    float xv[VEC];
    load_vec<T, VEC>(xp + static_cast<size_t>(c) * g.HW, xv);
    const float cac = cab[c];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float u = xv[e] * cac;                           // masked_cbam.py:130
      vsum[e] += u;
      if (u > vmax[e]) { vmax[e] = u; vidx[e] = c; }
    }
  }
  float* smax = sm;
  int* sidx = reinterpret_cast<int*>(sm + kBlock * VEC);
  float* ssum = sm + 2 * kBlock * VEC;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    smax[tid * VEC + e] = vmax[e]; sidx[tid * VEC + e] = vidx[e]; ssum[tid * VEC + e] = vsum[e];
  }
  __syncthreads();
  if (ty == 0 && active) {
    for (int r = 1; r < TY; ++r) {
      const int o = (r * TX + tx) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        argmax_combine(vmax[e], vidx[e], smax[o + e], sidx[o + e]);
        vsum[e] += ssum[o + e];
      }
    }
    float pavg[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) pavg[e] = vsum[e] / static_cast<float>(g.C);   // masked_cbam.py:136
    float* pl = A.c.planes + static_cast<size_t>(b) * 3 * g.HW + static_cast<size_t>(i) * VEC;
    store_vec<float, VEC>(pl, vmax);
    store_vec<float, VEC>(pl + g.HW, pavg);
    store_ivec<VEC>(A.c.cidx + static_cast<size_t>(b) * g.HW + static_cast<size_t>(i) * VEC, vidx);
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_chan(const Group<FwdArgs> G) {
  __shared__ float sm[kBlock * VEC * 3];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  chan_body<T, VEC>(G.lv[l], local, sm);
}

// ---------------------------------------------------------------------------------------------
// conv tiles (shared by k_conv_fwd and the backward conv kernels)
//   a tile is TH rows x TW = 4*TWQ columns of one sample; NP planes of (TH+k-1) x (TW+k-1) floats (tile + halo,
//   zero padded outside the image) are staged in LDS.  Loads are issued 8 at a time per thread before any LDS
//   store so the global-load latency is paid once per batch, not once per element.
// ---------------------------------------------------------------------------------------------
struct ConvTile {
  int b, y0, x0, TW, TH, PW, PH, pad, k;
};
__device__ __forceinline__ ConvTile conv_tile(const Geo& g, const Tune& t, int k, int bid) {
  ConvTile c;
  c.k = k; c.pad = k / 2;
  c.TW = t.conv_twq * 4; c.TH = t.conv_th;
  c.PW = c.TW + k - 1; c.PH = c.TH + k - 1;
  const int tiles_x = (g.W + c.TW - 1) / c.TW, tiles_y = (g.H + c.TH - 1) / c.TH;
  const int txi = bid % tiles_x; bid /= tiles_x;
  const int tyi = bid % tiles_y;
  c.b = bid / tiles_y;
  c.y0 = tyi * c.TH; c.x0 = txi * c.TW;
  return c;
}
// src(p) -> pointer to plane p of this sample (H*W floats)
template <int NP, typename SrcFn>
__device__ __forceinline__ void stage_tiles(float* tile, const ConvTile& c, const Geo& g, SrcFn src) {
  const int total = NP * c.PH * c.PW;
  const unsigned mpw = 0xFFFFFFFFu / static_cast<unsigned>(c.PW) + 1u;    // idx / PW == umulhi(idx, mpw) for idx < 2^16
  const unsigned mph = 0xFFFFFFFFu / static_cast<unsigned>(c.PH) + 1u;
  for (int base = 0; base < total; base += kBlock * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kBlock + threadIdx.x;
      const unsigned r = __umulhi(static_cast<unsigned>(idx), mpw);        // row over all planes
      const int xx = idx - static_cast<int>(r) * c.PW;
      const unsigned p = __umulhi(r, mph);
      const int yy = static_cast<int>(r) - static_cast<int>(p) * c.PH;
      const int gy_ = c.y0 + yy - c.pad, gx_ = c.x0 + xx - c.pad;
      v[u] = 0.f;
      if (idx < total && gy_ >= 0 && gy_ < g.H && gx_ >= 0 && gx_ < g.W) v[u] = src(static_cast<int>(p))[gy_ * g.W + gx_];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kBlock + threadIdx.x;
      if (idx < total) tile[idx] = v[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_conv_fwd: sa = sigmoid(conv_kxk([max_c u, mean_c u, sigma(mask)]))         masked_cbam.py:146-147
//   each thread produces 4 adjacent pixels so one LDS row segment of 4+k-1 floats feeds k taps x 4 outputs;
//   weights sit in LDS (broadcast reads).  K > 0: compile-time kernel size; K == 0: any odd k <= 15.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(kBlock) void k_conv_fwd(const Group<FwdArgs> G) {
  extern __shared__ float smem[];
  int local;
  const int lvl = find_level(G, blockIdx.x, local);
  const FwdArgs& A = G.lv[lvl];
  const Geo& g = A.g;
  const int k = K ? K : g.k;
  const ConvTile c = conv_tile(g, A.t, k, local);
  const int tid = threadIdx.x;
  float* wts = smem;                                   // 3*k*k (rounded up to a multiple of 4 floats)
  float* tile = smem + ((3 * k * k + 3) & ~3);
  for (int i = tid; i < 3 * k * k; i += kBlock) wts[i] = A.p.wsa[i];
  const float* pl = A.c.planes + static_cast<size_t>(c.b) * 3 * g.HW;
  stage_tiles<3>(tile, c, g, [&](int p) { return pl + static_cast<size_t>(p) * g.HW; });
  __syncthreads();
  const int TWQ = A.t.conv_twq;
  const int py = tid / TWQ, q = tid - py * TWQ;
  if (py >= c.TH) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int KK = K ? K : 1;
  if (K) {
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < KK; ++i) {
        const float* row = tile + (p * c.PH + py + i) * c.PW + q * 4;
        const float* wr = wts + (p * KK + i) * KK;
        float r[4 + KK - 1];
#pragma unroll
        for (int t = 0; t < 4 + KK - 1; ++t) r[t] = row[t];
#pragma unroll
        for (int j = 0; j < KK; ++j) {
          const float wv = wr[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += wv * r[e + j];
        }
      }
    }
  } else {
    for (int p = 0; p < 3; ++p)
      for (int i = 0; i < k; ++i) {
        const float* row = tile + (p * c.PH + py + i) * c.PW + q * 4;
        const float* wr = wts + (p * k + i) * k;
        for (int j = 0; j < k; ++j) {
          const float wv = wr[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += wv * row[e + j];
        }
      }
  }
  const int yg = c.y0 + py;
  if (yg < g.H) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int xg = c.x0 + q * 4 + e;
      if (xg < g.W) A.c.sa[static_cast<size_t>(c.b) * g.HW + yg * g.W + xg] = sigmoidf_(acc[e]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_apply: y = x + alpha * (x*ca*sa - x)                                       masked_cbam.py:130,148,166-171
//   same workgroup shape as k_pool: sa is loaded once per position and reused for CPT channels.
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int CPT>
__device__ __forceinline__ void apply_body(const FwdArgs& A, const int bid) {
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.apply_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int CPB = TY * CPT;
  const int ncg = (g.C + CPB - 1) / CPB;
  const int b = bid / ncg, cg = bid - b * ncg;
  const int c0 = cg * CPB + ty * CPT;
  const int nv = g.HW / VEC;
  const float a = softplusf_(*A.p.beta);                     // masked_cbam.py:150-152

  const T* xr[CPT];
  T* yr[CPT];
  float cac[CPT];
  bool live[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    live[j] = (c0 + j) < g.C;
    const int c = min(c0 + j, g.C - 1);
    const size_t o = static_cast<size_t>(b) * g.C + c;
    xr[j] = static_cast<const T*>(A.x) + o * g.HW;
    yr[j] = static_cast<T*>(A.y) + o * g.HW;
    cac[j] = A.c.ca[o];
  }
  const float* sab = A.c.sa + static_cast<size_t>(b) * g.HW;
  for (int i = tx; i < nv; i += TX) {
    float sav[VEC];
    load_vec<float, VEC>(sab + static_cast<size_t>(i) * VEC, sav);
    float xv[CPT][VEC];
#pragma unroll
    for (int j = 0; j < CPT; ++j) load_vec<T, VEC>(xr[j] + static_cast<size_t>(i) * VEC, xv[j]);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      float yv[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float u = xv[j][e] * cac[j];                   // masked_cbam.py:130
        const float v = u * sav[e];                          // masked_cbam.py:148
        yv[e] = xv[j][e] + a * (v - xv[j][e]);               // masked_cbam.py:171
      }
      if (live[j]) store_vec<T, VEC>(yr[j] + static_cast<size_t>(i) * VEC, yv);
    }
  }
}

template <typename T, int VEC, int CPT>
__global__ __launch_bounds__(kBlock) void k_apply(const Group<FwdArgs> G) {
  int local;
  const int l = find_level(G, blockIdx.x, local);
  apply_body<T, VEC, CPT>(G.lv[l], local);
}

// ---------------------------------------------------------------------------------------------
// k_resize_nearest: dst[p, y, x] = src[p, sy(y), sx(x)],  s(d) = min(floor(d * in/out), in-1) in fp32
//   mga_yolo/nn/losses/segmentation.py:103-110 -> F.interpolate(mode="nearest"): the integer index path
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_resize_nearest(const float* __restrict__ src, float* __restrict__ dst,
                                                           int n_planes, int in_h, int in_w, int out_h, int out_w) {
  const float sh = static_cast<float>(in_h) / static_cast<float>(out_h);
  const float sw = static_cast<float>(in_w) / static_cast<float>(out_w);
  const size_t total = static_cast<size_t>(n_planes) * out_h * out_w;
  for (size_t o = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; o < total;
       o += static_cast<size_t>(gridDim.x) * kBlock) {
    const int x = static_cast<int>(o % out_w);
    const size_t r = o / out_w;
    const int y = static_cast<int>(r % out_h);
    const size_t p = r / out_h;
    const int sy = min(static_cast<int>(floorf(static_cast<float>(y) * sh)), in_h - 1);
    const int sx = min(static_cast<int>(floorf(static_cast<float>(x) * sw)), in_w - 1);
    dst[o] = src[(p * in_h + sy) * in_w + sx];
  }
}

}  // namespace mgacbam
