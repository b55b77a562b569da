// Backward kernels of the mask-guided CBAM block: the gradients autograd derives for
// mga_yolo/nn/modules/masked_cbam.py:87-171, restated in SURVEY.md section 8a and oracle/maskcbam_oracle.py.
//
// With a = softplus(beta), u = x*ca, v = u*sa, N = H*W:
//   k_bwd_reduce1  x, gy (1 read each) -> A[b,c] = sum_hw gy*x*sa, Q[b,c] = sum_hw gy*x   (per hw-tile partials)
//                                         g_pre[b,hw] = a * sa(1-sa) * sum_c ca*gy*x
//   k_bwd_convT    g_pre, planes       -> g_planes = conv_transpose(g_pre, Wsa) ; per-workgroup partials of dWsa
//   k_bwd_reduce2  x (1 read)          -> g_ca[b,c] = a*A + sum_hw x * ([c == cidx]*gp0 + gp1/C) ; g_z ; (ca*A - Q)
//   k_bwd_mlp      g_z                 -> g_h*, g_avg, g_mx (shared-MLP backward for both descriptors), K_b
//   k_bwd_apply    gy (+ x when dL/dmask is wanted) -> gx (1 write), gmask
//   k_bwd_finalize per-sample / per-workgroup partials -> dW1, db1, dW2, db2, dWsa, dbeta
//
// Algebra that keeps traffic at 2+1+3 passes: the term of g_ca that needs g_planes only needs x (not gy),
// and gx needs gy, the saved arg-max indices and planes but x only for the masked-average part of dL/dmask.
// All cross-workgroup sums are two-stage (partials, then one reader), never float atomics, so results are
// bitwise reproducible run to run.
#pragma once
#include "args.cuh"
#include "common.cuh"
#include "fwd.cuh"   // ConvTile / stage_tiles

namespace mgacbam {

// ---------------------------------------------------------------------------------------------
// k_bwd_reduce1     (thread layout of k_chan: one H*W vector per lane, rows take channel slices; TX <= 64)
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__device__ __forceinline__ void bwd_reduce1_body(const BwdArgs& A, const int bid, float* sm) {
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.chan_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int nv = g.HW / VEC;
  const int ntile = A.nt;
  const int b = bid / ntile, tile = bid - b * ntile;
  const int i = tile * TX + tx;
  const bool active = i < nv;
  const int ii = active ? i : nv - 1;
  const size_t base = static_cast<size_t>(b) * g.C * g.HW + static_cast<size_t>(ii) * VEC;
  const T* xp = static_cast<const T*>(A.x) + base;
  const T* gp = static_cast<const T*>(A.gy) + base;
  const float* cab = A.c.ca + static_cast<size_t>(b) * g.C;
  const float a = softplusf_(*A.p.beta);

  float sav[VEC];
  load_vec<float, VEC>(A.c.sa + static_cast<size_t>(b) * g.HW + static_cast<size_t>(ii) * VEC, sav);
  const float live = active ? 1.f : 0.f;
  float accp[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) accp[e] = 0.f;

#pragma unroll 4
  for (int c = ty; c < g.C; c += TY) {
    float xv[VEC], gv[VEC];
    load_vec<T, VEC>(xp + static_cast<size_t>(c) * g.HW, xv);
    load_vec<T, VEC>(gp + static_cast<size_t>(c) * g.HW, gv);
    const float cac = cab[c];
    float pa = 0.f, pq = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float p = xv[e] * gv[e] * live;
      accp[e] += cac * p;
      pa += p * sav[e];
      pq += p;
    }
    pa = wave_group_sum(pa, TX);
    pq = wave_group_sum(pq, TX);
    if (tx == 0) {
      const size_t o = (static_cast<size_t>(b) * g.C + c) * ntile + tile;
      A.s.A_part[o] = pa;
      A.s.Q_part[o] = pq;
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) sm[tid * VEC + e] = accp[e];
  __syncthreads();
  if (ty == 0 && active) {
    for (int r = 1; r < TY; ++r) {
      const int o = (r * TX + tx) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) accp[e] += sm[o + e];
    }
    float gpre[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) gpre[e] = a * accp[e] * sav[e] * (1.f - sav[e]);   // g_sa * sigmoid'
    store_vec<float, VEC>(A.s.gpre + static_cast<size_t>(b) * g.HW + static_cast<size_t>(i) * VEC, gpre);
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_bwd_reduce1(const Group<BwdArgs> G) {
  __shared__ float sm[kBlock * VEC];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  bwd_reduce1_body<T, VEC>(G.lv[l], local, sm);
}

// ---------------------------------------------------------------------------------------------
// k_bwd_convT: g_planes[p] = sum_{i,j} W[p,i,j] * g_pre[h-i+pad, w-j+pad]   (transposed conv, flipped kernel)
//   same tiling as k_conv_fwd; LDS holds the g_pre tile + halo; 4 adjacent pixels x 3 planes per thread.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(kBlock) void k_bwd_convT(const Group<BwdArgs> G) {
  extern __shared__ float smem[];
  int local;
  const int lvl = find_level(G, blockIdx.x, local);
  const BwdArgs& A = G.lv[lvl];
  const Geo& g = A.g;
  const int k = K ? K : g.k;
  const ConvTile c = conv_tile(g, A.t, k, local);
  const int tid = threadIdx.x;
  float* wts = smem;
  float* tg = smem + ((3 * k * k + 3) & ~3);
  for (int i = tid; i < 3 * k * k; i += kBlock) wts[i] = A.p.wsa[i];
  const float* gpre = A.s.gpre + static_cast<size_t>(c.b) * g.HW;
  stage_tiles<1>(tg, c, g, [&](int) { return gpre; });
  __syncthreads();
  const int TWQ = A.t.conv_twq;
  const int py = tid / TWQ, q = tid - py * TWQ;
  if (py >= c.TH) return;
  float acc[3][4];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[p][e] = 0.f;
  constexpr int KK = K ? K : 1;
  if (K) {
#pragma unroll 1
    for (int i = 0; i < KK; ++i) {
      const float* row = tg + (py + i) * c.PW + q * 4;
      float r[4 + KK - 1];
#pragma unroll
      for (int t = 0; t < 4 + KK - 1; ++t) r[t] = row[t];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const float* wr = wts + (p * KK + (KK - 1 - i)) * KK;               // flipped kernel row
#pragma unroll
        for (int j = 0; j < KK; ++j) {
          const float wv = wr[KK - 1 - j];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[p][e] += wv * r[e + j];
        }
      }
    }
  } else {
    for (int i = 0; i < k; ++i) {
      const float* row = tg + (py + i) * c.PW + q * 4;
      for (int j = 0; j < k; ++j) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const float wv = wts[(p * k + (k - 1 - i)) * k + (k - 1 - j)];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[p][e] += wv * row[e + j];
        }
      }
    }
  }
  const int yg = c.y0 + py;
  if (yg < g.H) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int xg = c.x0 + q * 4 + e;
      if (xg < g.W) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          A.s.gplanes[(static_cast<size_t>(c.b) * 3 + p) * g.HW + yg * g.W + xg] = acc[p][e];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dWsa partial of one conv tile:  sum_{px in tile} g_pre[px] * planes[p][px + (i,j) - pad]
//   thread t < 3*k*k owns output (p,i,j) and walks the tile's pixels (LDS only).  This is LDS/VALU work with no
//   HBM traffic, so these workgroups ride in front of the HBM-bound k_bwd_reduce2 grid ("role" blocks) and
//   overlap with it instead of costing a launch of their own.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bwd_wsa_body(const BwdArgs& A, const int bid, float* tile) {
  const Geo& g = A.g;
  const int k = g.k;
  const ConvTile c = conv_tile(g, A.t, k, bid);
  const int plane_elems = c.PH * c.PW;
  float* tg = tile;                      // g_pre tile + halo
  float* tp = tile + plane_elems;        // 3 plane tiles + halo
  const float* gpre = A.s.gpre + static_cast<size_t>(c.b) * g.HW;
  const float* pl = A.c.planes + static_cast<size_t>(c.b) * 3 * g.HW;
  stage_tiles<4>(tile, c, g, [&](int p) { return p == 0 ? gpre : pl + static_cast<size_t>(p - 1) * g.HW; });
  __syncthreads();
  const int nout = 3 * k * k;
  for (int o = threadIdx.x; o < nout; o += kBlock) {
    const int p = o / (k * k), r = o - p * k * k;
    const int i = r / k, j = r - i * k;
    const float* pp = tp + p * plane_elems + i * c.PW + j;      // planes[p][y + i - pad][x + j - pad]
    const float* gg = tg + c.pad * c.PW + c.pad;                // g_pre[y][x]
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int yy = 0; yy < c.TH; ++yy) {
      const float* prow = pp + yy * c.PW;
      const float* grow = gg + yy * c.PW;
#pragma unroll 4
      for (int xx = 0; xx < c.TW; xx += 4) {                    // TW is a multiple of 4; zero fill outside the image
        a0 += grow[xx] * prow[xx];
        a1 += grow[xx + 1] * prow[xx + 1];
        a2 += grow[xx + 2] * prow[xx + 2];
        a3 += grow[xx + 3] * prow[xx + 3];
      }
    }
    A.s.gwsa_part[static_cast<size_t>(o) * A.nconv + bid] = (a0 + a1) + (a2 + a3);
  }
}

// ---------------------------------------------------------------------------------------------
// k_bwd_reduce2     (thread layout of k_pool: rows of TX lanes sweep H*W for CPT channels each)
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int CPT>
__device__ __forceinline__ void bwd_reduce2_body(const BwdArgs& A, const int bid, float* red) {
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.pool_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int CPB = TY * CPT;
  const int ncg = (g.C + CPB - 1) / CPB;
  const int b = bid / ncg, cg = bid - b * ncg;
  const int c0 = cg * CPB + ty * CPT;
  const int nv = g.HW / VEC;
  const float invC = 1.f / static_cast<float>(g.C);

  const T* xr[CPT];
  int cj[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    cj[j] = min(c0 + j, g.C - 1);
    xr[j] = static_cast<const T*>(A.x) + (static_cast<size_t>(b) * g.C + cj[j]) * g.HW;
  }
  const float* gp0 = A.s.gplanes + static_cast<size_t>(b) * 3 * g.HW;
  const float* gp1 = gp0 + g.HW;
  const int* cidx = A.c.cidx + static_cast<size_t>(b) * g.HW;
  float acc[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) acc[j] = 0.f;

  for (int i = tx; i < nv; i += TX) {
    float g0[VEC], g1[VEC];
    int ci[VEC];
    load_vec<float, VEC>(gp0 + static_cast<size_t>(i) * VEC, g0);
    load_vec<float, VEC>(gp1 + static_cast<size_t>(i) * VEC, g1);
    load_ivec<VEC>(cidx + static_cast<size_t>(i) * VEC, ci);
    float xv[CPT][VEC];
#pragma unroll
    for (int j = 0; j < CPT; ++j) load_vec<T, VEC>(xr[j] + static_cast<size_t>(i) * VEC, xv[j]);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float wgt = g1[e] * invC + (ci[e] == cj[j] ? g0[e] : 0.f);   // mean backward + max backward
        acc[j] += xv[j][e] * wgt;
      }
    }
  }
  row_sum<CPT>(acc, TX, tid, red);
  if (tx == 0) {
    const float a = softplusf_(*A.p.beta);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = c0 + j;
      if (c < g.C) {
        const size_t o = static_cast<size_t>(b) * g.C + c;
        float As = 0.f, Qs = 0.f;
        for (int t = 0; t < A.nt; ++t) { As += A.s.A_part[o * A.nt + t]; Qs += A.s.Q_part[o * A.nt + t]; }
        const float ca = A.c.ca[o];
        const float gca = a * As + acc[j];
        A.s.gz[o] = gca * ca * (1.f - ca);
        A.s.gbq[o] = ca * As - Qs;                             // sum_hw gy*(v - x) for this (b,c)
      }
    }
  }
}

// grid of a level = [nconv dWsa role workgroups][streaming workgroups]
template <typename T, int VEC, int CPT>
__global__ __launch_bounds__(kBlock) void k_bwd_reduce2(const Group<BwdArgs> G) {
  extern __shared__ float smem[];       // role blocks: 4 conv tiles; streaming blocks: 64 floats of reduction scratch
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const BwdArgs& A = G.lv[l];
  if (local < A.nconv) bwd_wsa_body(A, local, smem);
  else bwd_reduce2_body<T, VEC, CPT>(A, local - A.nconv, smem);
}

// ---------------------------------------------------------------------------------------------
// k_bwd_mlp: backward of the shared MLP for both applications (inputs avg and mx); one workgroup per sample
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_bwd_mlp(const Group<BwdArgs> G) {
  extern __shared__ float sm[];
  __shared__ float red[8];
  int local;
  const int lvl = find_level(G, blockIdx.x, local);
  const BwdArgs& A = G.lv[lvl];
  const Geo& g = A.g;
  const int b = local, tid = threadIdx.x, C = g.C, h = g.hidden;
  float* s_gz = sm;           // C
  float* s_ga = sm + C;       // h  gh_avg
  float* s_gm = s_ga + h;     // h  gh_mx
  for (int c = tid; c < C; c += kBlock) s_gz[c] = A.s.gz[static_cast<size_t>(b) * C + c];
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int j = wave; j < h; j += kBlock / kWave) {
    float d = 0.f;
    for (int c = lane; c < C; c += kWave) d += A.p.w2[static_cast<size_t>(c) * h + j] * s_gz[c];
    d = wave_group_sum(d, kWave);
    if (lane == 0) {
      const float ga = A.c.h_avg[static_cast<size_t>(b) * h + j] > 0.f ? d : 0.f;   // relu backward
      const float gm = A.c.h_mx[static_cast<size_t>(b) * h + j] > 0.f ? d : 0.f;
      s_ga[j] = ga; s_gm[j] = gm;
      A.s.gh_avg[static_cast<size_t>(b) * h + j] = ga;
      A.s.gh_mx[static_cast<size_t>(b) * h + j] = gm;
    }
  }
  __syncthreads();
  const float N = static_cast<float>(g.HW);
  const bool has_mask = A.mask != nullptr;
  const float live = (has_mask && A.c.S[b] >= g.eps) ? 1.f : 0.f;   // clamp_min passes grad only when not clamped
  float kpart = 0.f;
  for (int c = tid; c < C; c += kBlock) {
    float ga = 0.f, gm = 0.f;
    for (int j = 0; j < h; ++j) { const float wv = A.p.w1[static_cast<size_t>(j) * C + c]; ga += wv * s_ga[j]; gm += wv * s_gm[j]; }
    const size_t o = static_cast<size_t>(b) * C + c;
    const int valid = A.c.valid[o];
    float4 q;
    q.x = A.c.ca[o];
    q.y = ga;                               // g_avg
    q.z = valid ? gm : 0.f;                 // routed to the arg-max position
    q.w = valid ? 0.f : gm / N;             // GAP fallback: spread uniformly
    reinterpret_cast<float4*>(A.s.chan4)[o] = q;
    kpart += ga * A.c.mavg[o] * live;
  }
  kpart = block_sum(kpart, tid, red);
  if (tid == 0) A.s.Kb[b] = kpart;
}

// ---------------------------------------------------------------------------------------------
// k_bwd_apply       (thread layout of k_chan)
//   gx = gy*((1-a) + a*sa*ca) + ca*([c==cidx]*gp0 + gp1/C) + g_avg*wA + [hw==amax]*g_mx_pt + g_mx_uni
//   gmask = (gp2 + (use/den) * (sum_c g_avg*x - K_b)) * s(1-s)
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, bool GMASK>
__device__ __forceinline__ void bwd_apply_body(const BwdArgs& A, const int bid, float* sm) {
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
  const size_t base = static_cast<size_t>(b) * g.C * g.HW + static_cast<size_t>(ii) * VEC;
  const T* xp = static_cast<const T*>(A.x) + base;
  const T* gp = static_cast<const T*>(A.gy) + base;
  T* op = static_cast<T*>(A.gx) + base;
  const float a = softplusf_(*A.p.beta);
  const float N = static_cast<float>(g.HW);
  const bool has_mask = A.mask != nullptr;
  const size_t po = static_cast<size_t>(b) * g.HW + static_cast<size_t>(ii) * VEC;

  float sav[VEC], g0[VEC], g1[VEC], sv[VEC], wA[VEC];
  int ci[VEC];
  load_vec<float, VEC>(A.c.sa + po, sav);
  load_vec<float, VEC>(A.s.gplanes + static_cast<size_t>(b) * 3 * g.HW + static_cast<size_t>(ii) * VEC, g0);
  load_vec<float, VEC>(A.s.gplanes + (static_cast<size_t>(b) * 3 + 1) * g.HW + static_cast<size_t>(ii) * VEC, g1);
  load_vec<float, VEC>(A.c.planes + (static_cast<size_t>(b) * 3 + 2) * g.HW + static_cast<size_t>(ii) * VEC, sv);
  load_ivec<VEC>(A.c.cidx + po, ci);
  const float use = A.c.use[b], den = A.c.den[b];
  const float invC = 1.f / static_cast<float>(g.C);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    g1[e] *= invC;
    sav[e] *= a;                                              // a*sa
    wA[e] = has_mask ? (use * sv[e] / den + (1.f - use) / N) : 1.f / N;
  }
  const float4* ch4 = reinterpret_cast<const float4*>(A.s.chan4) + static_cast<size_t>(b) * g.C;
  const int* amax = A.c.amax + static_cast<size_t>(b) * g.C;
  float accp[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) accp[e] = 0.f;

#pragma unroll 4
  for (int c = ty; c < g.C; c += TY) {
    float gv[VEC], xv[VEC], ov[VEC];
    load_vec<T, VEC>(gp + static_cast<size_t>(c) * g.HW, gv);
    if (GMASK) load_vec<T, VEC>(xp + static_cast<size_t>(c) * g.HW, xv);
    const float4 q = ch4[c];
    const int am = amax[c] - ii * VEC;                        // offset of the arg-max inside this vector, if any
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float r = gv[e] * ((1.f - a) + sav[e] * q.x);
      r += q.x * ((ci[e] == c ? g0[e] : 0.f) + g1[e]);
      r += q.y * wA[e] + q.w;
      r += (am == e) ? q.z : 0.f;
      ov[e] = r;
      if (GMASK) accp[e] += q.y * xv[e];
    }
    if (active) store_vec<T, VEC>(op + static_cast<size_t>(c) * g.HW, ov);
  }
  if (GMASK) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) sm[tid * VEC + e] = accp[e];
    __syncthreads();
    if (ty == 0 && active) {
      for (int r = 1; r < TY; ++r) {
        const int o = (r * TX + tx) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) accp[e] += sm[o + e];
      }
      float g2[VEC], gm[VEC];
      load_vec<float, VEC>(A.s.gplanes + (static_cast<size_t>(b) * 3 + 2) * g.HW + static_cast<size_t>(i) * VEC, g2);
      const float kb = A.s.Kb[b];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float gs = g2[e] + (use / den) * (accp[e] - kb);
        gm[e] = g.use_sigmoid ? gs * sv[e] * (1.f - sv[e]) : gs;
      }
      store_vec<float, VEC>(A.gmask + static_cast<size_t>(b) * g.HW + static_cast<size_t>(i) * VEC, gm);
    }
  }
}

template <typename T, int VEC, bool GMASK>
__global__ __launch_bounds__(kBlock) void k_bwd_apply(const Group<BwdArgs> G) {
  __shared__ float sm[kBlock * VEC];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  bwd_apply_body<T, VEC, GMASK>(G.lv[l], local, sm);
}

// ---------------------------------------------------------------------------------------------
// k_bwd_finalize: parameter gradients from per-sample / per-tile partials (one reader per output, fixed order)
//   workgroup roles inside a level: [0, nb_mlp) thread-per-output over dW1 (h*C), db1 (h), dW2 (C*h), db2 (C),
//   each a sum over the B samples; [nb_mlp, nb_mlp + nb_wsa) wave-per-output over the nconv dWsa partials;
//   last workgroup: dbeta = sigmoid(beta) * sum_{b,c} (ca*A - Q).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_bwd_finalize(const Group<BwdArgs> G) {
  __shared__ float red[8];
  int local;
  const int lvl = find_level(G, blockIdx.x, local);
  const BwdArgs& A = G.lv[lvl];
  const Geo& g = A.g;
  const int C = g.C, h = g.hidden, B = g.B, kk3 = 3 * g.k * g.k;
  const int n_w1 = h * C, n_b1 = h, n_w2 = C * h, n_b2 = C;
  const int total = n_w1 + n_b1 + n_w2 + n_b2;
  const int nb_mlp = (total + kBlock - 1) / kBlock;
  const int nb_wsa = (kk3 + 3) / 4;
  const int tid = threadIdx.x;
  if (local >= nb_mlp + nb_wsa) {                              // dbeta
    float acc = 0.f;
    for (int o = tid; o < B * C; o += kBlock) acc += A.s.gbq[o];
    acc = block_sum(acc, tid, red);
    if (tid == 0) *A.gbeta = sigmoidf_(*A.p.beta) * acc;
    return;
  }
  if (local >= nb_mlp) {                                       // dWsa[p,i,j] = sum over conv tiles (lanes stride the partials)
    const int o = (local - nb_mlp) * 4 + (tid >> 6), lane = tid & 63;
    if (o >= kk3) return;
    const float* part = A.s.gwsa_part + static_cast<size_t>(o) * A.nconv;
    float acc = 0.f;
    for (int t = lane; t < A.nconv; t += kWave) acc += part[t];
    acc = wave_group_sum(acc, kWave);
    if (lane == 0) A.gwsa[o] = acc;
    return;
  }
  int o = local * kBlock + tid;
  if (o >= total) return;
  if (o < n_w1) {                                              // dW1[j,c] = sum_b gh_avg[b,j]*avg[b,c] + gh_mx[b,j]*mx[b,c]
    const int j = o / C, c = o - j * C;
    float acc = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b)
      acc += A.s.gh_avg[b * h + j] * A.c.avg[static_cast<size_t>(b) * C + c] + A.s.gh_mx[b * h + j] * A.c.mx[static_cast<size_t>(b) * C + c];
    A.gw1[o] = acc;
    return;
  }
  o -= n_w1;
  if (o < n_b1) {                                              // db1[j] = sum_b gh_avg + gh_mx
    float acc = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) acc += A.s.gh_avg[b * h + o] + A.s.gh_mx[b * h + o];
    A.gb1[o] = acc;
    return;
  }
  o -= n_b1;
  if (o < n_w2) {                                              // dW2[c,j] = sum_b g_z[b,c] * (h_avg[b,j] + h_mx[b,j])
    const int c = o / h, j = o - c * h;
    float acc = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) acc += A.s.gz[static_cast<size_t>(b) * C + c] * (A.c.h_avg[b * h + j] + A.c.h_mx[b * h + j]);
    A.gw2[o] = acc;
    return;
  }
  o -= n_w2;
  {                                                            // db2[c] = 2 * sum_b g_z[b,c]   (bias used twice)
    float acc = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) acc += A.s.gz[static_cast<size_t>(b) * C + o];
    A.gb2[o] = 2.f * acc;
  }
}

}  // namespace mgacbam
