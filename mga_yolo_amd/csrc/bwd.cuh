// Backward kernels of the mask-guided CBAM block: the gradients autograd derives for
// mga_yolo/nn/modules/masked_cbam.py:87-171, restated in SURVEY.md section 8a and oracle/maskcbam_oracle.py.
//
// With a = softplus(beta), u = x*ca, v = u*sa, N = H*W; three launches per step (four without MGACBAM_BWD_FOLD), each covering
// P3+P4+P5:
//   k_bwd_reduce1  x, gy (1 read each) -> A[b,c] = sum_hw gy*x*sa, D[b,c] = sum_hw gy*(v-x)   (per hw-tile partials)
//                                         g_pre[b,hw] = a * sa(1-sa) * sum_c ca*gy*x
//   k_bwd_convT    g_pre               -> g_planes = conv_transpose(g_pre, Wsa)     [tiny; with MGACBAM_BWD_FOLD its tiles are the
//                                         last workgroups of the k_bwd_reduce1 launch and take g_pre over inside the launch]
//   k_bwd_reduce2  x (1 read)          -> g_ca[b,c] = a*A + sum_hw x * ([c == cidx]*gp0 + gp1/C) ; g_z ; D ;
//                                         per-channel-group partials of W2^T g_z
//                                         + role workgroups (k_bwd_wsa body): dWsa tile partials
//   k_bwd_apply    gy (+ x when dL/dmask is wanted) -> prologue: g_h from the partials, g_avg, g_mx = W1^T g_h ;
//                                         body: gx (1 write), gmask
//                                         + role workgroups (k_bwd_params body): dW1 db1 dW2 db2 dWsa dbeta
//   (k_bwd_wsa / k_bwd_params also exist as stand-alone launches: MGACBAM_BWD_PARAMS, the split form for callers that want
//    the parameter gradients complete before the largest kernel runs.)
//
// Algebra that keeps traffic at 2+1+3 passes: the term of g_ca that needs g_planes only needs x (not gy),
// and gx needs gy, the saved arg-max indices and planes but x only for the masked-average part of dL/dmask.
// All cross-workgroup sums are two-stage (partials, then one reader, fixed order), never float atomics, so results are
// bitwise reproducible run to run.
#pragma once
#include "args.cuh"
#include "common.cuh"
#include "tile.cuh"   // ConvTile / stage_window

// tuning hooks (A/B builds): unroll 1/2/4/8 and prefetch 1/2 of k_bwd_apply all measure within noise (54.5-57.7 us)
#ifndef MGACBAM_BAPPLY_UNROLL
#define MGACBAM_BAPPLY_UNROLL 4
#endif
constexpr int kPghLds = 2048;   // floats of LDS for k_bwd_reduce2's hidden-gradient partials (rows x hidden chunk)

#ifndef MGACBAM_EARLY_H
#define MGACBAM_EARLY_H 8    // hidden sizes up to this issue k_bwd_apply's second-phase prologue loads early
#endif
#ifndef MGACBAM_BAPPLY_UN
#define MGACBAM_BAPPLY_UN 2
#endif

// A/B hook: occupancy cap of the two read-only backward kernels (-DMGACBAM_BWD_WAVES=4: what they would get inside one launch with
// k_bwd_apply, whose 104 VGPRs allow 4 waves per SIMD)
#ifdef MGACBAM_BWD_WAVES
#define BWD_OCC __attribute__((amdgpu_waves_per_eu(MGACBAM_BWD_WAVES, MGACBAM_BWD_WAVES)))
#else
#define BWD_OCC
#endif

namespace mgacbam {

// ---------------------------------------------------------------------------------------------
// k_bwd_reduce1     (thread layout of k_chan: one H*W vector per lane, rows take channel slices; TX <= 64)
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, bool FOLD = false>   // FOLD: the transposed-conv tiles ride at the end of this launch (k_bwd_reduce1, ROLES)
__device__ __forceinline__ void bwd_reduce1_body(const BwdArgs& A, const int bid, float* smem) {
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
  const float a = softplusf_(*A.p.beta);
  // LDS: [C ca][2*C tile partials (A then D)][256*VEC combine]
  float* s_ca = smem;
  float* s_aq = smem + g.C;
  float* sm = s_aq + 2 * g.C;
  for (int c = tid; c < g.C; c += kBlock) s_ca[c] = A.c.ca[static_cast<size_t>(b) * g.C + c];

  float sav[VEC];
  load_vec<float, VEC>(A.c.sa + static_cast<size_t>(b) * g.HW + static_cast<size_t>(ii) * VEC, sav);
  const float live = active ? 1.f : 0.f;
  float accp[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) accp[e] = 0.f;
  __syncthreads();

#pragma unroll 4
  for (int c = ty; c < g.C; c += TY) {
    float xv[VEC], gv[VEC];
    load_vec<T, VEC>(xp + static_cast<size_t>(c) * g.HW, xv);
    load_vec<T, VEC>(gp + static_cast<size_t>(c) * g.HW, gv);
    const float cac = s_ca[c];
    float pa = 0.f, pq = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float p = xv[e] * gv[e] * live;
      accp[e] += cac * p;
      pa += p * sav[e];
      pq += p * (cac * sav[e] - 1.f);                           // gy*(v - x), summed directly: ca*A - Q would cancel two large sums
    }
    pa = wave_group_sum(pa, TX);
    pq = wave_group_sum(pq, TX);
    if (tx == 0) { s_aq[c] = pa; s_aq[g.C + c] = pq; }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) sm[tid * VEC + e] = accp[e];
  __syncthreads();
  // this tile's partials of A[b,c] and D[b,c], written as two contiguous runs of C floats (layout (B, nt, 2, C))
  float* part = A.s.A_part + (static_cast<size_t>(b) * ntile + tile) * 2 * g.C;
  if (FOLD && A.merged) { for (int c = tid; c < 2 * g.C; c += kBlock) st_agent(part + c, s_aq[c]); }   // read by this launch's sweep workgroups
  else { for (int c = tid; c < 2 * g.C; c += kBlock) part[c] = s_aq[c]; }
  if (ty == 0 && active) {
    for (int r = 1; r < TY; ++r) {
      const int o = (r * TX + tx) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) accp[e] += sm[o + e];
    }
    float gpre[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) gpre[e] = a * accp[e] * sav[e] * (1.f - sav[e]);   // g_sa * sigmoid'
    float* gq = A.s.gpre + static_cast<size_t>(b) * g.HW + static_cast<size_t>(i) * VEC;
    if (FOLD) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) st_agent(gq + e, gpre[e]);   // consumed inside this launch by the transposed-conv roles
    } else {
      store_vec<float, VEC>(gq, gpre);
    }
  }
  if (FOLD) handoff_publish(A.c.sync + A.bflag0 + static_cast<size_t>(b) * A.nflag + tile);   // this tile's g_pre rows are out
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_bwd_reduce1(const Group<BwdArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  bwd_reduce1_body<T, VEC>(G.lv[l], local, smem);
}

// ---------------------------------------------------------------------------------------------
// k_bwd_convT: g_planes[p] = sum_{i,j} W[p,i,j] * g_pre[h-i+pad, w-j+pad]   (transposed conv = correlation with the
//   flipped kernel).  Tile = TH rows x TW columns of one sample, g_pre tile + halo in LDS (one load batch), 4 adjacent
//   pixels x 3 planes per thread.  On the critical path between k_bwd_reduce1 and k_bwd_reduce2, so it does nothing else.
// ---------------------------------------------------------------------------------------------
// COH: g_pre was published inside this launch by k_bwd_reduce1's tile workgroups (lower ids): wait (bounded) for the tiles whose
// pixels the window touches, then read it with agent-scope loads
template <int K, bool COH>
__device__ __forceinline__ void bwd_convT_body(const BwdArgs& A, const int local, float* smem) {
  const Geo& g = A.g;
  const int k = K ? K : g.k;
  const ConvTile c = conv_tile(g, A.t, k, local, A.t.conv_th);
  const int tid = threadIdx.x;
  float* wts = smem;
  float* tg = smem + ((3 * k * k + 3) & ~3);   // g_pre tile + halo
  for (int i = tid; i < 3 * k * k; i += kBlock) wts[i] = A.p.wsa[i];
  const float* gpre = A.s.gpre + static_cast<size_t>(c.b) * g.HW;
  bool bad = false;
  if (COH) {
    // generation counters: every k_bwd_reduce1 tile and every folded conv tile bumps its own flag exactly once per folded launch, so
    // after n such launches every flag reads n -- nothing is reset, and a backward that stops half-way leaves a consistent state
    int* own = A.c.sync + A.cflag0 + local;
    const int gen = static_cast<int>(static_cast<unsigned>(ld_agent(own)) + 1u);
    const int TP = A.t.chan_tx * A.vec;                        // pixels per k_bwd_reduce1 tile
    const int ra = max(c.y0 - c.pad, 0), rb = min(c.y0 + c.TH - 1 + c.pad, g.H - 1);
    bad = handoff_wait(A.c.sync + A.bflag0 + static_cast<size_t>(c.b) * A.nflag, (ra * g.W) / TP, ((rb + 1) * g.W - 1) / TP, gen,
                       A.c.sync + static_cast<size_t>(g.B) * A.nflag, A.spin_limit);
    if (tid == 0 && !A.merged) __hip_atomic_fetch_add(own, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  stage_window<8>(tg, 1, c.PH, c.PW, c.y0 - c.pad, c.x0 - c.pad, g, [&](int, int off) { return COH ? ld_agent(gpre + off) : gpre[off]; });
  __syncthreads();
  const int TWQ = A.t.conv_twq;
  const int py = tid / TWQ, q = tid - py * TWQ;
  constexpr int KK = K ? K : 1;
  const bool publish = COH && A.merged;                        // merged launch: the sweep workgroups of this sample wait for this tile
  if (py >= c.TH && !publish) return;
  float acc[3][4];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[p][e] = 0.f;
  if (py >= c.TH) {
    // (idle rows of the thread grid: nothing to accumulate; they only take part in the publish barrier)
  } else if (K) {
#pragma unroll 1
    for (int i = 0; i < KK; ++i) {
      const float* row = tg + (py + i) * c.PW + q * 4;
      float r[4 + KK - 1];
#pragma unroll
      for (int t = 0; t < 4 + KK - 1; ++t) r[t] = row[t];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const float* wr = wts + (p * KK + (KK - 1 - i)) * KK;             // flipped kernel row
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
  if (py < c.TH && yg < g.H) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int xg = c.x0 + q * 4 + e;
      if (xg < g.W) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          float* dst = A.s.gplanes + (static_cast<size_t>(c.b) * 3 + p) * g.HW + yg * g.W + xg;
          const float v = bad ? __builtin_nanf("") : acc[p][e];                                                       // NaN: timed-out hand-off
          if (publish) st_agent(dst, v); else *dst = v;
        }
      }
    }
  }
  if (publish) handoff_publish(A.c.sync + A.cflag0 + local);
}

template <int K>
__global__ __launch_bounds__(kBlock) void k_bwd_convT(const Group<BwdArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int lvl = find_level(G, blockIdx.x, local);
  bwd_convT_body<K, false>(G.lv[lvl], local, smem);
}

// k_bwd_reduce1 with the transposed conv folded in (MGACBAM_BWD_FOLD): grid of a level = [tile workgroups][nconv conv tiles LAST].
// The conv tiles start as the streaming workgroups retire and need only the g_pre rows of a few tiles each, so most of the
// latency-bound conv overlaps the streaming tail and one launch boundary disappears.  Flags: one generation counter per k_bwd_reduce1
// tile and per conv tile in ctx.sync (zero-filled by the caller once, never reset: see bwd_convT_body).
template <typename T, int VEC, int K>
__global__ __launch_bounds__(kBlock) BWD_OCC void k_bwd_reduce1_fold(const Group<BwdArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const BwdArgs& A = G.lv[l];
  const int tiles = ((A.g.B + 7) / 8) * 8 * A.nt;
  if (local >= tiles) {
    if (local - tiles < A.nconv) bwd_convT_body<K, true>(A, local - tiles, smem);
    return;
  }
  bwd_reduce1_body<T, VEC, true>(A, local, smem);
}

// ---------------------------------------------------------------------------------------------
// k_bwd_wsa: per conv tile, the partial of dWsa[p,i,j] = sum_px g_pre[px] * planes[p][px + (i,j) - pad].
//   Needs only g_pre and the forward planes and feeds only k_bwd_params, so it is OFF the critical path: a graph /
//   multi-stream caller runs it beside k_bwd_reduce2 (PyramidPlan does).  g_pre and the 3 planes (tile + halo) are
//   staged in LDS in one load batch; work item = (p, i, tile row): it slides a k-wide register window along the row,
//   so each LDS value feeds k FMAs (one g_pre and one plane read per pixel for k outputs); the rows are then summed
//   through LDS.  K == 0 (any odd k): one thread per output.
//   LDS: [g_pre tile][3 plane tiles][items * k partial sums]
// ---------------------------------------------------------------------------------------------
template <int K, bool AGENT = false, bool COH = false>   // AGENT: the partials are summed inside the same launch (k_bwd_apply's tail roles): written through
__device__ __forceinline__ void bwd_wsa_body(const BwdArgs& A, const int local, float* smem) {   // COH: g_pre is published inside this launch (k_bwd_r12)
  const Geo& g = A.g;
  const int k = K ? K : g.k;
  const ConvTile c = conv_tile(g, A.t, k, local, A.t.wsa_th);
  const int tid = threadIdx.x;
  const int plane_elems = c.PH * c.PW;
  float* tg = smem;                            // g_pre tile + halo
  float* tp = tg + plane_elems;                // 3 plane tiles + halo
  float* accs = tp + 3 * plane_elems;          // (3*k*TH) x k row partials
  const float* gpre = A.s.gpre + static_cast<size_t>(c.b) * g.HW;
  const float* pl = A.c.planes + static_cast<size_t>(c.b) * 3 * g.HW;
  bool bad = false;
  if (COH) {                                                    // as the folded transposed conv: own generation counter, bounded wait
    int* own = A.c.sync + A.wflag0 + local;
    const int gen = static_cast<int>(static_cast<unsigned>(ld_agent(own)) + 1u);
    const int TP = A.t.chan_tx * A.vec;
    const int ra = max(c.y0 - c.pad, 0), rb = min(c.y0 + c.TH - 1 + c.pad, g.H - 1);
    bad = handoff_wait(A.c.sync + A.bflag0 + static_cast<size_t>(c.b) * A.nflag, (ra * g.W) / TP, ((rb + 1) * g.W - 1) / TP, gen,
                       A.c.sync + static_cast<size_t>(g.B) * A.nflag, A.spin_limit);
    if (tid == 0) __hip_atomic_fetch_add(own, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  stage_window<8>(tg, 4, c.PH, c.PW, c.y0 - c.pad, c.x0 - c.pad, g,
                   [&](int p, int off) { return p == 0 ? (COH ? ld_agent(gpre + off) : gpre[off]) : pl[static_cast<size_t>(p - 1) * g.HW + off]; });
  __syncthreads();
  constexpr int KK = K ? K : 1;
  const int nout = 3 * k * k;
  if (K) {
    const int nitems = 3 * KK * c.TH;                            // (p, i, row)
    for (int item = tid; item < nitems; item += kBlock) {
      const int pi = item / c.TH, r = item - pi * c.TH;          // pi = p*K + i
      const int p = pi / KK, i = pi - p * KK;
      const float* prow = tp + p * plane_elems + (r + i) * c.PW;           // planes[p][y0 + r + i - pad][x0 - pad + ...]
      const float* grow = tg + (r + c.pad) * c.PW + c.pad;                 // g_pre[y0 + r][x0 + ...]
      float acc[KK];
#pragma unroll
      for (int j = 0; j < KK; ++j) acc[j] = 0.f;
      float w[KK + 3];
#pragma unroll
      for (int t = 0; t < KK - 1; ++t) w[t] = prow[t];
      for (int x = 0; x < c.TW; x += 4) {                        // TW is a multiple of 4; zeros outside the image
        float gq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { gq[e] = grow[x + e]; w[KK - 1 + e] = prow[x + KK - 1 + e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < KK; ++j) acc[j] += gq[e] * w[e + j];
#pragma unroll
        for (int t = 0; t < KK - 1; ++t) w[t] = w[t + 4];
      }
#pragma unroll
      for (int j = 0; j < KK; ++j) accs[item * KK + j] = acc[j];
    }
    __syncthreads();
    for (int o = tid; o < nout; o += kBlock) {                   // o = (p*K + i)*K + j
      const int pi = o / KK, j = o - pi * KK;
      float sum = 0.f;
      for (int r = 0; r < c.TH; ++r) sum += accs[(pi * c.TH + r) * KK + j];
      if (COH && bad) sum = __builtin_nanf("");
      if (AGENT) st_agent(A.s.gwsa_part + static_cast<size_t>(o) * A.nwsa + local, sum);
      else A.s.gwsa_part[static_cast<size_t>(o) * A.nwsa + local] = sum;
    }
  } else {
    for (int o = tid; o < nout; o += kBlock) {
      const int p = o / (k * k), r = o - p * k * k;
      const int i = r / k, j = r - i * k;
      const float* pp = tp + p * plane_elems + i * c.PW + j;
      const float* gg = tg + c.pad * c.PW + c.pad;
      float sum = 0.f;
      for (int yy = 0; yy < c.TH; ++yy)
        for (int xx = 0; xx < c.TW; ++xx) sum += gg[yy * c.PW + xx] * pp[yy * c.PW + xx];
      if (AGENT) st_agent(A.s.gwsa_part + static_cast<size_t>(o) * A.nwsa + local, sum);
      else A.s.gwsa_part[static_cast<size_t>(o) * A.nwsa + local] = sum;
    }
  }
}

template <int K>
__global__ __launch_bounds__(kBlock) void k_bwd_wsa(const Group<BwdArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int lvl = find_level(G, blockIdx.x, local);
  bwd_wsa_body<K>(G.lv[lvl], local, smem);
}

// ---------------------------------------------------------------------------------------------
// k_bwd_reduce2     (thread layout of k_pool: rows of TX lanes sweep H*W for CPT channels each)
// ---------------------------------------------------------------------------------------------
// 16-byte loads that bypass this CU's L1 (sc1: the pair of common.cuh's st_agent for vectors), for data published inside the launch:
// a raw buffer load with the sc1 bit, tracked by the compiler like any other load.  rsrc: wave-uniform base, byte offsets per lane.
// (ROCm 7.2's clang narrows the b128 builtin to ONE dword splatted over the vector when its elements are extracted with a loop index
//  -- seen in the .s and on the device -- so the vector is bit-cast whole and its members are named)
typedef unsigned int v4u32_t __attribute__((ext_vector_type(4)));
typedef float v4f32_t __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ void load_plane_agent(__amdgpu_buffer_rsrc_t rsrc, const int elem, float (&out)[VEC]) {
  if constexpr (VEC % 4 == 0) {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const v4u32_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (elem + 4 * q) * 4, 0, 16);
      const v4f32_t f = __builtin_bit_cast(v4f32_t, v);
      out[4 * q] = f.x; out[4 * q + 1] = f.y; out[4 * q + 2] = f.z; out[4 * q + 3] = f.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (elem + e) * 4, 0, 16));
  }
}

template <typename T, int VEC, int CPT, bool COH = false>   // COH: g_planes and the tile partials are published inside this launch (k_bwd_r12)
__device__ __forceinline__ void bwd_reduce2_body(const BwdArgs& A, const int bid, float* red) {
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
  bool bad = false;
  __amdgpu_buffer_rsrc_t gprs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gp0), 0, 2 * g.HW * 4, 0x00020000);
  if (COH) {
    // this sample's transposed-conv tiles (lower workgroup ids of this launch) publish g_planes; they in turn waited for the sample's
    // k_bwd_reduce1 tiles, so the tile partials read below are out as well.  Own generation counter, bounded wait, NaN on time-out.
    int* own = A.c.sync + A.sflag0 + static_cast<size_t>(b) * g.C + cg;
    const int gen = static_cast<int>(static_cast<unsigned>(ld_agent(own)) + 1u);
    const int cps = A.nconv / g.B;                                // conv tiles per sample (sample-major ids)
    bad = handoff_wait(A.c.sync + A.cflag0, b * cps, b * cps + cps - 1, gen, A.c.sync + static_cast<size_t>(g.B) * A.nflag, A.spin_limit);
    if (tid == 0) __hip_atomic_fetch_add(own, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  constexpr int PF = MGACBAM_POOL_PF;                         // positions per lane per memory round (see k_pool)
  for (int i0 = tx; i0 < nv; i0 += TX * PF) {
    float g0[PF][VEC], g1[PF][VEC], xv[PF][CPT][VEC];
    int ci[PF][VEC];
    bool ok[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int i = i0 + p * TX;
      ok[p] = i < nv;
      const size_t o = static_cast<size_t>(ok[p] ? i : nv - 1) * VEC;
      if (COH) {
        load_plane_agent<VEC>(gprs, static_cast<int>(o), g0[p]);
        load_plane_agent<VEC>(gprs, g.HW + static_cast<int>(o), g1[p]);
      } else {
        load_vec<float, VEC>(gp0 + o, g0[p]);
        load_vec<float, VEC>(gp1 + o, g1[p]);
      }
      load_ivec<VEC>(cidx + o, ci[p]);
#pragma unroll
      for (int j = 0; j < CPT; ++j) load_vec<T, VEC>(xr[j] + o, xv[p][j]);
    }
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      if (!ok[p]) continue;
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float wgt = g1[p][e] * invC + (ci[p][e] == cj[j] ? g0[p][e] : 0.f);   // mean backward + max backward
          acc[j] += xv[p][j][e] * wgt;
        }
      }
    }
  }
  // the hw-tile partials of k_bwd_reduce1 are summed by the same row reduction (lanes stride the tiles)
  float sums[3 * CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const float* part = A.s.A_part + static_cast<size_t>(b) * A.nt * 2 * g.C + cj[j];
    float As = 0.f, Qs = 0.f;
    if (COH) { for (int t = tx; t < A.nt; t += TX) { As += ld_agent(part + static_cast<size_t>(t) * 2 * g.C); Qs += ld_agent(part + static_cast<size_t>(t) * 2 * g.C + g.C); } }
    else { for (int t = tx; t < A.nt; t += TX) { As += part[static_cast<size_t>(t) * 2 * g.C]; Qs += part[static_cast<size_t>(t) * 2 * g.C + g.C]; } }
    sums[j] = acc[j]; sums[CPT + j] = As; sums[2 * CPT + j] = Qs;
  }
  row_sum<3 * CPT>(sums, TX, tid, red);
  float gzv[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) gzv[j] = 0.f;                  // channels past C contribute nothing
  if (tx == 0) {
    const float a = softplusf_(*A.p.beta);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int c = c0 + j;
      if (c < g.C) {
        const size_t o = static_cast<size_t>(b) * g.C + c;
        const float As = sums[CPT + j], Qs = sums[2 * CPT + j];
        const float ca = A.c.ca[o];
        const float gca = a * As + sums[j];
        gzv[j] = (COH && bad) ? __builtin_nanf("") : gca * ca * (1.f - ca);   // NaN: a hand-off that timed out must be loud
        A.s.gz[o] = gzv[j];
        A.s.gbq[o] = Qs;                                       // sum_hw gy*(v - x) for this (b,c)
      }
    }
  }
  // this workgroup's share of the hidden gradient: pgh[b,cg,j] = sum_{c in group} W2[c,j] * g_z[b,c]
  // (k_bwd_apply sums the groups in its prologue, so it does not wait for the parameter-gradient kernel)
  const int h = g.hidden;
  float* s_pg = red + 64;                                      // TY x hc (hidden units are processed in chunks of hc)
  const int hc = min(h, max(1, kPghLds / TY));
  for (int h0 = 0; h0 < h; h0 += hc) {
    const int hn = min(hc, h - h0);
    if (h0) __syncthreads();
    if (tx == 0) {
      for (int jh = 0; jh < hn; ++jh) {
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) p += A.p.w2[static_cast<size_t>(cj[j]) * h + h0 + jh] * gzv[j];
        s_pg[ty * hc + jh] = p;
      }
    }
    __syncthreads();
    for (int jh = tid; jh < hn; jh += kBlock) {
      float p = 0.f;
      for (int r = 0; r < TY; ++r) p += s_pg[r * hc + jh];
      A.s.pgh[(static_cast<size_t>(b) * ncg + cg) * h + h0 + jh] = p;
    }
  }
}

// ROLES: the level's grid is [nwsa dWsa-partial workgroups (k = 7)][streaming workgroups]: the LDS/VALU-bound role
// workgroups are dispatched first and overlap with the HBM-bound ones instead of costing a launch on the critical path.
template <typename T, int VEC, int CPT, bool ROLES>
__global__ __launch_bounds__(kBlock) BWD_OCC void k_bwd_reduce2(const Group<BwdArgs> G) {
  extern __shared__ __align__(16) float smem[];                // [64 reduction scratch][TY * hidden] | wsa tiles
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const BwdArgs& A = G.lv[l];
  if (ROLES) {
    // A.nrole role workgroups share the level's A.nwsa dWsa tiles (tile t -> role t % nrole, one after the other): the host sizes nrole to
    // the slots the streaming workgroups leave idle, so that the roles displace nothing and finish inside the streaming time
    const int npad = (A.nrole + 7) & ~7;                        // keeps the streaming ids' id % 8 <-> sample alignment
    // (tools/trace_r2.py, config 2: the 800 one-tile roles live 8.6 us each and hold 800 of the 1792 slots first, so the last streaming
    // workgroups start at 15 us and a 19 us stream ends at 24.6.  Interleaving roles and streaming workgroups in dispatch order was
    // measured: +2 us -- the roles first is the better of the two orders; the fix is fewer slot-microseconds of role work.)
    if (local < npad) {
      if (local < A.nrole) {
        TRACE_MARK(A.trace, blockIdx.x, 0);
        for (int t = local; t < A.nwsa; t += A.nrole) { bwd_wsa_body<7>(A, t, smem); __syncthreads(); }
#ifdef MGACBAM_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRACE_MARK(A.trace, blockIdx.x, 5);
#endif
      }
      return;
    }
    local -= npad;
  }
  TRACE_MARK(A.trace, blockIdx.x, 0);
  bwd_reduce2_body<T, VEC, CPT>(A, local, smem);
#ifdef MGACBAM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TRACE_MARK(A.trace, blockIdx.x, 10);
#endif
}

// ---------------------------------------------------------------------------------------------
// k_bwd_r12: k_bwd_reduce1 tiles, transposed-conv tiles, dWsa tiles and k_bwd_reduce2 sweeps as ONE launch (spatial kernel 7, whole
//   backward in one call on a zero-filled ctx.sync).  Workgroup ids are PHASE-major over the levels of the group --
//   [all tiles][all conv tiles][all dWsa tiles][all sweeps] -- and every wait points at lower ids only: conv and dWsa tiles wait for the
//   k_bwd_reduce1 tiles whose rows they read (generation counters, as in k_bwd_reduce1_fold), a sweep waits for the conv tiles of its
//   sample.  Producers never wait on higher ids, so progress does not depend on residency; every wait is bounded and poisons on
//   time-out.  What the merge buys is the boundary between the two launches and their ramp / tail: the sweeps of the first samples
//   start while the last tiles still stream.  All four bodies fit 64 VGPRs (occupancy 7-8): k_bwd_apply (104) stays a launch of its own
//   -- capped at its 4 waves per SIMD these two kernels lose 9 us at config 2.
//   The merged launch has generation counters of its own for every class (args.cuh): the fold form may run on the same ctx in between.
// ---------------------------------------------------------------------------------------------
struct R12Group {
  Group<BwdArgs> g;
  int seg[4][kGroupMax + 1];     // seg[p][l]: first workgroup id of level l in phase p; seg[p][n]: end of phase p
};
template <typename T, int VEC, int CPT>
__global__ __launch_bounds__(kBlock) BWD_OCC void k_bwd_r12(const R12Group R) {
  extern __shared__ __align__(16) float smem[];
  const int bid = blockIdx.x;
  int p = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (bid >= R.seg[q][0]) p = q;
  int l = 0;
#pragma unroll
  for (int i = 1; i < kGroupMax; ++i)
    if (i < R.g.n && bid >= R.seg[p][i]) l = i;
  const int local = bid - R.seg[p][l];
  const BwdArgs& A = R.g.lv[l];
#ifdef MGACBAM_TRACE
  TRACE_MARK(A.trace, blockIdx.x, 0);                           // tools/trace_r12.py: start, phase (slot 14), end
  if (A.trace && threadIdx.x == 0) A.trace[static_cast<size_t>(blockIdx.x) * 16 + 14] = p + 1;
#endif
  if (p == 0) bwd_reduce1_body<T, VEC, true>(A, local, smem);
  else if (p == 1) { if (local < A.nconv) bwd_convT_body<7, true>(A, local, smem); }
  else if (p == 2) { if (local < A.nwsa) bwd_wsa_body<7, false, true>(A, local, smem); }
  else bwd_reduce2_body<T, VEC, CPT, true>(A, local, smem);
#ifdef MGACBAM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TRACE_MARK(A.trace, blockIdx.x, 10);
#endif
}

// ---------------------------------------------------------------------------------------------
// k_bwd_params: every parameter gradient + the relu-masked hidden gradients, one launch.  Workgroup roles in a level:
//   [0, h)            hidden unit j: g_h[b,j] = sum_c W2[c,j] g_z[b,c] for all b (masked by relu for the avg / mx
//                     application) -> gh_avg, gh_mx ; dW1[j,:], db1[j], dW2[:,j]          (shared MLP used twice)
//   [h, h+nb2)        db2[c] = 2 sum_b g_z[b,c]
//   [.., +nb_wsa)     dWsa[p,i,j] = sum over conv tiles of the partials (one wave per output)
//   last              dbeta = sigmoid(beta) * sum_{b,c} D
//   LDS (dynamic): 3*B floats
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bwd_params_body(const BwdArgs& A, const int local, float* sm, float* red) {
  const Geo& g = A.g;
  const int C = g.C, h = g.hidden, B = g.B, kk3 = 3 * g.k * g.k;
  const int nb2 = (C + kBlock - 1) / kBlock, nb_wsa = (kk3 + 3) / 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (local < h) {
    const int j = local;
    float* s_ga = sm; float* s_gm = sm + B; float* s_hs = sm + 2 * B;   // 3*B
    float* s_part = sm + 3 * B;                                         // 2 * kBlock (b-group partials of dW1, dW2)
    // phase A: g_h[b,j] for every sample (one wave per sample, 4 samples in flight per workgroup)
#pragma unroll 2
    for (int b = wave; b < B; b += kBlock / kWave) {
      const float* gz = A.s.gz + static_cast<size_t>(b) * C;
      float d = 0.f;
#pragma unroll 4
      for (int c = lane; c < C; c += kWave) d += A.p.w2[static_cast<size_t>(c) * h + j] * gz[c];
      d = wave_group_sum(d, kWave);
      if (lane == 0) {
        const float ha = A.c.h_avg[b * h + j], hm = A.c.h_mx[b * h + j];
        const float ga = ha > 0.f ? d : 0.f, gm = hm > 0.f ? d : 0.f;          // relu backward
        s_ga[b] = ga; s_gm[b] = gm; s_hs[b] = ha + hm;
        A.s.gh_avg[b * h + j] = ga;
        A.s.gh_mx[b * h + j] = gm;
      }
    }
    __syncthreads();
    // phase B: dW1[j,c] and dW2[c,j]: thread = (channel, sample group); groups combined through LDS in fixed order
    const int cp = (C >= kBlock) ? kBlock : C;                   // channels handled per pass
    const int ng = kBlock / cp;                                  // sample groups (1 when C >= 256)
    const int cl = tid % cp, bg = tid / cp;
    for (int c0 = 0; c0 < C; c0 += cp) {
      const int c = c0 + cl;
      float w1 = 0.f, w2 = 0.f;
      if (c < C && bg < ng) {
#pragma unroll 4
        for (int b = bg; b < B; b += ng) {
          const size_t o = static_cast<size_t>(b) * C + c;
          w1 += s_ga[b] * A.c.avg[o] + s_gm[b] * A.c.mx[o];                    // dW1[j,c]
          w2 += A.s.gz[o] * s_hs[b];                                           // dW2[c,j]
        }
      }
      if (ng > 1) {
        __syncthreads();
        s_part[tid] = w1; s_part[kBlock + tid] = w2;
        __syncthreads();
        if (bg == 0 && c < C) {
          for (int q = 1; q < ng; ++q) { w1 += s_part[q * cp + cl]; w2 += s_part[kBlock + q * cp + cl]; }
        }
      }
      if (bg == 0 && c < C) {
        A.gw1[static_cast<size_t>(j) * C + c] = w1;
        A.gw2[static_cast<size_t>(c) * h + j] = w2;
      }
    }
    if (tid == 0) {
      float acc = 0.f;
      for (int b = 0; b < B; ++b) acc += s_ga[b] + s_gm[b];
      A.gb1[j] = acc;
    }
    return;
  }
  if (local < h + nb2) {                                         // db2[c] = 2 * sum_b g_z[b,c]   (bias used twice)
    const int c = (local - h) * kBlock + tid;
    if (c >= C) return;
    float acc = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) acc += A.s.gz[static_cast<size_t>(b) * C + c];
    A.gb2[c] = 2.f * acc;
    return;
  }
  if (local < h + nb2 + nb_wsa) {                                // dWsa: lanes stride the tile partials
    if (A.wsa_tail) return;                                      // (summed by the last-arriving tail role of this launch instead)
    const int o = (local - h - nb2) * 4 + wave;
    if (o >= kk3) return;
    const float* part = A.s.gwsa_part + static_cast<size_t>(o) * A.nwsa;
    float acc = 0.f;
#pragma unroll 4
    for (int t = lane; t < A.nwsa; t += kWave) acc += part[t];
    acc = wave_group_sum(acc, kWave);
    if (lane == 0) A.gwsa[o] = acc;
    return;
  }
  {                                                              // dbeta
    float acc = 0.f;
#pragma unroll 8
    for (int o = tid; o < B * C; o += kBlock) acc += A.s.gbq[o];
    acc = block_sum(acc, tid, red);
    if (tid == 0) *A.gbeta = sigmoidf_(*A.p.beta) * acc;
  }
}

__global__ __launch_bounds__(kBlock) void k_bwd_params(const Group<BwdArgs> G) {
  extern __shared__ __align__(16) float sm[];
  __shared__ float red[8];
  int local;
  const int lvl = find_level(G, blockIdx.x, local);
  bwd_params_body(G.lv[lvl], local, sm, red);
}

// ---------------------------------------------------------------------------------------------
// k_bwd_apply       (thread layout of k_chan)
//   gx = gy*((1-a) + a*sa*ca) + ca*([c==cidx]*gp0 + gp1/C) + g_avg*wA + [hw==amax]*g_mx_pt + g_mx_uni
//   gmask = (gp2 + (use/den) * (sum_c g_avg*x - K_b)) * s(1-s)
//   Prologue (per workgroup, for its sample): g_avg = W1^T gh_avg, g_mx = W1^T gh_mx -> q[c] = {ca, g_avg, g_mx routed to
//   the arg-max position, g_mx/N for the GAP fallback} in LDS, and K_b = sum_c g_avg * mavg * [S >= eps].
//   LDS: [C float4 q][2*hidden][256*VEC combine]
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, bool GMASK>
__device__ __forceinline__ void bwd_apply_body(const BwdArgs& A, const int bid, float* smem, float* red) {
  constexpr int UN = MGACBAM_BAPPLY_UN;
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.chan_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int nv = g.HW / VEC;
  const int ntile = (nv + TX - 1) / TX;
  int b, tile;
  if (!xcd_sample_part(bid, g.B, ntile, b, tile)) return;
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

  const int gid = blockIdx.x;
  TRACE_HWID(A.trace, gid);
  TRACE_MARK(A.trace, gid, 0);
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
  const int* amax = A.c.amax + static_cast<size_t>(b) * g.C;
  float accp[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) accp[e] = 0.f;

  // With the W1-projection planes saved by k_chan, sum_c g_avg[b,c]*x[b,c,hw] = sum_j g_h[b,j]*P[b,j,hw]: x is not read.
  const bool need_x = GMASK && g.proj_h == 0;
  // first feature vectors are requested before the prologue so its latency overlaps theirs
  float g0v[UN][VEC], x0v[UN][VEC];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const size_t co = static_cast<size_t>(min(ty + u * TY, g.C - 1)) * g.HW;
    load_vec<T, VEC>(gp + co, g0v[u]);
    if (GMASK) { if (need_x) load_vec<T, VEC>(xp + co, x0v[u]); }
  }

  // ---- prologue: q[c] and K_b for this sample -------------------------------------------------------------------
  float4* s_q = reinterpret_cast<float4*>(smem);
  int* s_am = reinterpret_cast<int*>(smem + 4 * g.C);          // arg-max position of (b,c), -1 when the GAP fallback was used
  float* s_gh = smem + 5 * g.C;
  float* sm = s_gh + 2 * g.hidden;
  const int h = g.hidden;
  const float live = (has_mask && A.c.S[b] >= g.eps) ? 1.f : 0.f;   // clamp_min passes grad only when not clamped
  float kpart = 0.f;
  // common sizes (one channel per thread, hidden <= 16): the per-channel operands of the second prologue phase do not depend
  // on the first, so they are requested now; 20 registers, live only until the main loop starts
  constexpr int kEarlyH = MGACBAM_EARLY_H;
  const bool early = g.C <= kBlock && h <= kEarlyH;
  float rw1[kEarlyH], rca = 0.f, rmavg = 0.f;
  int rvalid = 0, ramax = 0;
  if (early) {
    const int c = min(tid, g.C - 1);
    const size_t o = static_cast<size_t>(b) * g.C + c;
    rca = A.c.ca[o]; rmavg = A.c.mavg[o]; rvalid = A.c.valid[o]; ramax = amax[c];
#pragma unroll
    for (int j = 0; j < kEarlyH; ++j) rw1[j] = A.p.w1[static_cast<size_t>(min(j, h - 1)) * g.C + c];
  }
  {
    // hidden gradient of this sample: g_h[j] = sum over channel groups of k_bwd_reduce2's partials (fixed order),
    // relu-masked for the two applications of the shared MLP
    const float* pg = A.s.pgh + static_cast<size_t>(b) * A.ncg * h;
    const int hh = h < kBlock ? h : kBlock;
    const int G_ = kBlock / hh;                                  // threads per hidden unit
    const int j0 = tid % hh, k0 = tid / hh;
    for (int jb = 0; jb < h; jb += hh) {                         // one pass unless hidden > 256
      const int j = jb + j0;
      float p = 0.f;
      if (k0 < G_ && j < h)
        for (int cg = k0; cg < A.ncg; cg += G_) p += pg[cg * h + j];
      sm[tid] = p;
      __syncthreads();
      if (k0 == 0 && j < h) {
        for (int q = 1; q < G_; ++q) p += sm[q * hh + j0];
        s_gh[j] = A.c.h_avg[static_cast<size_t>(b) * h + j] > 0.f ? p : 0.f;
        s_gh[h + j] = A.c.h_mx[static_cast<size_t>(b) * h + j] > 0.f ? p : 0.f;
      }
      __syncthreads();
    }
  }
  TRACE_MARK(A.trace, gid, 1);                                 // hidden gradient in LDS
  if (early) {
    // per-channel operands were requested before the partial-sum barriers (one global-load latency instead of two)
    if (tid < g.C) {
      float ga = 0.f, gm = 0.f;
#pragma unroll
      for (int j = 0; j < kEarlyH; ++j)
        if (j < h) { ga += rw1[j] * s_gh[j]; gm += rw1[j] * s_gh[h + j]; }
      float4 q;
      q.x = rca;
      q.y = ga;                               // g_avg
      q.z = rvalid ? gm : 0.f;                // routed to the arg-max position
      q.w = rvalid ? 0.f : gm / N;            // GAP fallback: spread uniformly
      s_q[tid] = q;
      s_am[tid] = rvalid ? ramax : -1;
      kpart += ga * rmavg * live;
    }
  } else {
    for (int c = tid; c < g.C; c += kBlock) {
      float ga = 0.f, gm = 0.f;
      for (int j = 0; j < h; ++j) { const float wv = A.p.w1[static_cast<size_t>(j) * g.C + c]; ga += wv * s_gh[j]; gm += wv * s_gh[h + j]; }
      const size_t o = static_cast<size_t>(b) * g.C + c;
      const int valid = A.c.valid[o];
      float4 q;
      q.x = A.c.ca[o];
      q.y = ga;
      q.z = valid ? gm : 0.f;
      q.w = valid ? 0.f : gm / N;
      s_q[c] = q;
      s_am[c] = valid ? amax[c] : -1;
      kpart += ga * A.c.mavg[o] * live;
    }
  }

  kpart = block_sum(kpart, tid, red);       // (contains the barriers that publish s_q)
  if (tid == 0) red[7] = kpart;
  __syncthreads();
  const float kb = red[7];
  TRACE_MARK(A.trace, gid, 2);                                 // per-channel terms in LDS

  auto emit = [&](const float (&gv)[VEC], const float (&xv)[VEC], int c) {
    float ov[VEC];
    const float4 q = s_q[c];
    const int am = s_am[c] - ii * VEC;                        // offset of the arg-max inside this vector, if any
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float r = gv[e] * ((1.f - a) + sav[e] * q.x);
      r += q.x * ((ci[e] == c ? g0[e] : 0.f) + g1[e]);
      r += q.y * wA[e] + q.w;
      r += (am == e) ? q.z : 0.f;
      ov[e] = r;
      if (GMASK) { if (need_x) accp[e] += q.y * xv[e]; }
    }
    if (active) store_vec_stream<T, VEC>(op + static_cast<size_t>(c) * g.HW, ov, A.t.nt_stores);
  };
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const int c = ty + u * TY;
    if (c < g.C) emit(g0v[u], x0v[u], c);
  }
#pragma unroll MGACBAM_BAPPLY_UNROLL
  for (int c = ty + UN * TY; c < g.C; c += TY) {
    float gv[VEC], xv[VEC];
    load_vec<T, VEC>(gp + static_cast<size_t>(c) * g.HW, gv);
    if (GMASK) { if (need_x) load_vec<T, VEC>(xp + static_cast<size_t>(c) * g.HW, xv); }
    emit(gv, xv, c);
  }
  TRACE_MARK(A.trace, gid, 3);                                 // streaming loop done (stores issued)
  if (GMASK) {
    if (need_x) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) sm[tid * VEC + e] = accp[e];
      __syncthreads();
    }
    if (ty == 0 && active) {
      if (need_x) {
        for (int r = 1; r < TY; ++r) {
          const int o = (r * TX + tx) * VEC;
#pragma unroll
          for (int e = 0; e < VEC; ++e) accp[e] += sm[o + e];
        }
      } else {
        for (int j = 0; j < g.proj_h; ++j) {                   // sum_j g_h_avg[b,j] * P[b,j,hw]
          float pv[VEC];
          load_vec<float, VEC>(A.c.proj + (static_cast<size_t>(b) * g.proj_h + j) * g.HW + static_cast<size_t>(i) * VEC, pv);
          const float gh = s_gh[j];
#pragma unroll
          for (int e = 0; e < VEC; ++e) accp[e] += gh * pv[e];
        }
      }
      float g2[VEC], gm[VEC];
      load_vec<float, VEC>(A.s.gplanes + (static_cast<size_t>(b) * 3 + 2) * g.HW + static_cast<size_t>(i) * VEC, g2);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float gs = g2[e] + (use / den) * (accp[e] - kb);
        gm[e] = g.use_sigmoid ? gs * sv[e] * (1.f - sv[e]) : gs;
      }
      store_vec<float, VEC>(A.gmask + static_cast<size_t>(b) * g.HW + static_cast<size_t>(i) * VEC, gm);
    }
  }
#ifdef MGACBAM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TRACE_MARK(A.trace, gid, 10);                                // stores complete
#endif
}

// ROLES: the level's grid is [npg parameter-gradient workgroups][streaming workgroups]; the tiny, latency-bound
// parameter kernel hides behind the largest kernel of the step instead of standing alone on the critical path.
// (amdgpu_waves_per_eu(5) -- 94 instead of 104 VGPRs, 5 instead of 4 workgroups per CU -- was measured: config 2 fp32 57.9 -> 55.8 us,
//  bf16 37.3 -> 38.5, config 4 200 -> 225 us; not adopted.  tools/trace_gate.py bwd shows the launch as two rounds of
//  ~5 us prologue + ~19 us streaming per workgroup with HBM saturated during the streaming.)
// dWsa as TAIL roles of the k_bwd_apply launch (A.wsa_tail): the level's grid is [params roles][streaming][nwsa tile partials][nsum sums].
// As leading roles of k_bwd_reduce2 the 800 one-tile workgroups (8.6 us each at config 2) held 800 of its 1792 slots first and cost that
// launch 5.5-7 us (tools/trace_r2.py).  Dispatched behind the last streaming workgroup of the step's LONGEST launch they start when its
// last resident round does and run in the slots that round leaves free.  dWsa needs g_pre and the forward planes only (complete since
// k_bwd_reduce1).  Hand-off: every tile role adds to an arrival counter (status word 1 of ctx.sync) once its partials are out; the sum
// roles (4 outputs each, the fixed order of k_bwd_params: bitwise reproducible) are the very last workgroups, wait for nwsa arrivals --
// every producer has a lower id and never waits, so progress does not depend on residency; the wait is bounded like every other -- and
// the last of them to finish (word 2) hands both words back as 0.
// MEASURED (config 2, round 3) and therefore OPT-IN (MGACBAM_WSA_TAIL=1): k_bwd_reduce2 27.1 -> 22.2 us, k_bwd_apply 58.8 -> 74.3 us,
// step 200.5 -> 207.2 us.  The 800 tile roles are ~6,900 slot-microseconds of latency-bound work wherever they run: the last resident
// round of k_bwd_apply leaves 224 of 1024 slots free, not enough to absorb them.  (ONE last-arriving workgroup summing all partials was
// measured first: 470 KB through one CU, +31 us.)
__device__ __forceinline__ void bwd_wsa_tail(const BwdArgs& A, const int t, float* smem) {
  bwd_wsa_body<7, true>(A, t, smem);
  int* cnt = A.c.sync + static_cast<size_t>(A.g.B) * A.nflag + 1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this workgroup's partials are out (every wave waits before the barrier)
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void bwd_wsa_sum(const BwdArgs& A, const int s, const int nsum) {
  int* words = A.c.sync + static_cast<size_t>(A.g.B) * A.nflag;  // [0] time-out status, [1] arrivals, [2] finished sum roles
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kk3 = 3 * A.g.k * A.g.k;
  int timed_out = 0;
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (ld_agent(words + 1) < A.nwsa) {
      __builtin_amdgcn_s_sleep(16);
      if (++spins > A.spin_limit) { st_agent(words, 1); timed_out = 1; break; }
    }
  }
  const bool bad = __syncthreads_or(timed_out) != 0;
  const int o = s * (kBlock / kWave) + wave;
  if (o < kk3) {
    const float* part = A.s.gwsa_part + static_cast<size_t>(o) * A.nwsa;
    float acc = 0.f;
#pragma unroll 4
    for (int i = lane; i < A.nwsa; i += kWave) acc += ld_agent(part + i);
    acc = wave_group_sum(acc, kWave);
    if (lane == 0) A.gwsa[o] = bad ? __builtin_nanf("") : acc;   // NaN: a hand-off that timed out must be loud
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && __hip_atomic_fetch_add(words + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsum - 1) {
    st_agent(words + 1, 0);                                      // every sum role has seen the full count: the next call starts from 0
    st_agent(words + 2, 0);
  }
}

template <typename T, int VEC, bool GMASK, bool ROLES>
__global__ __launch_bounds__(kBlock) void k_bwd_apply(const Group<BwdArgs> G) {
  extern __shared__ __align__(16) float smem[];
  __shared__ float red[8];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const BwdArgs& A = G.lv[l];
  if (ROLES) {
    const int npad = (A.npg + 7) & ~7;
    if (local < npad) {
      if (local < A.npg) {
        TRACE_MARK(A.trace, blockIdx.x, 0);
        bwd_params_body(A, local, smem, red);
#ifdef MGACBAM_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        TRACE_MARK(A.trace, blockIdx.x, 9);
#endif
      }
      return;
    }
    local -= npad;
    if (A.wsa_tail) {
      const int tiles = ((A.g.B + 7) / 8) * 8 * A.nt;
      if (local >= tiles) {
        const int nsum = (3 * A.g.k * A.g.k + 3) / 4;
        if (local - tiles < A.nwsa) bwd_wsa_tail(A, local - tiles, smem);
        else if (local - tiles - A.nwsa < nsum) bwd_wsa_sum(A, local - tiles - A.nwsa, nsum);
        return;
      }
    }
  }
  bwd_apply_body<T, VEC, GMASK>(A, local, smem, red);
}

}  // namespace mgacbam
