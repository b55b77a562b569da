// Forward kernels of the mask-guided CBAM block (reference mga_yolo/nn/modules/masked_cbam.py:87-171).
//
//   k_pool   x (1 read)         -> avg, mx (+ arg-max), S/use/den, sigma(mask) plane                     [HBM-bound]
//   k_gate   x (1 read, the tile stays in registers) -> role workgroups: shared MLP -> ca ; tiles: planes[max_c, mean_c], cidx,
//                                  in-launch hand-off of the plane rows, k x k conv -> sa, y (1 write)      [MGACBAM_FWD_FUSE]
//   fallback (shapes k_gate does not take, or without the flag):
//   k_chan   x (1 read)         -> prologue: shared MLP -> ca ; body: planes[max_c, mean_c], cidx         [HBM-bound]
//   k_apply  x (1 read), planes -> prologue: k x k conv of the tile -> sa ; body: y (1 write)             [HBM-bound]
//
// Two (fallback: three) launches, each covering P3+P4+P5.  The two tiny steps of the block (the MLP: <= 74k MAC per sample;
// the conv: 147 MAC per pixel on 3 planes) never get launches of their own: a dependent tiny launch costs >= 5 us on MI355X
// (2.5 us boundary + a first load that always misses, because the producer's lines sit in another XCD's L2) whatever its
// arithmetic, while a prologue / role workgroup costs one such miss, overlapped with the first feature loads.
//
// No kernel materialises cam_out / sam_out / the expanded mask (the reference makes ~15 full-size
// temporaries).  Thread layout everywhere: 256 threads = TY rows x TX lanes, TX lanes run along H*W
// (contiguous, VEC elements per lane per access => 16-byte coalesced accesses for fp32).
#pragma once
#include "args.cuh"
#include "common.cuh"
#include "tile.cuh"

namespace mgacbam {

// ---------------------------------------------------------------------------------------------
// k_pool: masked average / masked max pooling over H*W for every (b, c)      masked_cbam.py:87-121
//   workgroup = (sample b, CPB = TY*CPT channels); each row of TX lanes sweeps the whole H*W extent of its
//   CPT channels, so sigma(mask) is evaluated once per position and reused from registers for CPT
//   channels; the per-channel sums and the (max, first arg-max) pair are then reduced with wave
//   shuffles (+ one LDS step when a row spans several waves).
// ---------------------------------------------------------------------------------------------
// The selector of the masked max is `sigmoid(mask) > 0.5` on fp32 values (masked_cbam.py:93,116).  With torch's sigmoid,
// 1 / (1 + exp(-m)), that is true exactly for logits m > 1.5 * 2^-24: exp(-m) rounds to 1 - k 2^-24 with k = RN(m 2^24), 1 + that rounds
// (ties to even) to 2 - 2^-23 or less iff k >= 2, and only then does the quotient exceed 0.5 + 2^-25, the half-way point above 0.5; at
// m = 1.5 * 2^-24 itself exp(-m) lies just above the tie and k = 1.  Checked against torch's CPU kernel for every float near the
// threshold and every 5th float in [2^-30, 2^-10] (vector and scalar-tail paths); pinned by tests/golden/case_boundary_logits.npz.
// Comparing the LOGIT keeps valid / amax bit-exact where a fast sigmoid (v_exp + v_rcp, ~1e-6) could flip a pixel within ~1e-6 of 0.
constexpr float kSelLogit = 0x1.8p-24f;

template <typename T, int VEC, int CPT, bool HAS_MASK>
__device__ __forceinline__ void pool_body(const FwdArgs& A, const int bid, float* red) {
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
  TRACE_HWID(A.trace, blockIdx.x);
  TRACE_MARK(A.trace, blockIdx.x, 0);

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

  // PF positions per lane are requested per round: a lane's sweep is a chain of dependent memory rounds (each ~2.5 us
  // under load), so the kernel cannot finish before rounds x latency -- fewer, fatter rounds (profiles/ notes, membw).
  constexpr int PF = MGACBAM_POOL_PF;
  // (A staggered start -- every workgroup beginning its sweep at a different round, to spread concurrent requests over the memory
  //  channels -- was measured at configs 2, 3, 4: no gain, k_bwd_reduce2 10 % slower from the extra index arithmetic; not kept.)
  for (int i0 = tx; i0 < nv; i0 += TX * PF) {
    float m[PF][VEC], xv[PF][CPT][VEC];
    bool ok[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int i = i0 + p * TX;
      ok[p] = i < nv;
      const size_t o = static_cast<size_t>(ok[p] ? i : nv - 1) * VEC;
      if (HAS_MASK) load_vec<float, VEC>(mb + o, m[p]);
#pragma unroll
      for (int j = 0; j < CPT; ++j) load_vec<T, VEC>(xr[j] + o, xv[p][j]);
    }
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      if (!ok[p]) continue;
      const int i = i0 + p * TX;
      float s[VEC];
      bool sel[VEC];
      if (HAS_MASK) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          s[e] = g.use_sigmoid ? sigmoid_fast(m[p][e]) : m[p][e];   // masked_cbam.py:93-94
          sel[e] = g.use_sigmoid ? m[p][e] > kSelLogit : s[e] > 0.5f;   // masked_cbam.py:116, decided EXACTLY (kSelLogit below)
          ssum += s[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s[e] = 0.f; sel[e] = true; }   // masked_cbam.py:138 (zero plane)
      }
      if (writes_plane) store_vec<float, VEC>(splane + static_cast<size_t>(i) * VEC, s);
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float v = xv[p][j][e];
          sx[j] += v;
          if (HAS_MASK) sxs[j] += v * s[e];
          if (sel[e] && v > vmax[j]) { vmax[j] = v; imax[j] = i * VEC + e; }   // strict > : first max wins
        }
      }
    }
  }

  TRACE_MARK(A.trace, blockIdx.x, 3);                          // sweep done
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
#ifdef MGACBAM_TRACE
  TRACE_MARK(A.trace, blockIdx.x, 8);                          // reductions done, stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TRACE_MARK(A.trace, blockIdx.x, 10);
#endif
}

template <typename T, int VEC, int CPT, bool HAS_MASK>
__global__ __launch_bounds__(kBlock) void k_pool(const Group<FwdArgs> G) {
  __shared__ float red[64];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  pool_body<T, VEC, CPT, HAS_MASK>(G.lv[l], local, red);
}

// ---------------------------------------------------------------------------------------------
// shared MLP + channel gate for ONE sample, all channels, result in LDS           masked_cbam.py:54-58, 128-129
//   s_in: 2*C floats scratch, s_h: 2*hidden, s_ca: C (output).  `publish` => also store ca, h_avg, h_mx of this
//   sample to ctx (done by one workgroup per sample; backward and k_apply read them).
//   (Holding W1/W2 slices in registers to issue all loads before the first barrier was tried: it cost ~40 VGPRs,
//   dropped the streaming loop from 7 to 3 waves/SIMD and made k_chan 40 % slower -- profiles/r01 notes.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void mlp_gate_to_lds(const FwdArgs& A, const int b, const bool publish,
                                                float* s_in, float* s_h, float* s_ca) {
  const Geo& g = A.g;
  const int tid = threadIdx.x, C = g.C, h = g.hidden;
  const int wave = tid >> 6, lane = tid & 63;
  float* s_avg = s_in;
  float* s_mx = s_in + C;
  for (int c = tid; c < C; c += kBlock) {
    s_avg[c] = A.c.avg[static_cast<size_t>(b) * C + c];
    s_mx[c] = A.c.mx[static_cast<size_t>(b) * C + c];
  }
  __syncthreads();
  for (int j = wave; j < h; j += kBlock / kWave) {
    const float* w = A.p.w1 + static_cast<size_t>(j) * C;
    float da = 0.f, dm = 0.f;
    for (int c = lane; c < C; c += kWave) { const float wv = w[c]; da += wv * s_avg[c]; dm += wv * s_mx[c]; }
    da = wave_group_sum(da, kWave);
    dm = wave_group_sum(dm, kWave);
    if (lane == 0) {
      const float bj = A.p.b1[j];
      const float ha = fmaxf(da + bj, 0.f), hm = fmaxf(dm + bj, 0.f);
      s_h[j] = ha; s_h[h + j] = hm;
      if (publish) {
        A.c.h_avg[static_cast<size_t>(b) * h + j] = ha;
        A.c.h_mx[static_cast<size_t>(b) * h + j] = hm;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += kBlock) {
    const float* w = A.p.w2 + static_cast<size_t>(c) * h;
    float za = 0.f, zm = 0.f;
    for (int j = 0; j < h; ++j) { const float wv = w[j]; za += wv * s_h[j]; zm += wv * s_h[h + j]; }
    const float bc = A.p.b2[c];
    const float z = (za + bc) + (zm + bc);
    const float ca = sigmoidf_(z);
    s_ca[c] = ca;
    if (publish) A.c.ca[static_cast<size_t>(b) * C + c] = ca;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// k_chan: u = x * ca ; per pixel max_c u (+ first arg-max channel) and mean_c u        masked_cbam.py:130,135-136
//   workgroup = (sample b, tile of TX vectors along H*W); row ty handles channels ty, ty+TY, ...; the TY partial
//   (max, idx, sum) triples are combined through LDS.  Prologue: the sample's MLP (above); the first UN feature
//   vectors of every lane are requested BEFORE it so the two latencies overlap.
//   PROJ (training steps that will want dL/dmask): while x streams through, also accumulate the W1-projection planes
//   P[b,j,hw] = sum_c W1[j,c] x[b,c,hw] (hidden <= kProjMax), so that k_bwd_apply never has to read x (see bwd.cuh).
//   LDS: [2C scratch][2h][C ca][C*kProjMax W1^T (PROJ)][4*256*VEC combine]
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, bool PROJ, bool NOPRO = false>   // NOPRO: ca comes from a preceding k_mlp launch
__device__ __forceinline__ void chan_body(const FwdArgs& A, const int bid, float* smem) {
  constexpr int UN = 4;
  constexpr int HP = kProjMax;
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.chanf_tx, lt = ilog2(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = kBlock >> lt;
  const int nv = g.HW / VEC;
  const int ntile = (nv + TX - 1) / TX;
  int b, tile;
  if (!xcd_sample_part(bid, g.B, ntile, b, tile)) return;
  const int i = tile * TX + tx;
  const bool active = i < nv;
  const int ii = active ? i : nv - 1;
  const T* xp = static_cast<const T*>(A.x) + static_cast<size_t>(b) * g.C * g.HW + static_cast<size_t>(ii) * VEC;

  float x0[UN][VEC];
#pragma unroll
  for (int u = 0; u < UN; ++u) load_vec<T, VEC>(xp + static_cast<size_t>(min(ty + u * TY, g.C - 1)) * g.HW, x0[u]);

  float* s_in = smem;
  float* s_h = smem + 2 * g.C;
  float* s_ca = s_h + 2 * g.hidden;
  float* s_w1t = s_ca + g.C;                                   // [c][HP], zero padded past hidden
  const bool proj = PROJ && g.proj_h > 0;
  float* sm = s_w1t + (PROJ ? g.C * HP : 0);
  if (proj) {
    for (int idx = tid; idx < g.C * HP; idx += kBlock) {
      const int c = idx / HP, j = idx - c * HP;
      s_w1t[idx] = j < g.proj_h ? A.p.w1[static_cast<size_t>(j) * g.C + c] : 0.f;
    }
  }
  if (NOPRO) {
    for (int c = tid; c < g.C; c += kBlock) s_ca[c] = A.c.ca[static_cast<size_t>(b) * g.C + c];
    __syncthreads();                                           // (also publishes s_w1t)
  } else {
    mlp_gate_to_lds(A, b, tile == 0, s_in, s_h, s_ca);         // (its barriers also publish s_w1t)
  }

  float vmax[VEC], vsum[VEC];
  int vidx[VEC];
  float accP[PROJ ? HP : 1][VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { vmax[e] = -INFINITY; vsum[e] = 0.f; vidx[e] = ty; }
#pragma unroll
  for (int j = 0; j < (PROJ ? HP : 1); ++j)
#pragma unroll
    for (int e = 0; e < VEC; ++e) accP[j][e] = 0.f;
  auto consume = [&](const float (&xv)[VEC], int c) {
    const float cac = s_ca[c];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float uu = xv[e] * cac;                          // masked_cbam.py:130
      vsum[e] += uu;
      if (uu > vmax[e]) { vmax[e] = uu; vidx[e] = c; }
    }
    if (PROJ) {
      if (proj) {
        float w[HP];
#pragma unroll
        for (int q4 = 0; q4 < HP / 4; ++q4) {
          const float4 wv = *reinterpret_cast<const float4*>(s_w1t + c * HP + 4 * q4);
          w[4 * q4] = wv.x; w[4 * q4 + 1] = wv.y; w[4 * q4 + 2] = wv.z; w[4 * q4 + 3] = wv.w;
        }
#pragma unroll
        for (int j = 0; j < HP; ++j)
#pragma unroll
          for (int e = 0; e < VEC; ++e) accP[j][e] += w[j] * xv[e];
      }
    }
  };
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const int c = ty + u * TY;
    if (c < g.C) consume(x0[u], c);
  }
#pragma unroll 4
  for (int c = ty + UN * TY; c < g.C; c += TY) {
    float xv[VEC];
    load_vec<T, VEC>(xp + static_cast<size_t>(c) * g.HW, xv);
    consume(xv, c);
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
  if (PROJ) {
    if (proj) {                                                // TY partials of the projection planes, 4 planes per LDS round
#pragma unroll
      for (int half = 0; half < HP / 4; ++half) {
        if (half * 4 >= g.proj_h) break;
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int e = 0; e < VEC; ++e) sm[(jj * kBlock + tid) * VEC + e] = accP[half * 4 + jj][e];
        __syncthreads();
        if (ty == 0 && active) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = half * 4 + jj;
            if (j < g.proj_h) {
              float pv[VEC];
#pragma unroll
              for (int e = 0; e < VEC; ++e) pv[e] = accP[j][e];
              for (int r = 1; r < TY; ++r)
#pragma unroll
                for (int e = 0; e < VEC; ++e) pv[e] += sm[(jj * kBlock + r * TX + tx) * VEC + e];
              store_vec<float, VEC>(A.c.proj + (static_cast<size_t>(b) * g.proj_h + j) * g.HW + static_cast<size_t>(i) * VEC, pv);
            }
          }
        }
      }
    }
  }
}

template <typename T, int VEC, bool PROJ, bool NOPRO = false>
__global__ __launch_bounds__(kBlock) void k_chan(const Group<FwdArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  chan_body<T, VEC, PROJ, NOPRO>(G.lv[l], local, smem);
}

// ---------------------------------------------------------------------------------------------
// k_apply: y = x + alpha * (x*ca*sa - x)                                       masked_cbam.py:130,146-148,166-171
//   workgroup = (sample b, tile of TX vectors = TP contiguous pixels), rows take channel slices (k_chan's layout).
//   Prologue: sa for the tile's TP pixels = sigmoid(conv_kxk([max_c u, mean_c u, sigma(mask)])): the image rows the
//   tile touches (+ k/2 halo rows, full width + halo columns, zero padded) of the 3 planes are staged in LDS, one
//   pixel per thread accumulates its 3*k*k taps, sa goes to LDS (for the body) and to ctx (for backward).
//   LDS: [3*k*k weights][3 * nrows * (W+k-1) planes][TP sa]
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int K>
__device__ __forceinline__ void apply_body(const FwdArgs& A, const int bid, float* smem) {
  constexpr int UN = 4;
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
  T* yp = static_cast<T*>(A.y) + base;
  const float* cab = A.c.ca + static_cast<size_t>(b) * g.C;

  float x0[UN][VEC];
#pragma unroll
  for (int u = 0; u < UN; ++u) load_vec<T, VEC>(xp + static_cast<size_t>(min(ty + u * TY, g.C - 1)) * g.HW, x0[u]);

  // ---- prologue: spatial gate of this tile -------------------------------------------------------------------
  const int k = K ? K : g.k, pad = k / 2;
  const int TP = TX * VEC;
  const int p0 = tile * TP;                                     // first pixel of the tile
  const int p1 = min(p0 + TP, g.HW) - 1;                        // last pixel
  const int r0 = p0 / g.W, r1 = p1 / g.W;
  const int PW = g.W + k - 1, PH = (r1 - r0 + 1) + k - 1;
  float* wts = smem;
  float* planes = smem + ((3 * k * k + 3) & ~3);
  float* s_sa = planes + 3 * A.t.apply_rows * PW;               // apply_rows >= PH (host-computed bound)
  float* s_ca = s_sa + TX * VEC;
  for (int t = tid; t < 3 * k * k; t += kBlock) wts[t] = A.p.wsa[t];
  for (int c = tid; c < g.C; c += kBlock) s_ca[c] = cab[c];
  const float* pl = A.c.planes + static_cast<size_t>(b) * 3 * g.HW;
  stage_window<12>(planes, 3, PH, PW, r0 - pad, -pad, g, [&](int p, int off) { return pl[static_cast<size_t>(p) * g.HW + off]; });
  __syncthreads();
  for (int tp = tid; tp < TP; tp += kBlock) {                  // one pixel per thread (TP <= 512)
    if (p0 + tp >= g.HW) break;
    const int p = p0 + tp;
    const int py = p / g.W, px = p - py * g.W;
    const float* origin = planes + (py - r0) * PW + px;         // tap (i,j) of plane q: origin[q*PH*PW + i*PW + j]
    float acc = 0.f;
    if (K) {
      constexpr int KK = K ? K : 1;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int ti = 0; ti < KK; ++ti) {
          const float* row = origin + q * PH * PW + ti * PW;
          const float* wr = wts + (q * KK + ti) * KK;
#pragma unroll
          for (int tj = 0; tj < KK; ++tj) acc += wr[tj] * row[tj];
        }
      }
    } else {
      for (int q = 0; q < 3; ++q)
        for (int ti = 0; ti < k; ++ti) {
          const float* row = origin + q * PH * PW + ti * PW;
          const float* wr = wts + (q * k + ti) * k;
          for (int tj = 0; tj < k; ++tj) acc += wr[tj] * row[tj];
        }
    }
    const float sa = sigmoidf_(acc);                            // masked_cbam.py:147
    s_sa[tp] = sa;
    A.c.sa[static_cast<size_t>(b) * g.HW + p] = sa;
  }
  __syncthreads();
  float sav[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) sav[e] = active ? s_sa[tx * VEC + e] : 0.f;
  const float a = softplusf_(*A.p.beta);                        // masked_cbam.py:150-152

  // ---- body ------------------------------------------------------------------------------------------------------
  auto emit = [&](const float (&xv)[VEC], int c) {
    const float cac = s_ca[c];
    float yv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float u = xv[e] * cac;                              // masked_cbam.py:130
      const float v = u * sav[e];                               // masked_cbam.py:148
      yv[e] = xv[e] + a * (v - xv[e]);                          // masked_cbam.py:171
    }
    if (active) store_vec_stream<T, VEC>(yp + static_cast<size_t>(c) * g.HW, yv, A.t.nt_stores);
  };
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const int c = ty + u * TY;
    if (c < g.C) emit(x0[u], c);
  }
#pragma unroll 4
  for (int c = ty + UN * TY; c < g.C; c += TY) {
    float xv[VEC];
    load_vec<T, VEC>(xp + static_cast<size_t>(c) * g.HW, xv);
    emit(xv, c);
  }
}

template <typename T, int VEC, int K>
__global__ __launch_bounds__(kBlock) void k_apply(const Group<FwdArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  apply_body<T, VEC, K>(G.lv[l], local, smem);
}

// ---------------------------------------------------------------------------------------------
// k_gate: k_chan + k_apply in ONE pass over x with the tile RESIDENT IN REGISTERS (MGACBAM_FWD_FUSE).
//   workgroup = (sample b, tile of TX vectors = TP contiguous pixels) x ALL channels: thread (tx, ty) keeps channels
//   ty, ty+TY, ... (kGateR of them) of its VEC pixels in registers -- 64 KB of x per workgroup, read from HBM once.
//     1. issue the loads; meanwhile wait for ca of the sample from its role workgroup (gate_role: the shared MLP, once per sample)
//     2. per pixel max_c / mean_c of x*ca (LDS combine over the TY slices) -> planes, cidx to ctx (backward needs them anyway)
//     3. publish the tile's plane rows; wait for the tiles whose rows the k x k window of this tile touches (in-launch
//        hand-off, common.cuh).  Consumers only ever wait on tiles at most `span` tile-ids away and workgroups are
//        dispatched in id order, so the producers are resident or done; the host only selects this kernel when 8*span
//        workgroups fit on the chip many times over, and every wait is bounded anyway.
//     4. conv + sigmoid -> sa (k_apply's prologue), y = x + alpha (x ca sa - x) straight from the registers.
//   HBM traffic: x once + y once (2E) instead of k_chan + k_apply's 3E.
//   LDS: [C ca] then max(3*256*VEC combine, [3*k*k weights][3*rows*(W+k-1) planes][TP sa]);  role workgroup: [2C][2h][C]
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int K>
__device__ __forceinline__ void gate_body(const FwdArgs& A, const int bid, float* smem) {
  constexpr int R = kGateR;
  const Geo& g = A.g;
  const int tid = threadIdx.x;
  const int TX = A.t.gate_tx, lt = ilog2(TX);
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
  T* yp = static_cast<T*>(A.y) + base;

  const int gid = blockIdx.x;
  TRACE_HWID(A.trace, gid);
  TRACE_MARK(A.trace, gid, 0);                                 // start
  Pack<T, VEC> xr[R];                                          // the resident tile, as loaded (fp16 / bf16 stay packed: VEC = 8 -> 16 B)
#pragma unroll
  for (int j = 0; j < R; ++j)
    xr[j] = *reinterpret_cast<const Pack<T, VEC>*>(xp + static_cast<size_t>(min(ty + j * TY, g.C - 1)) * g.HW);
  int* flags = A.c.sync + static_cast<size_t>(b) * A.nflag;
  const int gen = static_cast<int>(static_cast<unsigned>(ld_agent(flags + tile)) + 1u);   // the generation this call brings every flag of the level to

  float* s_ca = smem;
  float* work = s_ca + ((g.C + 3) & ~3);
  bool bad;                                                    // a hand-off timed out: this tile's sa (hence y) is poisoned with NaN
  {                                                            // ca of this sample: from its role workgroup (gate_role)
    int* caflag = A.c.sync + static_cast<size_t>(g.B) * A.nflag + 4;
    bad = handoff_wait(caflag, b, b, gen, caflag - 4, A.spin_limit);
    const float* cab = A.c.ca + static_cast<size_t>(b) * g.C;
    TRACE_MARK(A.trace, gid, 1);                               // ca flag seen
    for (int c = tid; c < g.C; c += kBlock) s_ca[c] = ld_agent(cab + c);
    __syncthreads();
  }
  TRACE_MARK(A.trace, gid, 2);                                 // ca in LDS

  // ---- 2. channel max / mean planes of this tile ---------------------------------------------------------------
  float vmax[VEC], vsum[VEC];
  int vidx[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { vmax[e] = -INFINITY; vsum[e] = 0.f; vidx[e] = ty; }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int c = ty + j * TY;
    if (c < g.C) {
      const float cac = s_ca[c];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float uu = to_f32<T>(xr[j].v[e]) * cac;          // masked_cbam.py:130
        vsum[e] += uu;
        if (uu > vmax[e]) { vmax[e] = uu; vidx[e] = c; }
      }
    }
  }
  TRACE_MARK(A.trace, gid, 3);                                 // x arrived, per-thread channel scan done
  {
    // The TY channel slices of a pixel vector meet in two steps.  (1) Inside a wave: with TX < 64 a wave holds 64 / TX slices of the same
    // TX vectors, TX lanes apart -- xor butterflies over the lane offsets TX .. 32 leave every lane with the wave's combined triple.
    // (2) Across the waves through LDS: one partial per wave and vector, combined by the ty == 0 lanes.  (Round 2 sent all TY slices
    // through LDS and let TX lanes walk them one after the other: with TY = 16 / 32 -- C = 256 / 512 -- that loop was 5.2 us of a 31 us
    // workgroup, tools/trace_gate.py fwd cfg4.)
    for (int off = TX; off < kWave; off <<= 1) {                 // (uniform)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float ov = __shfl_xor(vmax[e], off);
        const int oi = __shfl_xor(vidx[e], off);
        argmax_combine(vmax[e], vidx[e], ov, oi);
        vsum[e] += __shfl_xor(vsum[e], off);
      }
    }
    const int spw = TX < kWave ? kWave / TX : 1;                 // slices per wave
    const int ngrp = TY / spw;                                   // partials per vector left to combine (4 whenever TX <= 64)
    const int grp = ty / spw;
    float* smax = work;
    int* sidx = reinterpret_cast<int*>(work + kBlock * VEC);
    float* ssum = work + 2 * kBlock * VEC;
    if (grp > 0 && ty == grp * spw) {
      const int o = (grp * TX + tx) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { smax[o + e] = vmax[e]; sidx[o + e] = vidx[e]; ssum[o + e] = vsum[e]; }
    }
    __syncthreads();
    if (ty == 0 && active) {
      for (int r = 1; r < ngrp; ++r) {
        const int o = (r * TX + tx) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          argmax_combine(vmax[e], vidx[e], smax[o + e], sidx[o + e]);
          vsum[e] += ssum[o + e];
        }
      }
      float* pl = A.c.planes + static_cast<size_t>(b) * 3 * g.HW + static_cast<size_t>(i) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        st_agent(pl + e, vmax[e]);
        st_agent(pl + g.HW + e, vsum[e] / static_cast<float>(g.C));   // masked_cbam.py:136
      }
      store_ivec<VEC>(A.c.cidx + static_cast<size_t>(b) * g.HW + static_cast<size_t>(i) * VEC, vidx);
    }
  }

  // ---- 3. hand the plane rows over ------------------------------------------------------------------------------
  const int k = K ? K : g.k, pad = k / 2;
  const int TP = TX * VEC;
  const int p0 = tile * TP;                                     // first pixel of the tile
  const int p1 = min(p0 + TP, g.HW) - 1;                        // last pixel
  const int r0 = p0 / g.W, r1 = p1 / g.W;
  const int PW = g.W + k - 1, PH = (r1 - r0 + 1) + k - 1;
  TRACE_MARK(A.trace, gid, 4);                                 // planes stored (not yet complete)
  handoff_publish(flags + tile);                               // (its barrier also frees the combine area)
  TRACE_MARK(A.trace, gid, 5);                                 // published
  float* wts = work;
  float* planes = work + ((3 * k * k + 3) & ~3);
  float* s_sa = planes + 3 * A.t.gate_rows * PW;                // gate_rows >= PH (host-computed bound)
  for (int t = tid; t < 3 * k * k; t += kBlock) wts[t] = A.p.wsa[t];
  // A tile inside ONE image row and narrower than it (wide feature maps: W = 160 at C >= 256, where a tile is 64 or 32 pixels) stages
  // only the columns its k x k windows reach, and waits only for the tiles that hold them: 7 x 70 values per plane instead of 7 x 166,
  // 2 tiles per row instead of all of them.  Every other tile (several rows, or a run that wraps around a row end) stages whole rows.
  const int ra = max(r0 - pad, 0), rb = min(r1 + pad, g.H - 1);
  const int c0 = p0 - r0 * g.W;
  const bool narrow = r0 == r1 && (p1 - p0 + 1) + k - 1 < PW;
  const int xa = narrow ? c0 - pad : -pad;                      // image column of the window's first column
  const int PWs = narrow ? (p1 - p0 + 1) + k - 1 : PW;          // staged width
  if (narrow) {
    const int cl = max(c0 - pad, 0), ch = min(c0 + (p1 - p0) + pad, g.W - 1);
    bad |= handoff_wait_rows(flags, ra, rb, g.W, TP, cl, ch, gen, A.c.sync + static_cast<size_t>(g.B) * A.nflag, A.spin_limit);
  } else {
    bad |= handoff_wait(flags, (ra * g.W) / TP, ((rb + 1) * g.W - 1) / TP, gen, A.c.sync + static_cast<size_t>(g.B) * A.nflag, A.spin_limit);
  }
  TRACE_MARK(A.trace, gid, 6);                                 // neighbours' rows are there
  const float* pl = A.c.planes + static_cast<size_t>(b) * 3 * g.HW;
  stage_window<12>(planes, 3, PH, PWs, r0 - pad, xa, g, [&](int p, int off) {
    const float* q = pl + static_cast<size_t>(p) * g.HW + off;
    return p < 2 ? ld_agent(q) : *q;                       // plane 2 = sigma(mask), written by k_pool (previous launch)
  });
  __syncthreads();
  TRACE_MARK(A.trace, gid, 7);                                 // window staged

  // ---- 4. spatial gate of the tile, then y from the registers ----------------------------------------------------
  // (a register-tiled form -- wave = plane, lane = 4 adjacent pixels, ~5x fewer LDS instructions -- was measured: the conv
  //  phase went 2.2 -> 1.7 us per round, the launch did not move, 27 more VGPRs; not kept)
  for (int tp = tid; tp < TP; tp += kBlock) {
    if (p0 + tp >= g.HW) break;
    const int p = p0 + tp;
    const int py = p / g.W, px = p - py * g.W;
    const float* origin = planes + (py - r0) * PWs + (px - pad - xa);   // tap (i,j) of plane q: origin[q*PH*PWs + i*PWs + j]
    float acc = 0.f;
    if (K) {
      constexpr int KK = K ? K : 1;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int ti = 0; ti < KK; ++ti) {
          const float* row = origin + q * PH * PWs + ti * PWs;
          const float* wr = wts + (q * KK + ti) * KK;
#pragma unroll
          for (int tj = 0; tj < KK; ++tj) acc += wr[tj] * row[tj];
        }
      }
    } else {
      for (int q = 0; q < 3; ++q)
        for (int ti = 0; ti < k; ++ti) {
          const float* row = origin + q * PH * PWs + ti * PWs;
          const float* wr = wts + (q * k + ti) * k;
          for (int tj = 0; tj < k; ++tj) acc += wr[tj] * row[tj];
        }
    }
    const float sa = bad ? __builtin_nanf("") : sigmoidf_(acc);   // masked_cbam.py:147 (NaN: loud failure of a timed-out hand-off)
    s_sa[tp] = sa;
    A.c.sa[static_cast<size_t>(b) * g.HW + p] = sa;
  }
  __syncthreads();
  TRACE_MARK(A.trace, gid, 8);                                 // conv done
#ifndef MGACBAM_TRACE
  if (!active) return;
#endif
  float sav[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) sav[e] = active ? s_sa[tx * VEC + e] : 0.f;
  const float a = softplusf_(*A.p.beta);                        // masked_cbam.py:150-152
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int c = ty + j * TY;
    if (c < g.C) {
      const float cac = s_ca[c];
      float yv[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xe = to_f32<T>(xr[j].v[e]);
        const float u = xe * cac;                               // masked_cbam.py:130
        yv[e] = xe + a * (u * sav[e] - xe);                     // masked_cbam.py:148,171
      }
      if (active) store_vec_stream<T, VEC>(yp + static_cast<size_t>(c) * g.HW, yv, A.t.nt_stores);
    }
  }
#ifdef MGACBAM_TRACE
  TRACE_MARK(A.trace, gid, 9);                                 // stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TRACE_MARK(A.trace, gid, 10);                                // stores complete
#endif
}

// role workgroup of k_gate (one per sample, lowest ids of the launch): the shared MLP, once per sample instead of once per
// tile -- as a prologue of every tile workgroup its three dependent phases cost 8.7 us per step (ablation), as the tail of
// k_pool's last arriver 9.4 us; here it runs while the tile workgroups' x loads are in flight.
// The MLP is a chain of dependent steps (avg/mx -> hidden -> z); fetched step by step every step pays a cross-XCD miss (the
// operands were written by k_pool's workgroups on other XCDs, the weights are cold) and the loops over C and hidden serialise more
// of them: 9-14 us at C = 64..256, 15-20 us at C = 512, during which the sample's tiles sit on their ca flag (k_gate) or every
// workgroup of k_chan idles (prologue form).  mlp_stream requests operands in bulk instead: avg, mx and a whole pass of W1 rows
// (up to kMlpW1 16-byte vectors per lane) are in flight together, the W2 rows are requested as soon as the W1 registers are
// consumed, so the chain costs two memory latencies plus arithmetic.  Needs C % 4 == 0 and hidden % 4 == 0 (16-byte rows);
// other shapes use the staged form (mlp_gate_to_lds).
//   smem: [2C avg|mx][2h hidden]        ca is written with st_agent when `agent` (consumed inside the same launch), else plainly
constexpr int kMlpW1 = 8;       // 16-byte vectors of W1 a lane holds per pass
constexpr int kMlpW2 = 8;       // 16-byte vectors of W2 a thread holds per pass
__device__ __forceinline__ bool mlp_stream_ok(const Geo& g) { return (g.C & 3) == 0 && (g.hidden & 3) == 0 && g.C <= 8 * 4 * kWave; }

// hidden layer: wave w owns hidden units w, w+4, ...; a lane holds CVL vectors of each of UP rows per pass
template <int CVL>
__device__ __forceinline__ void mlp_hidden(const FwdArgs& A, const int b, float* s_avg, float* s_mx, float* s_h) {
  constexpr int UP = kMlpW1 / CVL;
  const Geo& g = A.g;
  const int tid = threadIdx.x, C = g.C, h = g.hidden, CV = C >> 2;
  const int wave = tid >> 6, lane = tid & 63;
  const float4* w1v = reinterpret_cast<const float4*>(A.p.w1);
  const int nunit = h > wave ? (h - wave + 3) >> 2 : 0;        // units of this wave
  const int npass = (((h + 3) >> 2) + UP - 1) / UP;            // uniform over the workgroup
  for (int p = 0; p < npass; ++p) {
    float4 wr[UP][CVL];
#pragma unroll
    for (int uu = 0; uu < UP; ++uu) {
      const int u = p * UP + uu, j = wave + 4 * u;
#pragma unroll
      for (int i = 0; i < CVL; ++i) {
        const int cv = lane + kWave * i;
        wr[uu][i] = (u < nunit && cv < CV) ? w1v[static_cast<size_t>(j) * CV + cv] : float4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (p == 0) {                                               // the pooled descriptors, requested right behind the first W1 rows
      for (int c = tid; c < C; c += kBlock) {
        s_avg[c] = A.c.avg[static_cast<size_t>(b) * C + c];
        s_mx[c] = A.c.mx[static_cast<size_t>(b) * C + c];
      }
      __syncthreads();
    }
#pragma unroll
    for (int uu = 0; uu < UP; ++uu) {
      const int u = p * UP + uu;
      if (u < nunit) {                                          // (uniform per wave)
        float da = 0.f, dm = 0.f;
#pragma unroll
        for (int i = 0; i < CVL; ++i) {
          const int cv = lane + kWave * i;
          if (cv < CV) {
            const float4 w = wr[uu][i];
            const float4 a4 = *reinterpret_cast<const float4*>(s_avg + 4 * cv), m4 = *reinterpret_cast<const float4*>(s_mx + 4 * cv);
            da += w.x * a4.x + w.y * a4.y + w.z * a4.z + w.w * a4.w;
            dm += w.x * m4.x + w.y * m4.y + w.z * m4.z + w.w * m4.w;
          }
        }
        da = wave_group_sum(da, kWave);
        dm = wave_group_sum(dm, kWave);
        if (lane == 0) {
          const int j = wave + 4 * u;
          const float bj = A.p.b1[j];
          const float ha = fmaxf(da + bj, 0.f), hm = fmaxf(dm + bj, 0.f);
          s_h[j] = ha; s_h[h + j] = hm;
          A.c.h_avg[static_cast<size_t>(b) * h + j] = ha;
          A.c.h_mx[static_cast<size_t>(b) * h + j] = hm;
        }
      }
    }
  }
}

template <bool AGENT>
__device__ __forceinline__ void mlp_stream(const FwdArgs& A, const int b, float* smem) {
  const Geo& g = A.g;
  const int tid = threadIdx.x, C = g.C, h = g.hidden;
  float* s_avg = smem;
  float* s_mx = smem + C;
  float* s_h = smem + 2 * C;
  const int cvl = ((C >> 2) + kWave - 1) / kWave;              // vectors of a W1 row per lane: 1 (C <= 256), 2, 4, 8 (C <= 2048)
  if (cvl <= 1) mlp_hidden<1>(A, b, s_avg, s_mx, s_h);
  else if (cvl <= 2) mlp_hidden<2>(A, b, s_avg, s_mx, s_h);
  else if (cvl <= 4) mlp_hidden<4>(A, b, s_avg, s_mx, s_h);
  else mlp_hidden<8>(A, b, s_avg, s_mx, s_h);
  // ---- output layer: thread = channel; its first W2 row is requested BEFORE the barrier that publishes the hidden activations -------
  const int HV = h >> 2;
  const float4* w2v = reinterpret_cast<const float4*>(A.p.w2);
  float4 wr[kMlpW2];
  {
    const int c = min(tid, C - 1);
#pragma unroll
    for (int q = 0; q < kMlpW2; ++q) wr[q] = q < HV ? w2v[static_cast<size_t>(c) * HV + q] : float4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  for (int c = tid; c < C; c += kBlock) {
    float za = 0.f, zm = 0.f;
    for (int v0 = 0; v0 < HV; v0 += kMlpW2) {
      if (c != tid || v0 != 0) {
#pragma unroll
        for (int q = 0; q < kMlpW2; ++q) wr[q] = v0 + q < HV ? w2v[static_cast<size_t>(c) * HV + v0 + q] : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < kMlpW2; ++q) {
        if (v0 + q < HV) {
          const float* ha = s_h + 4 * (v0 + q);
          const float* hm = ha + h;
          za += wr[q].x * ha[0]; zm += wr[q].x * hm[0];          // j ascending: the summation order of mlp_gate_to_lds
          za += wr[q].y * ha[1]; zm += wr[q].y * hm[1];
          za += wr[q].z * ha[2]; zm += wr[q].z * hm[2];
          za += wr[q].w * ha[3]; zm += wr[q].w * hm[3];
        }
      }
    }
    const float bc = A.p.b2[c];
    const float ca = sigmoidf_((za + bc) + (zm + bc));
    if (AGENT) st_agent(A.c.ca + static_cast<size_t>(b) * C + c, ca);
    else A.c.ca[static_cast<size_t>(b) * C + c] = ca;
  }
}

__device__ __forceinline__ void gate_role(const FwdArgs& A, const int b, float* smem) {
  const Geo& g = A.g;
  const int tid = threadIdx.x, C = g.C, h = g.hidden;
  int* caflag = A.c.sync + static_cast<size_t>(g.B) * A.nflag + 4 + b;
  TRACE_MARK(A.trace, blockIdx.x, 0);
  if (mlp_stream_ok(g)) {
    mlp_stream<true>(A, b, smem);
    TRACE_MARK(A.trace, blockIdx.x, 3);
  } else {
    float* s_in = smem;
    float* s_h = smem + 2 * C;
    float* s_ca = s_h + 2 * h;
    mlp_gate_to_lds(A, b, false, s_in, s_h, s_ca);
    TRACE_MARK(A.trace, blockIdx.x, 3);
    for (int c = tid; c < C; c += kBlock) st_agent(A.c.ca + static_cast<size_t>(b) * C + c, s_ca[c]);
    for (int j = tid; j < h; j += kBlock) {
      A.c.h_avg[static_cast<size_t>(b) * h + j] = s_h[j];
      A.c.h_mx[static_cast<size_t>(b) * h + j] = s_h[h + j];
    }
  }
  if (!(A.fault && b == 0)) handoff_publish(caflag);           // (fault injection, tests only: sample 0's tiles time out)
  TRACE_MARK(A.trace, blockIdx.x, 5);
}

// k_mlp: the shared MLP + channel gate as a launch of its own, one workgroup per sample (levels concatenated).  Used in front of
// k_chan when the prologue form would cost more than a launch: with C*hidden large every one of k_chan's ~2000 workgroups spends
// 15-20 us in the MLP before it streams (config 4: k_chan 98 us against 53 us of traffic).
__global__ __launch_bounds__(kBlock) void k_mlp(const Group<FwdArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const FwdArgs& A = G.lv[l];
  if (mlp_stream_ok(A.g)) {
    mlp_stream<false>(A, local, smem);
  } else {
    float* s_h = smem + 2 * A.g.C;
    mlp_gate_to_lds(A, local, true, smem, s_h, s_h + 2 * A.g.hidden);
  }
}

struct GateGroup {
  Group<FwdArgs> g;
  int nrole;                     // role workgroups = sum of the levels' batch sizes; tile workgroup ids follow
  int rstart[kGroupMax + 1];     // level l owns role ids [rstart[l], rstart[l+1])
};


#ifdef MGACBAM_GATE_WAVES
#define GATE_OCC __attribute__((amdgpu_waves_per_eu(MGACBAM_GATE_WAVES)))
#else
#define GATE_OCC
#endif
template <typename T, int VEC, int K>
__global__ __launch_bounds__(kBlock) GATE_OCC void k_gate(const GateGroup GG) {
  extern __shared__ __align__(16) float smem[];
  const int bid = blockIdx.x;
  if (bid < GG.nrole) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kGroupMax; ++i)
      if (i < GG.g.n && bid >= GG.rstart[i]) l = i;
    gate_role(GG.g.lv[l], bid - GG.rstart[l], smem);
    return;
  }
  int local;
  const int l = find_level(GG.g, bid - GG.nrole, local);
  gate_body<T, VEC, K>(GG.g.lv[l], local, smem);
}

}  // namespace mgacbam
