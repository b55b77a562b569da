// MGAMaskHead on the device: the producer of the mask logits MaskCBAM consumes.
//   reference: mga_yolo/nn/modules/segmentation.py:56-110 (default configuration: norm="bn", act=SiLU, dropout=0, out_channels=1)
//       z = Conv1x1(x)  (hid x C GEMM per pixel, no bias)  ->  BatchNorm2d (batch statistics in training)  ->  SiLU  ->  Conv3x3 + bias
//   layer-loop hand-off: mga_yolo/model/model.py:57-74 (the logits go to MaskCBAM as `mask` and to the segmentation loss)
//
// forward  (3 launches)  k_head_proj    x (1 read)   -> z (B,hid,HW) fp32 + per-workgroup partial sums of z, z^2           [MFMA fp32 16x16x4]
//                        k_head_stats   partials     -> mean, rstd (+ running statistics, num_batches_tracked)               [tiny]
//                        k_head_out     z            -> s = SiLU(gamma*zhat+beta), staged with a 1-px halo in LDS -> 3x3 conv -> logits
// backward (5 launches)  k_head_bwd_act z, g_logits  -> g_a (B,hid,HW) = convT3x3(g) * SiLU'(a); partials of sum g_a, sum g_a*zhat,
//                                                        dW_h (9 taps), db_h
//                        k_head_bwd_fin partials     -> dgamma, dbeta, dW_h, db_h and the per-channel constants of g_z
//                        k_head_bwd_gx  g_a, z       -> g_z on the fly -> gx = W1^T g_z  (1 write)                              [MFMA]
//                        k_head_bwd_gw  g_a, z, x    -> partials of dW1 = sum_px g_z x^T  (x: 1 read)                            [MFMA]
//                        k_head_bwd_gwf partials     -> dW1
// One launch covers every level of the call (P3+P4+P5), as everywhere in this library.
// The 1x1 conv is the only GEMM-shaped piece next to the hot path: M = hid (16..256), K = C (64..768), N = B*H*W.  fp32 MFMA runs at the
// FP32 vector rate (MI355X_MICROARCH.md: 256 FLOP/clk/CU either way), so it buys instruction-issue relief, not FLOPs: at YOLOv8n
// widths (hid 16..64) the kernels are HBM-bound on x; at m/l widths (hid 64..256) the fp32 GEMM is ~2x the HBM time.
// All cross-workgroup sums are two-stage with a fixed order (no float atomics): bitwise reproducible run to run.
#pragma once
#include "common.cuh"

namespace mgacbam {

typedef float v4f32 __attribute__((ext_vector_type(4)));

struct HeadGeo {
  int B, C, hid, H, W, HW;
  int hidp;            // hid rounded up to 16 (MFMA tiles)
  int cp;              // C rounded up to 16
  float eps, momentum;
  int training;
};
struct HeadPtrs {
  const float* w1;     // (hid, C)        proj.0.weight
  const float* gamma;  // (hid)           proj.1.weight
  const float* beta;   // (hid)           proj.1.bias
  float* rmean;        // (hid)           proj.1.running_mean   (updated in training)
  float* rvar;         // (hid)           proj.1.running_var
  long long* nbt;      // ()              proj.1.num_batches_tracked (int64) or null
  const float* wh;     // (1, hid, 3, 3)  head.weight
  const float* bh;     // (1)             head.bias
};
// ctx (saved for backward): z (B,hid,HW) fp32 | mean[hidp] | rstd[hidp] | part (nwg, 2, hidp)
struct HeadCtx { float* z; float* mean; float* rstd; float* part; };
// scratch (backward transients): g_a (B,hid,HW) | part1 (nwg1, hidp, 12) | kst (5, hidp) | gwpart (ncb*kHeadGwWG, hidp, kHeadCB)
struct HeadScratch { float* ga; float* part1; float* kst; float* gwpart; };

struct HeadArgs {
  const void* x; void* logits;                 // forward
  const void* gl; void* gx;                    // backward: dL/dlogits (B,1,H,W) T, dL/dx (B,C,H,W) T
  float* gw1; float* ggamma; float* gbeta; float* gwh; float* gbh;
  HeadPtrs p; HeadCtx c; HeadScratch s; HeadGeo g;
  int tile_px, tiles_per_sample, nwg;          // 1-D pixel tiling of k_head_gemm<FWD> (nwg = rows of the partial sums)
  int gx_tile_px, gx_tiles_per_sample;         // ... of k_head_gemm<GX>
  int t2x, t2y, nwg1;                          // 2-D tiling of the conv kernels (k_head_out, k_head_bwd_act): tiles per row / column / level
  int ncb;                                     // channel blocks of k_head_bwd_gw
};

constexpr int kHeadMTW = 4;      // 16-output tiles a wave accumulates at once (x VEC sub-tiles x 4 registers)
constexpr int kHeadLdsA = 8192;  // floats of LDS for the staged A operand (32 KB)
constexpr int kHeadT2 = 16;      // k_head_out / k_head_bwd_act: 16 x 16 pixel tiles
constexpr int kHeadJC = 16;      // ... hidden channels per LDS pass
constexpr int kHeadCB = 64;      // k_head_bwd_gw: channels of x per workgroup (4 N tiles)
constexpr int kHeadGwWG = 96;    // k_head_bwd_gw: workgroups per (level, channel block) = partial sets of dW1
constexpr int kHeadNStat = 12;   // per-channel partial sums of k_head_bwd_act: g_a, g_a*zhat, 9 taps of dW_h, db_h
__device__ __forceinline__ float siluf(float a) { return a / (1.f + expf(-a)); }

// ---------------------------------------------------------------------------------------------------------------------------
// The two pixel-parallel GEMMs:   FWD  z[b,j,px]  = sum_c W1[j,c] x[b,c,px]      (M = hid, K = C)          segmentation.py:81
//                                 GX   gx[b,c,px] = sum_j W1[j,c] g_z[b,j,px]    (M = C,   K = hid)        its autograd backward
//   v_mfma_f32_16x16x4_f32: A = weight tile (16 outputs x 4 k; lane l: output l%16, k l/16), B = activation tile (4 k x 16 pixel groups;
//   lane l: k l/16, group l%16), D 16 x 16 in 4 registers (lane l, register v: output 4*(l/16)+v, group l%16).  With VEC = 4 a lane loads
//   16 B = 4 consecutive pixels of its k-channel (a wave-load = 4 channels x 256 B contiguous) and register r of that vector is the B
//   operand of sub-tile r: 64 pixels per wave and K step; VEC = 1 (H*W % 4 != 0): one dword per lane, 16 pixels per wave.
//   Workgroup = 4 waves = MW (along M) x PW (along pixels); the weight block [M block][K block] is staged in LDS in the A layout
//   ([kstep][mtile][lane]: conflict-free ds_read_b32).  Blocks: M in blocks of 16*kHeadMTW*MW outputs, K in blocks that fit kHeadLdsA.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int VEC, bool GX>
__device__ __forceinline__ void head_gemm_body(const HeadArgs& A, const int wg, float* smem) {
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lk = lane >> 4, ln = lane & 15;
  const int M = GX ? g.C : g.hid, K = GX ? g.hid : g.C;
  const int MT = ((GX ? g.cp : g.hidp)) >> 4;
  const int MW = min(4, (MT + kHeadMTW - 1) / kHeadMTW);        // 1, 2 or 4 (MT = 3 tiles of waves -> 4)
  const int MWp = MW == 3 ? 4 : MW;
  const int PW = 4 / MWp;
  const int mw = wave % MWp, pw = wave / MWp;
  constexpr int WPX = 16 * VEC;
  const int tps = GX ? A.gx_tiles_per_sample : A.tiles_per_sample;
  const int b = wg / tps, tile = wg - b * tps;
  const int px = tile * (GX ? A.gx_tile_px : A.tile_px) + pw * WPX + ln * VEC;   // first pixel of this lane
  const bool px_ok = px < g.HW;
  const size_t pxo = px_ok ? px : 0;
  const int mblk = kHeadMTW * MWp;                              // M tiles per block
  const int rows = min(MT, mblk) * 16;
  const int KB = min((K + 3) & ~3, max(4, (kHeadLdsA / rows) & ~3));   // K per LDS block (multiple of 4)
  float* s_kst = smem + kHeadLdsA;                              // GX: per-hidden-channel constants of g_z [5][hidp]
  if (GX) {
    for (int i = tid; i < 5 * g.hidp; i += kBlock) s_kst[i] = A.s.kst[i];
  }
  float* s_sum = smem + kHeadLdsA;                              // FWD: [PW][2][hidp] tile sums
  const T* xb = GX ? nullptr : static_cast<const T*>(A.x) + static_cast<size_t>(b) * g.C * g.HW + pxo;
  const float* gab = GX ? A.s.ga + static_cast<size_t>(b) * g.hid * g.HW + pxo : nullptr;
  const float* zb = A.c.z + static_cast<size_t>(b) * g.hid * g.HW + pxo;

  for (int mt0 = 0; mt0 < MT; mt0 += mblk) {
    const int mtn = min(mblk, MT - mt0);                        // tiles in this block
    v4f32 acc[kHeadMTW][VEC];
#pragma unroll
    for (int t = 0; t < kHeadMTW; ++t)
#pragma unroll
      for (int r = 0; r < VEC; ++r) acc[t][r] = v4f32{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += KB) {
      const int kn = min(KB, K - k0);
      const int ksteps = (kn + 3) >> 2;
      __syncthreads();                                          // previous block consumed (and s_kst written)
      for (int idx = tid; idx < ksteps * mtn * 64; idx += kBlock) {
        const int l = idx & 63, q = idx >> 6;
        const int mt = q % mtn, ks = q / mtn;
        const int out = (mt0 + mt) * 16 + (l & 15), kk = k0 + ks * 4 + (l >> 4);
        float w = 0.f;
        if (out < M && kk < k0 + kn) w = GX ? A.p.w1[static_cast<size_t>(kk) * g.C + out] : A.p.w1[static_cast<size_t>(out) * g.C + kk];
        smem[idx] = w;
      }
      __syncthreads();
      for (int ks = 0; ks < ksteps; ++ks) {
        const int kk = k0 + ks * 4 + lk;                        // this lane's k channel
        float bv[VEC];
#pragma unroll
        for (int r = 0; r < VEC; ++r) bv[r] = 0.f;
        if (px_ok && kk < k0 + kn) {
          if (GX) {                                             // g_z = k_j (g_a - gbeta_j/n - zhat gamma'_j/n), zhat = (z - mean) rstd
            float ga[VEC], zv[VEC];
            load_vec<float, VEC>(gab + static_cast<size_t>(kk) * g.HW, ga);
            load_vec<float, VEC>(zb + static_cast<size_t>(kk) * g.HW, zv);
            const float kj = s_kst[kk], mean = s_kst[g.hidp + kk], rstd = s_kst[2 * g.hidp + kk];
            const float gbn = s_kst[3 * g.hidp + kk], ggn = s_kst[4 * g.hidp + kk];
#pragma unroll
            for (int r = 0; r < VEC; ++r) bv[r] = kj * (ga[r] - gbn - (zv[r] - mean) * rstd * ggn);
          } else {
            load_vec<T, VEC>(xb + static_cast<size_t>(kk) * g.HW, bv);
          }
        }
#pragma unroll
        for (int t = 0; t < kHeadMTW; ++t) {
          const int mt = mw * kHeadMTW + t;
          if (mt < mtn) {                                       // uniform per wave
            const float a = smem[(ks * mtn + mt) * 64 + lane];
#pragma unroll
            for (int r = 0; r < VEC; ++r) acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[r], acc[t][r], 0, 0, 0);
          }
        }
      }
    }
    // ---- epilogue of this M block -------------------------------------------------------------------------------------------
    if (!GX && g.training) __syncthreads();                     // LDS block consumed before s_sum (separate region, but keep waves together)
#pragma unroll
    for (int t = 0; t < kHeadMTW; ++t) {
      const int mt = mw * kHeadMTW + t;
      if (mt < mtn) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int out = (mt0 + mt) * 16 + lk * 4 + v;
          float ov[VEC];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int r = 0; r < VEC; ++r) {
            ov[r] = acc[t][r][v];
            s1 += ov[r]; s2 += ov[r] * ov[r];                    // pixels past H*W were loaded as zeros: they add nothing
          }
          if (out < M && px_ok) {
            if (GX) store_vec_stream<T, VEC>(static_cast<T*>(A.gx) + (static_cast<size_t>(b) * g.C + out) * g.HW + px, ov, true);
            else store_vec<float, VEC>(A.c.z + (static_cast<size_t>(b) * g.hid + out) * g.HW + px, ov);
          }
          if (!GX && g.training) {
            s1 = wave_group_sum(s1, 16);                         // over the 16 pixel-group lanes that share this output channel
            s2 = wave_group_sum(s2, 16);
            if (ln == 0) { s_sum[(pw * 2 + 0) * g.hidp + out] = s1; s_sum[(pw * 2 + 1) * g.hidp + out] = s2; }
          }
        }
      }
    }
  }
  if (!GX && g.training) {                                      // tile sums of z and z^2 per output channel (hid <= 16*kHeadMTW*4: one M block)
    __syncthreads();
    float* part = A.c.part + static_cast<size_t>(wg) * 2 * g.hidp;
    for (int i = tid; i < 2 * g.hidp; i += kBlock) {
      float s = 0.f;
      for (int w = 0; w < PW; ++w) s += s_sum[w * 2 * g.hidp + i];
      part[i] = s;
    }
  }
}

template <typename T, int VEC, bool GX>
__global__ __launch_bounds__(kBlock) void k_head_gemm(const Group<HeadArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  head_gemm_body<T, VEC, GX>(G.lv[l], local, smem);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_head_stats: batch statistics of z per hidden channel from the tile partials (fixed order, double accumulation), BatchNorm's
//   running-statistics update (segmentation.py:83 -> torch BatchNorm2d: biased variance for the normalisation, unbiased for the
//   running estimate, momentum m).  Eval mode: mean / rstd from the running statistics.  Workgroup = 8 channels (16 sums) x 16 strides.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_head_stats(const Group<HeadArgs> G) {
  __shared__ double red[16][17];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const HeadArgs& A = G.lv[l];
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x, s = tid & 15, gq = tid >> 4;
  const int j0 = local * 8;
  const int j = j0 + (s >> 1), which = s & 1;                   // sum index s = 2*(j - j0) + {0: z, 1: z^2}
  double acc = 0.0;
  if (g.training && j < g.hid) {
    for (int w = gq; w < A.nwg; w += 16) acc += static_cast<double>(A.c.part[(static_cast<size_t>(w) * 2 + which) * g.hidp + j]);
  }
  red[gq][s] = acc;
  __syncthreads();
  if (tid < 8) {
    const int jj = j0 + tid;
    if (jj < g.hid) {
      float mean, var;
      if (g.training) {
        double s1 = 0.0, s2 = 0.0;
        for (int q = 0; q < 16; ++q) { s1 += red[q][2 * tid]; s2 += red[q][2 * tid + 1]; }
        const double n = static_cast<double>(g.B) * g.HW;
        const double m = s1 / n;
        double v = s2 / n - m * m;
        if (v < 0.0) v = 0.0;
        mean = static_cast<float>(m); var = static_cast<float>(v);
        const double unb = n > 1.0 ? v * n / (n - 1.0) : v;
        A.p.rmean[jj] = (1.f - g.momentum) * A.p.rmean[jj] + g.momentum * mean;
        A.p.rvar[jj] = (1.f - g.momentum) * A.p.rvar[jj] + g.momentum * static_cast<float>(unb);
      } else {
        mean = A.p.rmean[jj]; var = A.p.rvar[jj];
      }
      A.c.mean[jj] = mean;
      A.c.rstd[jj] = 1.0f / sqrtf(var + g.eps);
    }
    if (g.training && local == 0 && tid == 0 && A.p.nbt) *A.p.nbt += 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_head_out: logits = conv3x3(SiLU(gamma * zhat + beta)) + bias                                  segmentation.py:83-92
//   workgroup = 16 x 16 pixels of one sample, one pixel per thread; the activations of kHeadJC channels with a 1-px halo are staged
//   in LDS per pass (zero padding outside the image), the 3x3 weights of the pass beside them.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void head_out_body(const HeadArgs& A, const int wg, float* smem) {
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x;
  constexpr int TS = kHeadT2, PS = TS + 2;
  const int per = A.t2x * A.t2y;
  const int b = wg / per, tt = wg - b * per;
  const int y0 = (tt / A.t2x) * TS, x0 = (tt % A.t2x) * TS;
  const int ty = tid / TS, tx = tid - ty * TS;
  float* s_act = smem;                                          // [JC][PS][PS]
  float* s_w = smem + kHeadJC * PS * PS;                        // [JC][9]
  float* s_bn = s_w + kHeadJC * 9;                              // [JC][2]: scale = gamma*rstd, shift = beta - mean*scale
  float acc = 0.f;
  for (int j0 = 0; j0 < g.hid; j0 += kHeadJC) {
    const int jn = min(kHeadJC, g.hid - j0);
    __syncthreads();
    for (int i = tid; i < jn * 9; i += kBlock) s_w[i] = A.p.wh[static_cast<size_t>(j0) * 9 + i];
    if (tid < jn) {
      const float sc = A.p.gamma[j0 + tid] * A.c.rstd[j0 + tid];
      s_bn[2 * tid] = sc; s_bn[2 * tid + 1] = A.p.beta[j0 + tid] - A.c.mean[j0 + tid] * sc;
    }
    __syncthreads();
    for (int i = tid; i < jn * PS * PS; i += kBlock) {
      const int jj = i / (PS * PS), r = i - jj * PS * PS;
      const int yy = y0 + r / PS - 1, xx = x0 + r % PS - 1;
      float sv = 0.f;
      if (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) {
        const float z = A.c.z[(static_cast<size_t>(b) * g.hid + j0 + jj) * g.HW + yy * g.W + xx];
        sv = siluf(z * s_bn[2 * jj] + s_bn[2 * jj + 1]);
      }
      s_act[i] = sv;
    }
    __syncthreads();
    for (int jj = 0; jj < jn; ++jj) {
      const float* a = s_act + jj * PS * PS + ty * PS + tx;
      const float* w = s_w + jj * 9;
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) acc += w[u * 3 + v] * a[u * PS + v];
    }
  }
  const int y = y0 + ty, x = x0 + tx;
  if (y < g.H && x < g.W)
    static_cast<T*>(A.logits)[static_cast<size_t>(b) * g.HW + y * g.W + x] = from_f32<T>(acc + A.p.bh[0]);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_head_out(const Group<HeadArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  head_out_body<T>(G.lv[l], local, smem);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_head_bwd_act: g_s = convT3x3(g_logits, W_h);  g_a = g_s * SiLU'(a)  (stored);  per tile and hidden channel the partial sums
//   [0] sum g_a   [1] sum g_a*zhat   [2..10] dW_h taps: sum s(y,x) * g(y-u+1, x-v+1)   [11] sum g (channel 0 only: db_h)
//   Same 16 x 16 tiling; g_logits with its halo in LDS, each thread keeps its 9 neighbours in registers for all channels.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void head_bwd_act_body(const HeadArgs& A, const int wg, float* smem) {
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int TS = kHeadT2, PS = TS + 2;
  const int per = A.t2x * A.t2y;
  const int b = wg / per, tt = wg - b * per;
  const int y0 = (tt / A.t2x) * TS, x0 = (tt % A.t2x) * TS;
  const int ty = tid / TS, tx = tid - ty * TS;
  float* s_g = smem;                                            // [PS][PS]
  float* s_red = smem + PS * PS;                                // [4 waves][kHeadJC][kHeadNStat]
  const T* gl = static_cast<const T*>(A.gl) + static_cast<size_t>(b) * g.HW;
  for (int i = tid; i < PS * PS; i += kBlock) {
    const int yy = y0 + i / PS - 1, xx = x0 + i % PS - 1;
    s_g[i] = (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? to_f32<T>(gl[yy * g.W + xx]) : 0.f;
  }
  __syncthreads();
  float g9[9];                                                  // g9[u*3+v] = g(y-u+1, x-v+1)
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) g9[u * 3 + v] = s_g[(ty + 2 - u) * PS + (tx + 2 - v)];
  const int y = y0 + ty, x = x0 + tx;
  const bool in = y < g.H && x < g.W;
  const size_t po = static_cast<size_t>(in ? y * g.W + x : 0);
  float* part = A.s.part1 + static_cast<size_t>(wg) * g.hidp * kHeadNStat;
  for (int j0 = 0; j0 < g.hid; j0 += kHeadJC) {
    const int jn = min(kHeadJC, g.hid - j0);
    for (int jj = 0; jj < jn; ++jj) {
      const int j = j0 + jj;
      float r[kHeadNStat];
#pragma unroll
      for (int q = 0; q < kHeadNStat; ++q) r[q] = 0.f;
      if (in) {
        const size_t o = (static_cast<size_t>(b) * g.hid + j) * g.HW + po;
        const float z = A.c.z[o];
        const float rstd = A.c.rstd[j];
        const float zh = (z - A.c.mean[j]) * rstd;
        const float a = zh * A.p.gamma[j] + A.p.beta[j];
        const float sg = 1.f / (1.f + expf(-a));
        const float sv = a * sg;
        const float* w = A.p.wh + static_cast<size_t>(j) * 9;
        float gs = 0.f;
#pragma unroll
        for (int q = 0; q < 9; ++q) gs += w[q] * g9[q];
        const float ga = gs * (sg * (1.f + a * (1.f - sg)));
        A.s.ga[o] = ga;
        r[0] = ga; r[1] = ga * zh;
#pragma unroll
        for (int q = 0; q < 9; ++q) r[2 + q] = sv * g9[q];
        r[11] = j == 0 ? g9[4] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < kHeadNStat; ++q) r[q] = wave_group_sum(r[q], kWave);
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kHeadNStat; ++q) s_red[(wave * kHeadJC + jj) * kHeadNStat + q] = r[q];
      }
    }
    __syncthreads();
    for (int i = tid; i < jn * kHeadNStat; i += kBlock) {
      float s = 0.f;
      for (int w = 0; w < 4; ++w) s += s_red[w * kHeadJC * kHeadNStat + i];
      part[static_cast<size_t>(j0) * kHeadNStat + i] = s;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_head_bwd_act(const Group<HeadArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  head_bwd_act_body<T>(G.lv[l], local, smem);
}

// k_head_bwd_fin: workgroup = one hidden channel: its 12 sums over all tiles (16 strides, fixed order, double), then dgamma, dbeta,
//   dW_h[j,:], (channel 0) db_h and the constants k_head_bwd_gx / _gw need: k = gamma*rstd, mean, rstd, dbeta/n, dgamma/n (0, 0 in eval)
__global__ __launch_bounds__(kBlock) void k_head_bwd_fin(const Group<HeadArgs> G) {
  __shared__ double red[16][17];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const HeadArgs& A = G.lv[l];
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x, s = tid & 15, gq = tid >> 4;
  const int j = local;
  double acc = 0.0;
  if (s < kHeadNStat)
    for (int w = gq; w < A.nwg1; w += 16) acc += static_cast<double>(A.s.part1[(static_cast<size_t>(w) * g.hidp + j) * kHeadNStat + s]);
  red[gq][s] = acc;
  __syncthreads();
  if (tid < kHeadNStat) {
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += red[q][tid];
    red[0][tid] = t;                                            // (each thread overwrites only its own column)
  }
  __syncthreads();
  if (tid == 0) {
    const double n = static_cast<double>(g.B) * g.HW;
    const float gb = static_cast<float>(red[0][0]), gg = static_cast<float>(red[0][1]);
    A.gbeta[j] = gb; A.ggamma[j] = gg;
    for (int q = 0; q < 9; ++q) A.gwh[static_cast<size_t>(j) * 9 + q] = static_cast<float>(red[0][2 + q]);
    if (j == 0) A.gbh[0] = static_cast<float>(red[0][11]);
    const float rstd = A.c.rstd[j];
    A.s.kst[j] = A.p.gamma[j] * rstd;
    A.s.kst[g.hidp + j] = A.c.mean[j];
    A.s.kst[2 * g.hidp + j] = rstd;
    A.s.kst[3 * g.hidp + j] = g.training ? static_cast<float>(red[0][0] / n) : 0.f;
    A.s.kst[4 * g.hidp + j] = g.training ? static_cast<float>(red[0][1] / n) : 0.f;
  }
  if (local == 0) {                                             // padding channels: zero constants (their g_z is never used: weights are 0)
    for (int jj = g.hid + tid; jj < g.hidp; jj += kBlock)
      for (int q = 0; q < 5; ++q) A.s.kst[q * g.hidp + jj] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_head_bwd_gw: dW1[j,c] = sum_{b,px} g_z[b,j,px] x[b,c,px]      (M = hid, N = C, K = pixels)
//   A = g_z (lane l: hidden channel l%16 of its tile, pixel slot l/16), B = x (lane l: pixel slot l/16, channel l%16 of its tile): both are
//   "16 channels x 16 pixels" loads (16 B per lane with VEC = 4: register r = pixel 4*(l/16)+r of the 16; K sub-step r uses register r
//   of both operands).  Workgroup = (channel block of kHeadCB channels, one of kHeadGwWG pixel shares); its 4 waves take different
//   16*VEC... pixel chunks, accumulate all (hid tile, channel tile) products of the block and are summed through LDS at the end.
//   Accumulators: MTB x 4 tiles per wave (MTB = min(MT, 4): hid > 64 runs in passes of 64 hidden channels, re-reading x).
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__device__ __forceinline__ void head_bwd_gw_body(const HeadArgs& A, const int wg, float* smem) {
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const int cb = wg / kHeadGwWG, share = wg - cb * kHeadGwWG;
  const int c0 = cb * kHeadCB;
  const int MT = g.hidp >> 4;
  constexpr int NT = kHeadCB / 16;
  constexpr int CHP = 4 * VEC;                                  // pixels per K step (4 slots x VEC)
  const int nch = (g.HW + CHP - 1) / CHP;                       // chunks per sample
  const long long total = static_cast<long long>(g.B) * nch;
  float* s_kst = smem;                                          // [5][hidp]
  float* s_acc = smem + 5 * g.hidp;                             // [4 waves][16 x 16 tile] staging for the cross-wave sum
  for (int i = tid; i < 5 * g.hidp; i += kBlock) s_kst[i] = A.s.kst[i];
  __syncthreads();
  float* outp = A.s.gwpart + static_cast<size_t>(wg) * g.hidp * kHeadCB;
  for (int mt0 = 0; mt0 < MT; mt0 += kHeadMTW) {
    const int mtn = min(kHeadMTW, MT - mt0);
    v4f32 acc[kHeadMTW][NT];
#pragma unroll
    for (int t = 0; t < kHeadMTW; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][n] = v4f32{0.f, 0.f, 0.f, 0.f};
    float kj[kHeadMTW], mean[kHeadMTW], rstd[kHeadMTW], gbn[kHeadMTW], ggn[kHeadMTW];
#pragma unroll
    for (int t = 0; t < kHeadMTW; ++t) {
      const int j = min((mt0 + t) * 16 + lr, g.hidp - 1);
      kj[t] = s_kst[j]; mean[t] = s_kst[g.hidp + j]; rstd[t] = s_kst[2 * g.hidp + j]; gbn[t] = s_kst[3 * g.hidp + j]; ggn[t] = s_kst[4 * g.hidp + j];
    }
    for (long long ch = static_cast<long long>(share) * 4 + wave; ch < total; ch += 4ll * kHeadGwWG) {
      const int b = static_cast<int>(ch / nch);
      const int px = (static_cast<int>(ch - static_cast<long long>(b) * nch)) * CHP + lq * VEC;
      const bool ok = px < g.HW;
      const size_t pxo = ok ? px : 0;
      float av[kHeadMTW][VEC], bv[NT][VEC];
#pragma unroll
      for (int t = 0; t < kHeadMTW; ++t) {
        const int j = (mt0 + t) * 16 + lr;
#pragma unroll
        for (int r = 0; r < VEC; ++r) av[t][r] = 0.f;
        if (t < mtn && ok && j < g.hid) {
          float ga[VEC], zv[VEC];
          const size_t o = (static_cast<size_t>(b) * g.hid + j) * g.HW + pxo;
          load_vec<float, VEC>(A.s.ga + o, ga);
          load_vec<float, VEC>(A.c.z + o, zv);
#pragma unroll
          for (int r = 0; r < VEC; ++r) av[t][r] = kj[t] * (ga[r] - gbn[t] - (zv[r] - mean[t]) * rstd[t] * ggn[t]);
        }
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int c = c0 + n * 16 + lr;
#pragma unroll
        for (int r = 0; r < VEC; ++r) bv[n][r] = 0.f;
        if (ok && c < g.C) load_vec<T, VEC>(static_cast<const T*>(A.x) + (static_cast<size_t>(b) * g.C + c) * g.HW + pxo, bv[n]);
      }
#pragma unroll
      for (int t = 0; t < kHeadMTW; ++t) {
        if (t < mtn) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < VEC; ++r) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][r], bv[n][r], acc[t][n], 0, 0, 0);
        }
      }
    }
    // cross-wave sum, one 16 x 16 tile at a time (fixed order: wave 0..3); D layout: lane l, register v: row 4*(l/16)+v, column l%16
#pragma unroll
    for (int t = 0; t < kHeadMTW; ++t) {
      if (t >= mtn) break;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 4; ++v) s_acc[wave * 256 + (lq * 4 + v) * 16 + lr] = acc[t][n][v];
        __syncthreads();
        const int row = tid >> 4, col = tid & 15;
        const float s = (s_acc[tid] + s_acc[256 + tid]) + (s_acc[512 + tid] + s_acc[768 + tid]);
        outp[static_cast<size_t>((mt0 + t) * 16 + row) * kHeadCB + n * 16 + col] = s;
      }
    }
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_head_bwd_gw(const Group<HeadArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  head_bwd_gw_body<T, VEC>(G.lv[l], local, smem);
}

// k_head_bwd_gwf: dW1[j,c] = sum over the kHeadGwWG pixel shares (fixed order)
__global__ __launch_bounds__(kBlock) void k_head_bwd_gwf(const Group<HeadArgs> G) {
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const HeadArgs& A = G.lv[l];
  const HeadGeo& g = A.g;
  const int idx = local * kBlock + threadIdx.x;
  if (idx >= g.hid * g.C) return;
  const int j = idx / g.C, c = idx - j * g.C;
  const int cb = c / kHeadCB, cc = c - cb * kHeadCB;
  const float* p = A.s.gwpart + (static_cast<size_t>(cb) * kHeadGwWG * g.hidp + j) * kHeadCB + cc;
  float s = 0.f;
#pragma unroll 8
  for (int w = 0; w < kHeadGwWG; ++w) s += p[static_cast<size_t>(w) * g.hidp * kHeadCB];
  A.gw1[idx] = s;
}

}  // namespace mgacbam
