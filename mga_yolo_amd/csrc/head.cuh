// MGAMaskHead on the device: the producer of the mask logits MaskCBAM consumes.
//   reference: mga_yolo/nn/modules/segmentation.py:56-110 (default configuration: norm="bn", act=SiLU, dropout=0, out_channels=1)
//       z = Conv1x1(x)  (hid x C GEMM per pixel, no bias)  ->  BatchNorm2d (batch statistics in training)  ->  SiLU  ->  Conv3x3 + bias
//   layer-loop hand-off: mga_yolo/model/model.py:57-74 (the logits go to MaskCBAM as `mask` and to the segmentation loss)
//
// forward  (3 launches)  k_head_gemm<F> x (1 read)   -> z (B,hid,HW) fp32 + per-workgroup partial sums of z, z^2           [MFMA fp32 16x16x4]
//                        k_head_stats   partials     -> mean, rstd (+ running statistics, num_batches_tracked)               [tiny]
//                        k_head_out     z            -> s = SiLU(gamma*zhat+beta), staged with a 1-px halo in LDS -> 3x3 conv -> logits
// backward (5 launches)  k_head_bwd_act z, g_logits  -> g_a (B,hid,HW) = convT3x3(g) * SiLU'(a); partials of sum g_a, sum g_a*zhat,
//                                                        dW_h (9 taps), db_h
//                        k_head_bwd_fin partials     -> dgamma, dbeta, dW_h, db_h and the per-channel constants of g_z
//                        k_head_gemm<X> g_a, z       -> g_z on the fly -> gx (+)= W1^T g_z  (1 write, 1 read when accumulating)  [MFMA]
//                        k_head_bwd_gw2 g_a, z, x    -> partials of dW1 = sum_px g_z x^T  (x: 1 read; operands through LDS; _gw: direct) [MFMA]
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
typedef float v2f32 __attribute__((ext_vector_type(2)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef _Float16 v4f16 __attribute__((ext_vector_type(4)));
typedef __bf16 v4bf16 __attribute__((ext_vector_type(4)));

// Half-precision features take the matrix cores' NATIVE rate: v_mfma_f32_16x16x16_{f16,bf16} (K = 16 per instruction at 16 cycles per
// SIMD against K = 4 at 32 for the fp32 form: 8x the MAC rate), fp32 accumulation.  Lane l supplies row / column l % 16 and the FOUR
// consecutive k 4 (l / 16) .. +3 of each operand -- the grouping the fp32 kernels already load in (a lane's 16-byte weight vector, its
// four channel rows), so the operands are the fp32 values the loads produced, packed pairwise (v_cvt_pk_*): feature values are half
// already (exact), the fp32 weights and g_z are rounded to the feature type (2^-11 / 2^-8 relative, inside the half-precision bars).
template <typename T> struct HalfMma { static constexpr bool on = false; };
#ifndef MGAHEAD_HALF_MMA
#define MGAHEAD_HALF_MMA 1      // A/B builds: 0 = half features through the fp32 MFMA (round 2's path)
#endif
template <> struct HalfMma<__half> {
  static constexpr bool on = MGAHEAD_HALF_MMA != 0;
  __device__ static __forceinline__ v4f32 mma(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3, v4f32 c) {
    const v4f16 a = {static_cast<_Float16>(a0), static_cast<_Float16>(a1), static_cast<_Float16>(a2), static_cast<_Float16>(a3)};
    const v4f16 b = {static_cast<_Float16>(b0), static_cast<_Float16>(b1), static_cast<_Float16>(b2), static_cast<_Float16>(b3)};
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  }
};
template <> struct HalfMma<bf16_t> {
  static constexpr bool on = MGAHEAD_HALF_MMA != 0;
  __device__ static __forceinline__ v4f32 mma(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3, v4f32 c) {
    const v4bf16 a = {static_cast<__bf16>(a0), static_cast<__bf16>(a1), static_cast<__bf16>(a2), static_cast<__bf16>(a3)};
    const v4bf16 b = {static_cast<__bf16>(b0), static_cast<__bf16>(b1), static_cast<__bf16>(b2), static_cast<__bf16>(b3)};
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(v4i16, a), __builtin_bit_cast(v4i16, b), c, 0, 0, 0);
  }
};

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
// ctx (saved for backward): z (B,hid,HW) fp32 | mean[hidp] | rstd[hidp] | par [hidp][16] | part (nwg, 2, hidp)
//   par[j] = {scale = gamma*rstd, shift = beta - mean*scale, mean, rstd, W_h[j][0..8]}: written by k_head_stats, read with SCALAR loads
//   (uniform address -> s_load / SGPR operands) by k_head_out and k_head_bwd_act instead of LDS broadcasts
struct HeadCtx { float* z; float* mean; float* rstd; float* par; float* part; };
constexpr int kHeadPar = 16;
// scratch (backward transients): g_a (B,hid,HW) | part1 (nwg1, hidp, 12) | kst (5, hidp) | gwpart (ncb*nshare, hidp, kHeadCB)
struct HeadScratch { float* ga; float* part1; float* kst; float* gwpart; };

struct HeadArgs {
  const void* x; void* logits;                 // forward
  const void* gl; void* gx;                    // backward: dL/dlogits (B,1,H,W) T, dL/dx (B,C,H,W) T
  const float* gl2;                            // optional second addend of dL/dlogits (fp32), or null
  float* gw1; float* ggamma; float* gbeta; float* gwh; float* gbh;
  HeadPtrs p; HeadCtx c; HeadScratch s; HeadGeo g;
  int tile_px, tiles_per_sample, nwg;          // 1-D pixel tiling of k_head_gemm<FWD> (nwg = rows of the partial sums)
  int gx_tile_px, gx_tiles_per_sample;         // ... of k_head_gemm<GX>
  int fw_kw, gx_kw;                            // waves of a workgroup that split the K steps (1, 2, 4)
  int fw_mtw, gx_mtw;                          // M tiles per wave (template parameter of the launch the level rides in)
  int nwg_out, out_ppt, out_px, out_per;       // k_head_out: workgroups, staged pixels per thread, outputs per workgroup, workgroups per sample
  int nwg1, act_ppt, act_hl_max;               // k_head_bwd_act: pixel runs of the level, pixels per thread, LDS floats of the launch's longest run
  int ncb, nshare, gw2;                        // k_head_bwd_gw[2]: channel blocks, pixel shares per block (= partial sets of dW1), LDS-staged form
  long long* trace;                            // MGACBAM_TRACE builds only (tools/trace_head.py), else nullptr
  int trace_base;                              // first trace slot of the launch the level rides in
  int accum_gx;                                // MGAHEAD_BWD_ACCUM_GX: gx += W1^T g_z
};

constexpr int kHeadMTW = 4;      // upper bound of the 16-output tiles a wave accumulates at once (template MTW = 2 or 4: 16 registers x VEC/4... each;
                                 // forward: 2 up to hidden = 128 (every level of a call in one launch, 4 workgroups per CU), 4 beyond; gx: 2)
constexpr int kHeadLdsX = 4096;  // floats of LDS for the K-split reduction of k_head_gemm (4 waves x VEC x 4 registers x 64 lanes)
constexpr int kHeadJC = 4;       // k_head_bwd_act: hidden channels per workgroup (= one batch of z loads: every workgroup is one memory round trip deep)
constexpr int kHeadCB = 64;      // k_head_bwd_gw: channels of x per workgroup (4 N tiles)
constexpr int kHeadNStat = 12;   // per-channel partial sums of k_head_bwd_act: g_a, g_a*zhat, 9 taps of dW_h, db_h
// SiLU on the streaming paths: v_exp_f32 + v_rcp_f32 (common.cuh: sigmoid_fast, relative error ~1e-6); the accurate expf and IEEE
// division cost ~25 VALU slots per element and made k_head_out / k_head_bwd_act VALU-bound (a wave64 instruction takes 4 cycles)
__device__ __forceinline__ float siluf(float a) { return a * sigmoid_fast(a); }

// ---------------------------------------------------------------------------------------------------------------------------
// The two pixel-parallel GEMMs:   FWD  z[b,j,px]  = sum_c W1[j,c] x[b,c,px]      (M = hid, K = C)          segmentation.py:81
//                                 GX   gx[b,c,px] = sum_j W1[j,c] g_z[b,j,px]    (M = C,   K = hid)        its autograd backward
//   v_mfma_f32_16x16x4_f32: A = weight tile (16 outputs x 4 k; lane l: output l%16, k l/16), B = activation tile (4 k x 16 pixel groups;
//   lane l: k l/16, group l%16), D 16 x 16 in 4 registers (lane l, register v: output 4*(l/16)+v, group l%16).  With VEC = 4 a lane loads
//   16 B = 4 consecutive pixels of its k-channel (a wave-load = 4 channels x 256 B contiguous) and register r of that vector is the B
//   operand of sub-tile r: 64 pixels per wave and K step; VEC = 1 (H*W % 4 != 0): one dword per lane, 16 pixels per wave.
//   Workgroup = 4 waves = MW (along M) x KW (along K) x PW (along pixels); M in blocks of 16*MTW*MW outputs.  The weights are
//   read from global memory directly in the A layout, batch by batch together with the activations (no LDS staging, no barriers);
//   forward: as 16-byte loads, the K order permuted inside each group of 16 channels (see wperm below).
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int VEC, bool GX, int MTW>
__device__ __forceinline__ void head_gemm_body(const HeadArgs& A, const int wg, float* smem) {
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lk = lane >> 4, ln = lane & 15;
  const int M = GX ? g.C : g.hid, K = GX ? g.hid : g.C;
  const int MT = ((GX ? g.cp : g.hidp)) >> 4;
  const int MW = min(4, (MT + MTW - 1) / MTW);        // 1, 2 or 4 (3 -> 4)
  const int MWp = MW == 3 ? 4 : MW;
  // KW waves split the K steps (levels with few pixels and a long K: the workgroup's chain of dependent memory round trips shrinks
  // KW-fold; their accumulators are summed through LDS, fixed order, before the epilogue)
  const int KW = GX ? A.gx_kw : A.fw_kw;
  const int PW = 4 / (MWp * KW);
  const int mw = wave % MWp, kwi = (wave / MWp) % KW, pw = wave / (MWp * KW);
  constexpr int WPX = 16 * VEC;
  const int tps = GX ? A.gx_tiles_per_sample : A.tiles_per_sample;
  const int b = wg / tps, tile = wg - b * tps;
  const int px = tile * (GX ? A.gx_tile_px : A.tile_px) + pw * WPX + ln * VEC;   // first pixel of this lane
  const bool px_ok = px < g.HW;
  const size_t pxo = px_ok ? px : 0;
  const int mblk = MTW * MWp;                              // M tiles per block
  float* s_kst = smem;                                          // GX: per-hidden-channel constants of g_z [5][hidp]
  float* s_sum = smem;                                          // FWD: [PW][2][hidp] tile sums
  float* s_x = smem + 8 * g.hidp;                               // K-split reduction scratch [4 waves][VEC][4][64]
  if (GX) {
    for (int i = tid; i < 5 * g.hidp; i += kBlock) s_kst[i] = A.s.kst[i];
    __syncthreads();
  }
  const T* xb = GX ? nullptr : static_cast<const T*>(A.x) + static_cast<size_t>(b) * g.C * g.HW + pxo;
  const float* gab = GX ? A.s.ga + static_cast<size_t>(b) * g.hid * g.HW + pxo : nullptr;
  const float* zb = A.c.z + static_cast<size_t>(b) * g.hid * g.HW + pxo;
  // forward with C % 4 == 0: the K order inside a group of 4 steps (16 channels) is permuted so that a lane's weights of the group are
  // 4 CONSECUTIVE floats of a W1 row -- lane (lk, ln) at step 4q + r takes channel 16q + 4lk + r -- one 16-byte load instead of four
  // gathers of 16 cache lines each (those made the TA, not HBM, the limit of the levels with long K); groups are whole per wave
  const bool wperm = !GX && (g.C & 3) == 0;
  // half-precision gx: the same grouping of K (a lane's four CONSECUTIVE hidden channels per group of 16), so that four K steps of a lane
  // are one bf16 / fp16 MFMA operand (HalfMma above); its weights W1[k][out] stay per-element gathers, as in the fp32 form
  const bool half_gx = GX && HalfMma<T>::on;
  const bool grp = wperm || half_gx;
  const int ksteps = grp ? ((K + 15) >> 4) << 2 : (K + 3) >> 2;
  int kper_ = (ksteps + KW - 1) / KW;                           // this wave's share of the K steps: [kbeg, kend)
  if (grp) kper_ = (kper_ + 3) & ~3;
  const int kper = kper_;
  const int kbeg = kwi * kper, kend = min(ksteps, kbeg + kper);
  const int gid = A.trace_base + blockIdx.x;
  TRACE_HWID(A.trace, gid);
  TRACE_MARK(A.trace, gid, 0);                                  // start
  const int n_wave = GX ? 0 : max(0, min(WPX, g.HW - (tile * A.tile_px + pw * WPX)));   // real pixels of this wave's window (forward statistics)

  for (int mt0 = 0; mt0 < MT; mt0 += mblk) {
    const int mtn = min(mblk, MT - mt0);                        // tiles in this block
    v4f32 acc[MTW][VEC];
#pragma unroll
    for (int t = 0; t < MTW; ++t)
#pragma unroll
      for (int r = 0; r < VEC; ++r) acc[t][r] = v4f32{0.f, 0.f, 0.f, 0.f};
    // K steps in batches of KU.  BOTH operands of a batch are requested before its first MFMA: the activations (16-B loads, 4 channels x
    // 256 B per wave-load) and the weights, read from global memory straight in the MFMA lane layout (W1 is <= 64 KB and shared by
    // every workgroup: L1 / L2 hits) -- no LDS staging, no barrier in the loop; a workgroup is one or two memory round trips deep.
    constexpr int KU = (GX || MTW == 4) ? 4 : 8;                 // (MTW = 4 with 8 steps in flight needs 156 VGPRs: 2 workgroups per CU)
    for (int ks0 = kbeg; ks0 < kend; ks0 += KU) {
      float bv[KU][VEC], zq[GX ? KU : 1][VEC], aw[KU][MTW];
      if (wperm) {
#pragma unroll
        for (int gi = 0; gi < KU / 4; ++gi) {
          const int kk0 = ((ks0 >> 2) + gi) * 16 + 4 * lk;      // this lane's 4 channels of the group
          const bool gok = ks0 + 4 * gi < kend && kk0 < K;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) bv[4 * gi + r][q] = 0.f;
            if (gok && px_ok) load_vec<T, VEC>(xb + static_cast<size_t>(kk0 + r) * g.HW, bv[4 * gi + r]);
          }
#pragma unroll
          for (int t = 0; t < MTW; ++t) {
            const int mt = mw * MTW + t;
            const int out = (mt0 + mt) * 16 + ln;
            float w4[4] = {0.f, 0.f, 0.f, 0.f};
            if (gok && mt < mtn && out < M) load_vec<float, 4>(A.p.w1 + static_cast<size_t>(out) * g.C + kk0, w4);
#pragma unroll
            for (int r = 0; r < 4; ++r) aw[4 * gi + r][t] = w4[r];
          }
        }
      } else if (half_gx) {
#pragma unroll
        for (int gi = 0; gi < KU / 4; ++gi) {
          const int kk0 = ((ks0 >> 2) + gi) * 16 + 4 * lk;      // this lane's 4 hidden channels of the group
          const bool gok = ks0 + 4 * gi < kend;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kk = kk0 + r, u = 4 * gi + r;
            const bool kok = gok && kk < K;
#pragma unroll
            for (int q = 0; q < VEC; ++q) { bv[u][q] = 0.f; zq[GX ? u : 0][q] = 0.f; }
            if (kok && px_ok) {
              load_vec<float, VEC>(gab + static_cast<size_t>(kk) * g.HW, bv[u]);
              load_vec<float, VEC>(zb + static_cast<size_t>(kk) * g.HW, zq[GX ? u : 0]);
            }
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
              const int mt = mw * MTW + t;
              const int out = (mt0 + mt) * 16 + ln;
              aw[u][t] = (kok && mt < mtn && out < M) ? A.p.w1[static_cast<size_t>(kk) * g.C + out] : 0.f;
            }
          }
        }
      } else {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int kk = (ks0 + u) * 4 + lk;                      // this lane's k channel
        const bool kok = (ks0 + u) < kend && kk < K;
#pragma unroll
        for (int r = 0; r < VEC; ++r) bv[u][r] = 0.f;
        if (GX) {
#pragma unroll
          for (int r = 0; r < VEC; ++r) zq[u][r] = 0.f;
          if (kok && px_ok) {
            load_vec<float, VEC>(gab + static_cast<size_t>(kk) * g.HW, bv[u]);
            load_vec<float, VEC>(zb + static_cast<size_t>(kk) * g.HW, zq[u]);
          }
        } else if (kok && px_ok) {
          load_vec<T, VEC>(xb + static_cast<size_t>(kk) * g.HW, bv[u]);
        }
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
          const int mt = mw * MTW + t;
          const int out = (mt0 + mt) * 16 + ln;
          aw[u][t] = 0.f;
          if (kok && mt < mtn && out < M)
            aw[u][t] = GX ? A.p.w1[static_cast<size_t>(kk) * g.C + out] : A.p.w1[static_cast<size_t>(out) * g.C + kk];
        }
      }
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (ks0 + u < kend) {                                   // uniform per wave
          if (GX) {                                             // g_z = k_j (g_a - gbeta_j/n - zhat gamma'_j/n), zhat = (z - mean) rstd
            const int kk = min(half_gx ? ((ks0 + u) >> 2) * 16 + 4 * lk + ((ks0 + u) & 3) : (ks0 + u) * 4 + lk, g.hidp - 1);
            const float kj = s_kst[kk], mean = s_kst[g.hidp + kk], rstd = s_kst[2 * g.hidp + kk];
            const float gbn = s_kst[3 * g.hidp + kk], ggn = s_kst[4 * g.hidp + kk];
#pragma unroll
            for (int r = 0; r < VEC; ++r) bv[u][r] = kj * (bv[u][r] - gbn - (zq[u][r] - mean) * rstd * ggn);
            // (lanes whose loads were skipped hold g_a = z = 0: their g_z meets a ZERO weight, or its result row is never stored)
          }
          if (!(HalfMma<T>::on && grp)) {                       // fp32 MFMA: one K step (4 channels over the lane groups) per instruction
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
              if (mw * MTW + t < mtn) {                       // uniform per wave
#pragma unroll
                for (int r = 0; r < VEC; ++r) acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[u][t], bv[u][r], acc[t][r], 0, 0, 0);
              }
            }
          }
        }
      }
      if constexpr (HalfMma<T>::on) {
        if (grp) {                                              // half-precision features: four K steps of a lane = ONE 16x16x16 MFMA
#pragma unroll
          for (int gi = 0; gi < KU / 4; ++gi) {
            if (ks0 + 4 * gi < kend) {                          // uniform per wave
#pragma unroll
              for (int t = 0; t < MTW; ++t) {
                if (mw * MTW + t < mtn) {
#pragma unroll
                  for (int r = 0; r < VEC; ++r)
                    acc[t][r] = HalfMma<T>::mma(aw[4 * gi][t], aw[4 * gi + 1][t], aw[4 * gi + 2][t], aw[4 * gi + 3][t],
                                                bv[4 * gi][r], bv[4 * gi + 1][r], bv[4 * gi + 2][r], bv[4 * gi + 3][r], acc[t][r]);
                }
              }
            }
          }
        }
      }
    }
#ifdef MGACBAM_TRACE
    { float tsum = 0.f;
#pragma unroll
      for (int t = 0; t < MTW; ++t) tsum += acc[t][0][0];
      if (tsum == 1.2345e30f) s_x[0] = tsum; }                   // force the MFMA results before the stamp
    TRACE_MARK(A.trace, gid, 2);                                // K loop done (operands arrived, MFMAs issued)
#endif
    // ---- K split: sum the KW waves' accumulators (wave kwi = 0 keeps the result) ------------------------------------------------------
    if (KW > 1) {
#pragma unroll
      for (int t = 0; t < MTW; ++t) {
        __syncthreads();
        if (kwi > 0) {
#pragma unroll
          for (int r = 0; r < VEC; ++r)
#pragma unroll
            for (int v = 0; v < 4; ++v) s_x[((wave * VEC + r) * 4 + v) * 64 + lane] = acc[t][r][v];
        }
        __syncthreads();
        if (kwi == 0) {
          for (int q = 1; q < KW; ++q) {
            const int ow = wave + q * MWp;                      // the wave with the same (mw, pw) and kwi = q
#pragma unroll
            for (int r = 0; r < VEC; ++r)
#pragma unroll
              for (int v = 0; v < 4; ++v) acc[t][r][v] += s_x[((ow * VEC + r) * 4 + v) * 64 + lane];
          }
        }
      }
    }
    TRACE_MARK(A.trace, gid, 3);                                // K split summed
    // ---- epilogue of this M block -------------------------------------------------------------------------------------------
    float oldv[GX ? MTW : 1][GX ? 4 : 1][VEC];             // GX + accumulate: every old value is requested before the first store
    if (GX && A.accum_gx && kwi == 0) {
#pragma unroll
      for (int t = 0; t < MTW; ++t) {
        const int mt = mw * MTW + t;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int out = (mt0 + mt) * 16 + lk * 4 + v;
#pragma unroll
          for (int r = 0; r < VEC; ++r) oldv[GX ? t : 0][GX ? v : 0][r] = 0.f;
          if (mt < mtn && out < M && px_ok)
            load_vec<T, VEC>(static_cast<const T*>(A.gx) + (static_cast<size_t>(b) * g.C + out) * g.HW + px, oldv[GX ? t : 0][GX ? v : 0]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
      const int mt = mw * MTW + t;
      if (mt < mtn && kwi == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int out = (mt0 + mt) * 16 + lk * 4 + v;
          float ov[VEC];
          float s1 = 0.f;
#pragma unroll
          for (int r = 0; r < VEC; ++r) {
            ov[r] = acc[t][r][v];
            s1 += ov[r];                                         // pixels past H*W were loaded as zeros: they add nothing
          }
          if (out < M && px_ok) {
            if (GX) {
              T* gp = static_cast<T*>(A.gx) + (static_cast<size_t>(b) * g.C + out) * g.HW + px;
              if (A.accum_gx) {                                  // the feature's other consumer (MaskCBAM) already left its gradient here
#pragma unroll
                for (int r = 0; r < VEC; ++r) ov[r] += oldv[GX ? t : 0][GX ? v : 0][r];
              }
              store_vec_stream<T, VEC>(gp, ov, true);
            }
            else store_vec<float, VEC>(A.c.z + (static_cast<size_t>(b) * g.hid + out) * g.HW + px, ov);
          }
          if (!GX && g.training) {
            // batch statistics, Chan-style: the wave's sum and its squared deviations about ITS OWN mean (the values are still in
            // registers, so the second pass costs one more 16-lane sum).  sum z^2 - n mean^2 in fp32 cancels catastrophically for a
            // channel with |mean| >> std (relative error ~1e-7 mean^2 / var in rstd, every gradient and running_var); torch uses Welford
            s1 = wave_group_sum(s1, 16);                         // over the 16 pixel-group lanes that share this output channel
            const float mw_ = n_wave > 0 ? s1 / static_cast<float>(n_wave) : 0.f;
            float m2 = 0.f;
            if (px_ok) {
#pragma unroll
              for (int r = 0; r < VEC; ++r) { const float d = acc[t][r][v] - mw_; m2 += d * d; }
            }
            m2 = wave_group_sum(m2, 16);
            if (ln == 0) { s_sum[(pw * 2 + 0) * g.hidp + out] = s1; s_sum[(pw * 2 + 1) * g.hidp + out] = m2; }
          }
        }
      }
    }
  }
#ifdef MGACBAM_TRACE
  TRACE_MARK(A.trace, gid, 5);                                  // stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TRACE_MARK(A.trace, gid, 10);                                 // stores complete
#endif
  if (!GX && g.training) {                                      // this tile's (sum, M2 about the tile mean) per output channel
    __syncthreads();
    float* part = A.c.part + static_cast<size_t>(wg) * 2 * g.hidp;
    for (int i = tid; i < g.hidp; i += kBlock) {
      float n = 0.f, S = 0.f, M2 = 0.f;
      for (int w = 0; w < PW; ++w) {                            // the pixel waves' partials, combined pairwise (Chan et al.), fixed order
        const float nb = static_cast<float>(max(0, min(WPX, g.HW - (tile * A.tile_px + w * WPX))));
        if (nb > 0.f) {
          const float Sb = s_sum[(w * 2 + 0) * g.hidp + i], Mb = s_sum[(w * 2 + 1) * g.hidp + i];
          if (n == 0.f) { S = Sb; M2 = Mb; n = nb; }
          else { const float d = Sb / nb - S / n; M2 += Mb + d * d * (n * nb / (n + nb)); S += Sb; n += nb; }
        }
      }
      part[i] = S; part[g.hidp + i] = M2;
    }
  }
}

template <typename T, int VEC, bool GX, int MTW>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((MTW == 4 || GX) ? 3 : 4))) void k_head_gemm(const Group<HeadArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  head_gemm_body<T, VEC, GX, MTW>(G.lv[l], local, smem);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_head_stats: batch statistics of z per hidden channel from the tile partials (fixed order, double accumulation), BatchNorm's
//   running-statistics update (segmentation.py:83 -> torch BatchNorm2d: biased variance for the normalisation, unbiased for the
//   running estimate, momentum m).  Eval mode: mean / rstd from the running statistics.  Workgroup = one channel, 256 strides over the tiles.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_head_stats(const Group<HeadArgs> G) {
  __shared__ double red[2][kBlock];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const HeadArgs& A = G.lv[l];
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x;                                  // workgroup = one hidden channel: 256 strides over the tiles
  const int j = local;
  // everything thread 0 needs for the tail is requested NOW, beside the partials (after the reduction each load would be a round trip
  // of its own: the stores to the running statistics between them keep the compiler from batching)
  float rm0 = 0.f, rv0 = 0.f, gam = 0.f, bet = 0.f, whv[9];
  long long nbt0 = 0;
#pragma unroll
  for (int q = 0; q < 9; ++q) whv[q] = 0.f;
  if (tid == 0) {
    rm0 = A.p.rmean[j]; rv0 = A.p.rvar[j]; gam = A.p.gamma[j]; bet = A.p.beta[j];
#pragma unroll
    for (int q = 0; q < 9; ++q) whv[q] = A.p.wh[static_cast<size_t>(j) * 9 + q];
    if (j == 0 && A.p.nbt && g.training) nbt0 = *A.p.nbt;
  }
  // per tile w: (S_w, M2_w about the tile's own mean, n_w pixels).  Over all tiles:  sum (x - m)^2 = sum_w M2_w + sum_w S_w^2 / n_w - S^2 / n,
  // accumulated in double (the last two terms cancel to ~1e-16 mean^2 / var relative, harmless)
  double acc1 = 0.0, acc2 = 0.0;
  if (g.training) {
    const float* p = A.c.part + j;
    const size_t stride = static_cast<size_t>(2) * g.hidp;
    constexpr int U = 4;
    auto tile_n = [&](int w) { const int t = w % A.tiles_per_sample; return static_cast<double>(min(A.tile_px, g.HW - t * A.tile_px)); };
    int w = tid;
    for (; w + (U - 1) * kBlock < A.nwg; w += U * kBlock) {     // U independent load pairs in flight
      float sv[U], mv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { sv[u] = p[(w + u * kBlock) * stride]; mv[u] = p[(w + u * kBlock) * stride + g.hidp]; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double sd = static_cast<double>(sv[u]);
        acc1 += sd; acc2 += static_cast<double>(mv[u]) + sd * sd / tile_n(w + u * kBlock);
      }
    }
    for (; w < A.nwg; w += kBlock) {
      const double sd = static_cast<double>(p[w * stride]);
      acc1 += sd; acc2 += static_cast<double>(p[w * stride + g.hidp]) + sd * sd / tile_n(w);
    }
  }
  red[0][tid] = acc1; red[1][tid] = acc2;
  __syncthreads();
  for (int o = kBlock >> 1; o >= 1; o >>= 1) {                  // fixed-order tree
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    float mean, var;
    if (g.training) {
      const double n = static_cast<double>(g.B) * g.HW;
      const double m = red[0][0] / n;
      double v = (red[1][0] - red[0][0] * m) / n;                // biased variance: sum (x - m)^2 / n
      if (v < 0.0) v = 0.0;
      mean = static_cast<float>(m); var = static_cast<float>(v);
      const double unb = n > 1.0 ? v * n / (n - 1.0) : v;
      A.p.rmean[j] = (1.f - g.momentum) * rm0 + g.momentum * mean;
      A.p.rvar[j] = (1.f - g.momentum) * rv0 + g.momentum * static_cast<float>(unb);
      if (j == 0 && A.p.nbt) *A.p.nbt = nbt0 + 1;
    } else {
      mean = rm0; var = rv0;
    }
    const float rstd = 1.0f / sqrtf(var + g.eps);
    A.c.mean[j] = mean;
    A.c.rstd[j] = rstd;
    float* par = A.c.par + static_cast<size_t>(j) * kHeadPar;
    const float sc = gam * rstd;
    par[0] = sc; par[1] = bet - mean * sc; par[2] = mean; par[3] = rstd;
#pragma unroll
    for (int q = 0; q < 9; ++q) par[4 + q] = whv[q];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_head_out: logits = conv3x3(SiLU(gamma * zhat + beta)) + bias                                  segmentation.py:83-92
//   The 3x3 conv over `hidden` channels is linear in the taps:  logits[p] = sum_{u,v} t_uv[p + (u-1) W + (v-1)]  with the PER-PIXEL sums
//   t_uv[q] = sum_c W_h[c][u][v] * s_c[q]  -- nine numbers per pixel that need no neighbour.  So the activations never visit LDS: a thread
//   owns PPT consecutive staged pixels, loads ALL hidden channels of them straight into registers (16- / 8- / 4-byte loads, every one
//   requested before the first is used at the YOLOv8 widths: one memory round trip per workgroup) and accumulates 9 x PPT tap sums;
//   only the nine-plane shift-sum goes through LDS, in two steps: the horizontal one (a thread needs one tap of each neighbour: 6 values
//   per thread) gives h_u[r] = [x>0] t_u0[r-1] + t_u1[r] + [x<W-1] t_u2[r+1], the vertical one logits[p] = h_0[p-W] + h_1[p] + h_2[p+W]
//   (two planes of the run in LDS, 8 KB).  Workgroup = a run of 256 x PPT staged pixels of one sample = its outputs plus W+1 pixels either
//   side (zero outside the sample = the conv's padding in y; taps that would wrap a row end are the [x..] factors = padding in x).
//   PPT = 4 / 2 / 1 for hidden <= 16 / 32 / more, so every workgroup of a YOLOv8 pyramid carries the same 64 values per thread; rows so
//   wide that the halo eats the run take the next PPT.  (Round 2's form staged the ACTIVATIONS of a run in LDS, hidden/4 channels per
//   wave: 32 KB per workgroup, 1.6x halo overhead, 1,440 workgroups in two resident rounds of ~11 us each at config 2 -- 20.7 us for 23 MB.)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kHeadOutParCh = 128;                             // channels whose constants are staged in LDS at a time
__host__ __device__ inline int head_out_max(int ppt, int W) { return (kBlock * ppt - 2 * (W + 1) - 3) & ~3; }   // outputs a run can hold (3: the run starts on a 16-byte boundary)
// pixels per thread, outputs per workgroup (a multiple of 4) and workgroups per sample of a level; false: rows too wide (W > ~500)
__host__ __device__ inline bool head_out_shape(int hid, int HW, int W, int& ppt, int& opx, int& per) {
  ppt = hid <= 16 ? 4 : (hid <= 32 ? 2 : 1);
  while (ppt < 4 && head_out_max(ppt, W) < 64) ppt *= 2;
  const int mo = head_out_max(ppt, W);
  if (mo < 4) return false;
  per = (HW + mo - 1) / mo;
  opx = ((HW + per - 1) / per + 3) & ~3;
  return true;
}
template <typename T, int VEC, int PPT>
__device__ __forceinline__ void head_out_run(const HeadArgs& A, const int wg, float* s_par, float* s_ex, float* s_h) {
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x;
  constexpr int SLOTS = kBlock * PPT;
  constexpr int CB = 32 / PPT;                                  // channels per batch: 32 values per thread; two batches are requested ahead
  const int per = A.out_per, opx = A.out_px;
  const int b = wg / per, p0 = (wg - b * per) * opx, pend = min(p0 + opx, g.HW);
  int lo = p0 - (g.W + 1);
  if (VEC == 4) lo &= ~3;                                       // H*W % 4 == 0: every aligned group of PPT pixels is wholly inside or outside the sample
  const int s0 = lo + tid * PPT;                                // this thread's first staged pixel (index in the sample; may lie outside it)
  bool ins[PPT];
  unsigned keep[PPT];
  float ml[PPT], mr[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    ins[i] = s0 + i >= 0 && s0 + i < g.HW;
    keep[i] = ins[i] ? 0xffffffffu : 0u;
    const int x = (s0 + i + 8 * g.W) % g.W;                     // (s0 >= -(W + 4))
    ml[i] = x > 0 ? 1.f : 0.f; mr[i] = x < g.W - 1 ? 1.f : 0.f;
  }
  const int gid = 24576 + blockIdx.x;
  TRACE_HWID(A.trace, gid);
  TRACE_MARK(A.trace, gid, 0);
  const float* zb = A.c.z + static_cast<size_t>(b) * g.hid * g.HW;
  // every load is unconditional from a valid address (pixels outside the sample: the channel's first group, zeroed where the activation
  // is formed), as a raw buffer load: descriptor = the sample's z block, scalar offset = the channel row, ONE byte-offset VGPR per lane
  // for all rows (global loads kept a 64-bit VGPR address per row in flight -- 2 x 64 registers -- and no exec-mask branches either)
  unsigned so[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) so[i] = ins[i] ? static_cast<unsigned>(s0 + i) * 4u : 0u;
  const unsigned row_bytes = static_cast<unsigned>(g.HW) * 4u;
  const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zb), 0, static_cast<int>(row_bytes * static_cast<unsigned>(g.hid)), 0x00020000);
  auto fetch = [&](const int c0, float (&zv)[CB][PPT]) {
#pragma unroll
    for (int jj = 0; jj < CB; ++jj) {
      const unsigned ro = static_cast<unsigned>(min(c0 + jj, g.hid - 1)) * row_bytes;   // past the last channel: a valid row, weight 0 below
      if constexpr (VEC == 4 && PPT == 4) {
        const v4f32 f = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(zrs, so[0], ro, 0));   // (whole-vector cast: see bwd.cuh load_plane_agent)
        zv[jj][0] = f.x; zv[jj][1] = f.y; zv[jj][2] = f.z; zv[jj][3] = f.w;
      } else if constexpr (VEC == 4 && PPT == 2) {
        const v2f32 f = __builtin_bit_cast(v2f32, __builtin_amdgcn_raw_buffer_load_b64(zrs, so[0], ro, 0));
        zv[jj][0] = f.x; zv[jj][1] = f.y;
      } else {
#pragma unroll
        for (int i = 0; i < PPT; ++i) zv[jj][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zrs, so[i], ro, 0));
      }
    }
  };
  float t[9][PPT];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int i = 0; i < PPT; ++i) t[q][i] = 0.f;
  // constants {scale, shift, W_h[0..8]} of the channels: staged in LDS (chunks of kHeadOutParCh channels), read back as broadcasts two
  // channels ahead of their use.  That distance is ENFORCED: the LDS offset of channel j + 2 passes through an empty asm that takes a tap
  // sum of channel j as input.  Left alone the compiler hoists every channel's twelve reads above the arithmetic (400+ registers: one wave
  // per SIMD); scalar loads from the table instead (SGPR operands) serialise at ~1.4 us per channel under load (measured: 26-90 us).
  auto accumulate = [&](const int c0, const int ch0, const float (&zv)[CB][PPT]) {
    unsigned tk[CB + 2];
    tk[0] = 0u; tk[1] = 0u;
    asm volatile("" : "+v"(tk[0]), "+v"(tk[1]));
#pragma unroll
    for (int jj = 0; jj < CB; ++jj) {
      const float* pc = s_par + (c0 + jj - ch0) * 12 + tk[jj];   // uniform address: LDS broadcast (weights of channels past the last are 0)
      const float sc = pc[0], sh = pc[1];
      float w[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) w[q] = pc[2 + q];
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        // outside the sample: the conv's zero padding (of the ACTIVATION) -- as a bit mask: a select here became a branch per channel
        const float a = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, siluf(zv[jj][i] * sc + sh)) & keep[i]);
#pragma unroll
        for (int q = 0; q < 9; ++q) t[q][i] += w[q] * a;
      }
      tk[jj + 2] = tk[jj];
#pragma unroll
      for (int i = 0; i < PPT; ++i)                             // (every tap sum: the ones not named here were deferred, their operands parked in registers)
        asm volatile("" : "+v"(tk[jj + 2]) : "v"(t[0][i]), "v"(t[1][i]), "v"(t[2][i]), "v"(t[3][i]), "v"(t[4][i]), "v"(t[5][i]), "v"(t[6][i]), "v"(t[7][i]), "v"(t[8][i]));
    }
  };
  float z0[CB][PPT], z1[CB][PPT];
  for (int ch0 = 0; ch0 < g.hid; ch0 += kHeadOutParCh) {
    // the chunk's constants from the per-channel table, requested before the activations
    constexpr int NPC = kHeadOutParCh * 12 / kBlock;
    float pv[NPC];
#pragma unroll
    for (int u = 0; u < NPC; ++u) {
      const int idx = tid + u * kBlock, jj = idx / 12, q = idx - jj * 12;
      pv[u] = 0.f;
      if (ch0 + jj < g.hid && q < 11) pv[u] = A.c.par[static_cast<size_t>(ch0 + jj) * kHeadPar + (q < 2 ? q : q + 2)];
    }
    if (ch0 == 0) {
      fetch(0, z0);
      if (CB < g.hid) fetch(CB, z1);
    } else {
      __syncthreads();                                          // the previous chunk's constants have been read
    }
#pragma unroll
    for (int u = 0; u < NPC; ++u) s_par[tid + u * kBlock] = pv[u];
    __syncthreads();
    if (ch0 == 0) TRACE_MARK(A.trace, gid, 1);                   // constants staged
    const int chn = min(g.hid, ch0 + kHeadOutParCh);
    for (int c0 = ch0; c0 < chn; c0 += 2 * CB) {                // (kHeadOutParCh is a multiple of 2 * CB)
      accumulate(c0, ch0, z0);
      if (c0 + 2 * CB < g.hid) fetch(c0 + 2 * CB, z0);
      if (c0 + CB < g.hid) {
        accumulate(c0 + CB, ch0, z1);
        if (c0 + 3 * CB < g.hid) fetch(c0 + 3 * CB, z1);
      }
    }
  }
  TRACE_MARK(A.trace, gid, 2);                                  // tap sums done
  // horizontal step: t_u0 of the pixel left of the thread's first one, t_u2 of the pixel right of its last one
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    s_ex[u * kBlock + tid] = t[u * 3][PPT - 1];
    s_ex[(3 + u) * kBlock + tid] = t[u * 3 + 2][0];
  }
  __syncthreads();
  float h[3][PPT];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const float lft = tid > 0 ? s_ex[u * kBlock + tid - 1] : 0.f;             // (the run's first and last pixel are halo of halo: never read below)
    const float rgt = tid < kBlock - 1 ? s_ex[(3 + u) * kBlock + tid + 1] : 0.f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float tl = i > 0 ? t[u * 3][i > 0 ? i - 1 : 0] : lft;
      const float tr = i < PPT - 1 ? t[u * 3 + 2][i < PPT - 1 ? i + 1 : 0] : rgt;
      h[u][i] = ml[i] * tl + t[u * 3 + 1][i] + mr[i] * tr;
    }
  }
  // vertical step: rows y-1 and y+1 through LDS
  store_vec<float, PPT>(s_h + tid * PPT, h[0]);
  store_vec<float, PPT>(s_h + SLOTS + tid * PPT, h[2]);
  __syncthreads();
  const float bias = A.p.bh[0];
  T* lg = static_cast<T*>(A.logits) + static_cast<size_t>(b) * g.HW;
  float o[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid * PPT + i;
    const bool out = s0 + i >= p0 && s0 + i < pend;             // (then k - W >= 1 and k + W <= SLOTS - 2)
    o[i] = out ? (s_h[k - g.W] + h[1][i] + s_h[SLOTS + k + g.W]) + bias : 0.f;
  }
  if (VEC == 4 && PPT > 1) {                                    // p0, pend, s0 are multiples of PPT: a thread's group is wholly an output or not
    if (s0 >= p0 && s0 < pend) store_vec<T, PPT>(lg + s0, o);
  } else {
#pragma unroll
    for (int i = 0; i < PPT; ++i)
      if (s0 + i >= p0 && s0 + i < pend) lg[s0 + i] = from_f32<T>(o[i]);
  }
#ifdef MGACBAM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TRACE_MARK(A.trace, gid, 10);
#endif
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2))) void k_head_out(const Group<HeadArgs> G) {
  __shared__ __align__(16) float s_par[kHeadOutParCh * 12];
  __shared__ __align__(16) float s_ex[6 * kBlock];
  __shared__ __align__(16) float s_h[2 * 4 * kBlock];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const HeadArgs& A = G.lv[l];
  if (A.out_ppt == 4) head_out_run<T, VEC, 4>(A, local, s_par, s_ex, s_h);
  else if (A.out_ppt == 2) head_out_run<T, VEC, 2>(A, local, s_par, s_ex, s_h);
  else head_out_run<T, VEC, 1>(A, local, s_par, s_ex, s_h);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_head_bwd_act: g_s = convT3x3(g_logits, W_h);  g_a = g_s * SiLU'(a)  (stored);  per tile and hidden channel the partial sums
//   [0] sum g_a   [1] sum g_a*zhat   [2..10] dW_h taps: sum s(y,x) * g(y-u+1, x-v+1)   [11] sum g (channel 0 only: db_h)
//   workgroup = (run of 256*ppt consecutive pixels of one sample, group of kHeadJC hidden channels); thread = ppt pixels 256 apart
//   (ppt = 4 / 2 / 1 by image size: the 12 wave reductions per channel are paid once for ppt pixels).  g_logits of the run plus W+1
//   pixels either side in LDS; every thread keeps the 9 neighbours of each of its pixels in registers for all channels (taps that would
//   wrap around a row end are zeroed).
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void head_bwd_act_body(const HeadArgs& A, const int wgl, float* smem) {
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int PP = 4;                                         // upper bound of pixels per thread
  constexpr int ZU = kHeadJC;                                   // the workgroup's channels: their z values are requested together (x ppt pixels)
  const int ppt = A.act_ppt, TP = kBlock * ppt;
  const int ncg = (g.hid + kHeadJC - 1) / kHeadJC;
  const int wg = wgl / ncg, cgq = wgl - wg * ncg;
  const int jlo = cgq * kHeadJC, jhi = min(g.hid, jlo + kHeadJC);
  const int per = (g.HW + TP - 1) / TP;
  const int b = wg / per, p0 = (wg - b * per) * TP;
  const int halo = g.W + 1, HL = TP + 2 * halo;
  float* s_g = smem;                                            // [HL]
  float* s_red = smem + A.act_hl_max;                           // [16 lane rows][ZU][kHeadNStat], behind the longest run of the launch
  const int gid = 32768 + blockIdx.x;
  TRACE_HWID(A.trace, gid);
  TRACE_MARK(A.trace, gid, 0);
  const T* gl = static_cast<const T*>(A.gl) + static_cast<size_t>(b) * g.HW;
  const float* gl2 = A.gl2 ? A.gl2 + static_cast<size_t>(b) * g.HW : nullptr;
  // the first (normally only) channel batch's z values and constants are requested before anything else: they depend on nothing the
  // staging below produces, and every dependent trip to memory is ~1.3 us of a 7 us workgroup
  const float* zb = A.c.z + static_cast<size_t>(b) * g.hid * g.HW;
  float zq[ZU][PP];
  float pc[ZU][4 + 9];
  auto fetch = [&](const int j0) {
    const int jn = min(ZU, jhi - j0);
#pragma unroll
    for (int jj = 0; jj < ZU; ++jj)
#pragma unroll
      for (int i = 0; i < PP; ++i) {
        const int p = p0 + tid + i * kBlock;
        zq[jj][i] = 0.f;
        if (jj < jn && i < ppt && p < g.HW) zq[jj][i] = zb[static_cast<size_t>(j0 + jj) * g.HW + p];
      }
    // the channels' constants (BatchNorm scale / shift / mean / rstd, 3x3 weights) into scalar registers, once: read inside the pixel
    // loop they were vector loads re-issued after every g_a store (possible alias), 16 dependent round trips per thread
#pragma unroll
    for (int jj = 0; jj < ZU; ++jj) {
      const float* pr = A.c.par + static_cast<size_t>(min(j0 + jj, g.hid - 1)) * kHeadPar;
#pragma unroll
      for (int q = 0; q < 13; ++q) pc[jj][q] = uniform_load(pr + q);
    }
  };
  fetch(jlo);
  for (int i = tid; i < HL; i += kBlock) {
    const int p = p0 - halo + i;
    float v = 0.f;
    if (p >= 0 && p < g.HW) {
      v = to_f32<T>(gl[p]);
      if (gl2) v += gl2[p];                                     // the logits' second consumer (MaskCBAM's dL/dmask beside the loss's gradient)
    }
    s_g[i] = v;
  }
  __syncthreads();
  TRACE_MARK(A.trace, gid, 1);                                  // g_logits staged
  float g9[PP][9];                                              // g9[i][u*3+v] = g(y-u+1, x-v+1) of pixel i
  bool in[PP];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    const int p = p0 + tid + i * kBlock;
    in[i] = i < ppt && p < g.HW;
    const int x = p % g.W;
    const float m_in = in[i] ? 1.f : 0.f;                        // pixels past the run / the sample: every tap 0, so they add nothing to any sum
    const float m_lo = x > 0 ? m_in : 0.f, m_hi = x < g.W - 1 ? m_in : 0.f;   // g(.., x-1) / g(.., x+1) exist
    const float* c = s_g + (i < ppt ? tid + i * kBlock : tid) + halo;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const float* row = c + (1 - u) * g.W;                     // row y - u + 1
      g9[i][u * 3 + 0] = m_hi * row[1];                          // v = 0: x + 1
      g9[i][u * 3 + 1] = m_in * row[0];
      g9[i][u * 3 + 2] = m_lo * row[-1];                         // v = 2: x - 1
    }
  }
  float* part = A.s.part1 + static_cast<size_t>(wg) * g.hidp * kHeadNStat;
  float* gab = A.s.ga + static_cast<size_t>(b) * g.hid * g.HW;
  for (int j0 = jlo; j0 < jhi; j0 += ZU) {
    const int jn = min(ZU, jhi - j0);
    if (j0 != jlo) fetch(j0);
#ifdef MGACBAM_TRACE
    { float tsum = 0.f;
#pragma unroll
      for (int jj = 0; jj < ZU; ++jj)
#pragma unroll
        for (int i = 0; i < PP; ++i) tsum += zq[jj][i];
      if (tsum == 1.2345e30f) s_g[0] = tsum; }
    TRACE_MARK(A.trace, gid, 3);                                // z arrived
#endif
#pragma unroll
    for (int jj = 0; jj < ZU; ++jj) {
      if (jj < jn) {                                            // uniform
        const int j = j0 + jj;
        const float* pr = pc[jj];
        float r[kHeadNStat];
#pragma unroll
        for (int q = 0; q < kHeadNStat; ++q) r[q] = 0.f;
#pragma unroll
        for (int i = 0; i < PP; ++i) {
          if (i < ppt) {                                        // uniform; lanes past the sample run with g9 = 0 and z = 0 (g_a = 0, sums unchanged)
            const float z = zq[jj][i];
            const float zh = (z - pr[2]) * pr[3];
            const float a = z * pr[0] + pr[1];
            const float sg = sigmoid_fast(a);
            const float sv = a * sg;
            float gs = 0.f;
#pragma unroll
            for (int q = 0; q < 9; ++q) gs += pr[4 + q] * g9[i][q];
            const float ga = gs * (sg * (1.f + a * (1.f - sg)));
            if (in[i]) gab[static_cast<size_t>(j) * g.HW + p0 + tid + i * kBlock] = ga;
            r[0] += ga; r[1] += ga * zh;
#pragma unroll
            for (int q = 0; q < 9; ++q) r[2 + q] += sv * g9[i][q];
            if (j == 0) r[11] += g9[i][4];
          }
        }
        // sums over the 16 lanes of a DPP row only (4 one-instruction steps per value); the 16 rows of the workgroup meet in LDS below
#pragma unroll
        for (int q = 0; q < kHeadNStat; ++q) r[q] = wave_group_sum(r[q], 16);
        if ((lane & 15) == 0) {
#pragma unroll
          for (int q = 0; q < kHeadNStat; ++q) s_red[((wave * 4 + (lane >> 4)) * ZU + jj) * kHeadNStat + q] = r[q];
        }
      }
    }
    TRACE_MARK(A.trace, gid, 4);                                // wave 0: channels done (g_a stores issued, row sums)
    __syncthreads();
    TRACE_MARK(A.trace, gid, 2);                                // all waves there
    for (int i = tid; i < jn * kHeadNStat; i += kBlock) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) sum += s_red[w * ZU * kHeadNStat + i];       // fixed order: lane row 0..15
      part[static_cast<size_t>(j0) * kHeadNStat + i] = sum;
    }
    __syncthreads();
  }
#ifdef MGACBAM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TRACE_MARK(A.trace, gid, 10);
#endif
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
  float rstd0 = 0.f, gam0 = 0.f, mean0 = 0.f;                   // thread 0's tail operands, requested beside the partials
  if (tid == 0) { rstd0 = A.c.rstd[j]; gam0 = A.p.gamma[j]; mean0 = A.c.mean[j]; }
  double acc = 0.0;
  if (s < kHeadNStat) {
    const float* p = A.s.part1 + static_cast<size_t>(j) * kHeadNStat + s;
    const size_t stride = static_cast<size_t>(g.hidp) * kHeadNStat;
    constexpr int U = 8;
    int w = gq;
    for (; w + (U - 1) * 16 < A.nwg1; w += U * 16) {            // U independent loads in flight
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p[(w + u * 16) * stride];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += static_cast<double>(v[u]);
    }
    for (; w < A.nwg1; w += 16) acc += static_cast<double>(p[w * stride]);
  }
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
    A.s.kst[j] = gam0 * rstd0;
    A.s.kst[g.hidp + j] = mean0;
    A.s.kst[2 * g.hidp + j] = rstd0;
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
//   of both operands).  Workgroup = (channel block of kHeadCB channels, one of nshare pixel shares); its 4 waves take different
//   16*VEC... pixel chunks, accumulate all (hid tile, channel tile) products of the block and are summed through LDS at the end.
//   Accumulators: MTB x 4 tiles per wave (MTB = min(MT, 4): hid > 64 runs in passes of 64 hidden channels, re-reading x).
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__device__ __forceinline__ void head_bwd_gw_body(const HeadArgs& A, const int wg, float* smem) {
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const int cb = wg / A.nshare, share = wg - cb * A.nshare;
  const int c0 = cb * kHeadCB;
  const int MT = g.hidp >> 4;
  constexpr int NT = kHeadCB / 16;
  constexpr int CHP = 4 * VEC;                                  // pixels per K step (4 slots x VEC)
  const int nch = (g.HW + CHP - 1) / CHP;                       // chunks per sample
  const long long total = static_cast<long long>(g.B) * nch;
  float* s_kst = smem;                                          // [5][hidp]
  float* s_acc = smem + 5 * g.hidp;                             // [4 waves][16 x 16 tile] staging for the cross-wave sum
  const int gid = 40960 + blockIdx.x;
  TRACE_HWID(A.trace, gid);
  TRACE_MARK(A.trace, gid, 0);
  for (int i = tid; i < 5 * g.hidp; i += kBlock) s_kst[i] = A.s.kst[i];
  __syncthreads();
  TRACE_MARK(A.trace, gid, 1);                                  // constants staged
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
    // two pixel chunks per trip: both chunks' loads are issued before the first MFMA (a chunk per memory round trip would leave each
    // wave a chain of dependent latencies).  Measured and left alone (cfg2, 47-48 us, 152 MB fetched = 3.1 TB/s): more shares (1024), 3-4
    // chunks per trip for the levels with few hidden tiles, two hidden tiles per pass at 3 waves per SIMD, longest-chain level first,
    // register double buffering -- none moved it by more than +-2 us; H ADJACENT chunks per wave (instead of across the 4 waves) cost
    // +10 us.  The 64-byte-per-row-and-instruction footprint of the MFMA operand layout (4 lanes share a channel row) is what is left.
    const long long stride = 4ll * A.nshare;
    for (long long ch0 = static_cast<long long>(share) * 4 + wave; ch0 < total; ch0 += 2 * stride) {
      float av[2][kHeadMTW][VEC], bv[2][NT][VEC];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const long long ch = ch0 + h * stride;
        const bool live = ch < total;
        const int b = live ? static_cast<int>(ch / nch) : 0;
        const int px = live ? (static_cast<int>(ch - static_cast<long long>(b) * nch)) * CHP + lq * VEC : g.HW;
        const bool ok = px < g.HW;
        const size_t pxo = ok ? px : 0;
#pragma unroll
        for (int t = 0; t < kHeadMTW; ++t) {
          const int j = (mt0 + t) * 16 + lr;
#pragma unroll
          for (int r = 0; r < VEC; ++r) av[h][t][r] = 0.f;
          if (t < mtn && ok && j < g.hid) {
            float ga[VEC], zv[VEC];
            const size_t o = (static_cast<size_t>(b) * g.hid + j) * g.HW + pxo;
            load_vec<float, VEC>(A.s.ga + o, ga);
            load_vec<float, VEC>(A.c.z + o, zv);
#pragma unroll
            for (int r = 0; r < VEC; ++r) av[h][t][r] = kj[t] * (ga[r] - gbn[t] - (zv[r] - mean[t]) * rstd[t] * ggn[t]);
          }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int c = c0 + n * 16 + lr;
#pragma unroll
          for (int r = 0; r < VEC; ++r) bv[h][n][r] = 0.f;
          if (ok && c < g.C) load_vec<T, VEC>(static_cast<const T*>(A.x) + (static_cast<size_t>(b) * g.C + c) * g.HW + pxo, bv[h][n]);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int t = 0; t < kHeadMTW; ++t) {
          if (t < mtn) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              if constexpr (HalfMma<T>::on && VEC == 4) {        // a lane's 4 pixels = one K group of the half-precision MFMA
                acc[t][n] = HalfMma<T>::mma(av[h][t][0], av[h][t][1], av[h][t][2], av[h][t][3], bv[h][n][0], bv[h][n][1], bv[h][n][2], bv[h][n][3], acc[t][n]);
              } else {
#pragma unroll
                for (int r = 0; r < VEC; ++r) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][t][r], bv[h][n][r], acc[t][n], 0, 0, 0);
              }
            }
          }
        }
      }
    }
#ifdef MGACBAM_TRACE
    if (acc[0][0][0] == 1.2345e30f) s_acc[0] = 0.f;
    TRACE_MARK(A.trace, gid, 2);                                // wave 0: pixel loop done
#endif
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
#ifdef MGACBAM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TRACE_MARK(A.trace, gid, 10);
#endif
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_head_bwd_gw(const Group<HeadArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  head_bwd_gw_body<T, VEC>(G.lv[l], local, smem);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_head_bwd_gw2: the same product with both operands staged through LDS (H*W % 4 == 0, hid <= 64; k_head_bwd_gw otherwise).
//   Why: in the MFMA operand layout only 4 lanes share a channel row, so k_head_bwd_gw's global loads touch 64 B per row and
//   instruction -- 3.1 TB/s whatever the occupancy, share count or prefetch depth (notes there).  Here the global loads are
//   row-contiguous (a wave-load = 4 rows x 256 B, like the forward GEMM's 5+ TB/s), LDS does the transposition into operands:
//   workgroup = (channel block of 64 channels, pixel share); per chunk of 64 consecutive pixels of one sample
//     registers -> LDS:  x tile [64 channels][64 px], g_z tile [hidp][64 px] (g_z formed from g_a, z while storing); row pitch 68 floats:
//                        16-byte aligned, and the 16 lanes of a quarter wave (16 rows, same 16-byte column) hit 16 x 4 distinct banks
//     next chunk's global loads are issued, then the MFMAs of this chunk run from LDS (ds_read_b128: 4 K steps per read);
//   wave w owns channel tile w and every hidden tile (MT x 16 accumulator registers): no cross-wave sum at the end.
//   Two barriers per chunk; the loads of chunk i+1 are in flight during the MFMAs of chunk i.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kHeadGwPx = 64;                                   // pixels per chunk
constexpr int kHeadGwPitch = kHeadGwPx + 4;                     // LDS row pitch (floats)
template <typename T>
__device__ __forceinline__ void head_bwd_gw2_body(const HeadArgs& A, const int wg, float* smem) {
  const HeadGeo& g = A.g;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const int cb = wg / A.nshare, share = wg - cb * A.nshare;
  const int c0 = cb * kHeadCB;
  const int MT = g.hidp >> 4;                                   // <= 4 (host)
  const int nch = (g.HW + kHeadGwPx - 1) / kHeadGwPx;           // chunks per sample
  const int total = g.B * nch;
  float* s_kst = smem;                                          // [5][hidp]
  float* s_x = smem + 5 * g.hidp;                               // [64][pitch]
  float* s_g = s_x + kHeadCB * kHeadGwPitch;                    // [hidp][pitch]
  const int gid = 40960 + blockIdx.x;
  TRACE_HWID(A.trace, gid);
  TRACE_MARK(A.trace, gid, 0);
  for (int i = tid; i < 5 * g.hidp; i += kBlock) s_kst[i] = A.s.kst[i];
  // staging roles: a wave-load covers 4 rows x 256 B (lane: row lq of the group, 16-byte column lr)
  //   x: wave w stages channel rows 16w .. 16w+15 (4 loads); g_z: hidden rows in groups of 4, dealt round-robin over the waves
  const int ngz = g.hidp >> 2;                                  // row groups of g_z (4 per hidden tile)
  constexpr int GZ = 4;                                         // row groups per wave at most (hidp <= 64)
  float xv[4][4], gav[GZ][4], zvv[GZ][4];
  auto issue = [&](const int ch) {
    const int b = ch / nch, px0 = (ch - b * nch) * kHeadGwPx + 4 * lr;
    const bool pok = ch < total && px0 < g.HW;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + wave * 16 + q * 4 + lq;
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[q][r] = 0.f;
      if (pok && c < g.C) load_vec<T, 4>(static_cast<const T*>(A.x) + (static_cast<size_t>(b) * g.C + c) * g.HW + px0, xv[q]);
    }
#pragma unroll
    for (int q = 0; q < GZ; ++q) {
      const int j = (wave + 4 * q) * 4 + lq;                    // row group wave + 4q
#pragma unroll
      for (int r = 0; r < 4; ++r) { gav[q][r] = 0.f; zvv[q][r] = 0.f; }
      if (wave + 4 * q < ngz && pok && j < g.hid) {
        const size_t o = (static_cast<size_t>(b) * g.hid + j) * g.HW + px0;
        load_vec<float, 4>(A.s.ga + o, gav[q]);
        load_vec<float, 4>(A.c.z + o, zvv[q]);
      }
    }
  };
  auto to_lds = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) store_vec<float, 4>(s_x + (wave * 16 + q * 4 + lq) * kHeadGwPitch + 4 * lr, xv[q]);
#pragma unroll
    for (int q = 0; q < GZ; ++q) {
      if (wave + 4 * q < ngz) {                                 // uniform
        const int j = (wave + 4 * q) * 4 + lq;
        const float kj = s_kst[j], mean = s_kst[g.hidp + j], rstd = s_kst[2 * g.hidp + j], gbn = s_kst[3 * g.hidp + j], ggn = s_kst[4 * g.hidp + j];
        float gz[4];                                            // (pixels that were not loaded meet x = 0; padding rows have zero constants)
#pragma unroll
        for (int r = 0; r < 4; ++r) gz[r] = kj * (gav[q][r] - gbn - (zvv[q][r] - mean) * rstd * ggn);
        store_vec<float, 4>(s_g + j * kHeadGwPitch + 4 * lr, gz);
      }
    }
  };
  v4f32 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = v4f32{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                              // s_kst
  TRACE_MARK(A.trace, gid, 1);                                  // constants staged
  int ch = share;
  issue(ch);
  for (; ch < total; ch += A.nshare) {                          // (uniform over the workgroup)
    __syncthreads();                                            // everyone is done reading the previous chunk
    to_lds();
    __syncthreads();
    issue(ch + A.nshare);                                       // in flight during the MFMAs below (a chunk past the end loads nothing)
#pragma unroll
    for (int sp = 0; sp < kHeadGwPx / 16; ++sp) {               // 16 pixels = 4 K steps per LDS read
      float bq[4];
      load_vec<float, 4>(s_x + (wave * 16 + lr) * kHeadGwPitch + sp * 16 + 4 * lq, bq);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < MT) {                                           // uniform
          float aq[4];
          load_vec<float, 4>(s_g + (t * 16 + lr) * kHeadGwPitch + sp * 16 + 4 * lq, aq);
          if constexpr (HalfMma<T>::on) {
            acc[t] = HalfMma<T>::mma(aq[0], aq[1], aq[2], aq[3], bq[0], bq[1], bq[2], bq[3], acc[t]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[r], bq[r], acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
#ifdef MGACBAM_TRACE
  if (acc[0][0] == 1.2345e30f) s_x[0] = 0.f;
  TRACE_MARK(A.trace, gid, 2);                                  // wave 0: pixel loop done
#endif
  // D layout: lane l, register v: row (hidden) 4*(l/16)+v, column (channel) l%16; wave w owns channel tile w
  float* outp = A.s.gwpart + static_cast<size_t>(wg) * g.hidp * kHeadCB;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < MT) {
#pragma unroll
      for (int v = 0; v < 4; ++v) outp[static_cast<size_t>(t * 16 + lq * 4 + v) * kHeadCB + wave * 16 + lr] = acc[t][v];
    }
  }
#ifdef MGACBAM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TRACE_MARK(A.trace, gid, 10);
#endif
}

template <typename T>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3))) void k_head_bwd_gw2(const Group<HeadArgs> G) {
  extern __shared__ __align__(16) float smem[];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  head_bwd_gw2_body<T>(G.lv[l], local, smem);
}

// k_head_bwd_gwf: dW1[j,c] = sum over the pixel shares: workgroup = 16 outputs x 16 share strides, up to 16 loads in flight per thread
//   (512 shares = two round trips), then a fixed-order combine through LDS
constexpr int kHeadGwfOut = 16;
__global__ __launch_bounds__(kBlock) void k_head_bwd_gwf(const Group<HeadArgs> G) {
  __shared__ float red[kBlock];
  int local;
  const int l = find_level(G, blockIdx.x, local);
  const HeadArgs& A = G.lv[l];
  const HeadGeo& g = A.g;
  constexpr int NS = kBlock / kHeadGwfOut;                      // share strides
  const int tid = threadIdx.x, o = tid % kHeadGwfOut, q = tid / kHeadGwfOut;
  const int idx = local * kHeadGwfOut + o;
  float s = 0.f;
  if (idx < g.hid * g.C) {
    const int j = idx / g.C, c = idx - j * g.C;
    const int cb = c / kHeadCB, cc = c - cb * kHeadCB;
    const float* p = A.s.gwpart + (static_cast<size_t>(cb) * A.nshare * g.hidp + j) * kHeadCB + cc;
    const size_t stride = static_cast<size_t>(g.hidp) * kHeadCB;
    constexpr int U = 16;
    for (int w0 = q; w0 < A.nshare; w0 += U * NS) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int w = w0 + u * NS; v[u] = w < A.nshare ? p[w * stride] : 0.f; }
#pragma unroll
      for (int u = 0; u < U; ++u) s += v[u];
    }
  }
  red[tid] = s;
  __syncthreads();
  if (q == 0 && idx < g.hid * g.C) {
    float t = red[o];
#pragma unroll
    for (int w = 1; w < NS; ++w) t += red[w * kHeadGwfOut + o];
    A.gw1[idx] = t;
  }
}

}  // namespace mgacbam
