// LDS staging of image-plane windows and the 2-D conv tiling shared by the forward (k_apply, k_gate) and backward
// (k_bwd_convT, k_bwd_wsa) kernels.
#pragma once
#include "args.cuh"
#include "common.cuh"

namespace mgacbam {

// ---------------------------------------------------------------------------------------------
// LDS staging of image planes with zero padding.  Loads are issued U at a time per thread before any LDS store,
// so a tile costs one or two global-load latencies, not one per element.
//   generic form: `total` elements, element idx -> (plane p, row yy, col xx) of a PH x PW window whose top-left
//   image coordinate is (ya, xa); load(p, off) -> element `off` of plane p of this sample (H*W floats)
// ---------------------------------------------------------------------------------------------
template <int U, typename LoadFn>
__device__ __forceinline__ void stage_window(float* tile, int NP, int PH, int PW, int ya, int xa, const Geo& g, LoadFn load) {
  const int total = NP * PH * PW;
  // idx / d == umulhi(idx, 2^32/d + 1) for idx < 2^16; d == 1 would overflow the magic, so it gets the identity
  const unsigned mpw = PW > 1 ? 0xFFFFFFFFu / static_cast<unsigned>(PW) + 1u : 0u;
  const unsigned mph = PH > 1 ? 0xFFFFFFFFu / static_cast<unsigned>(PH) + 1u : 0u;
  for (int base = 0; base < total; base += kBlock * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * kBlock + threadIdx.x;
      const unsigned r = PW > 1 ? __umulhi(static_cast<unsigned>(idx), mpw) : static_cast<unsigned>(idx);   // row over all planes
      const int xx = idx - static_cast<int>(r) * PW;
      const unsigned p = PH > 1 ? __umulhi(r, mph) : r;
      const int yy = static_cast<int>(r) - static_cast<int>(p) * PH;
      const int gy_ = ya + yy, gx_ = xa + xx;
      v[u] = 0.f;
      if (idx < total && gy_ >= 0 && gy_ < g.H && gx_ >= 0 && gx_ < g.W) v[u] = load(static_cast<int>(p), gy_ * g.W + gx_);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * kBlock + threadIdx.x;
      if (idx < total) tile[idx] = v[u];
    }
  }
}

// 2-D conv tiles used by the backward conv kernel: TH rows x TW = 4*TWQ columns of one sample
struct ConvTile {
  int b, y0, x0, TW, TH, PW, PH, pad, k;
};
__device__ __forceinline__ ConvTile conv_tile(const Geo& g, const Tune& t, int k, int bid, int th) {
  ConvTile c;
  c.k = k; c.pad = k / 2;
  c.TW = t.conv_twq * 4; c.TH = th;
  c.PW = c.TW + k - 1; c.PH = c.TH + k - 1;
  const int tiles_x = (g.W + c.TW - 1) / c.TW, tiles_y = (g.H + c.TH - 1) / c.TH;
  const int txi = bid % tiles_x; bid /= tiles_x;
  const int tyi = bid % tiles_y;
  c.b = bid / tiles_y;
  c.y0 = tyi * c.TH; c.x0 = txi * c.TW;
  return c;
}

}  // namespace mgacbam
