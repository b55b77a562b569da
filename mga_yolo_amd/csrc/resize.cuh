// Nearest-neighbour resize: the integer index path of the reference's target resize (bit-exact by construction).
#pragma once
#include "common.cuh"

namespace mgacbam {

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
