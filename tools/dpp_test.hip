// Validates the DPP-based wave reductions of csrc/common.cuh (no LDS crossbar traffic, unlike __shfl_xor = ds_bpermute_b32) against
// the shuffle forms, for group widths 2..64, on random data.   hipcc -O3 --offload-arch=gfx950 tools/dpp_test.hip -o /tmp/dpp_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "../mga_yolo_amd/csrc/common.cuh"
using namespace mgacbam;

__global__ void k(const float* in, float* out_shfl, float* out_dpp, int width) {
  const float v = in[blockIdx.x * 64 + threadIdx.x];
  float a = v;
  for (int o = width >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  out_shfl[blockIdx.x * 64 + threadIdx.x] = a;
  out_dpp[blockIdx.x * 64 + threadIdx.x] = wave_group_sum(v, width);
}
__global__ void ka(const float* in, float* vs, int* is, float* vd, int* id, int width) {   // arg-max: (value, first index) pairs
  const int t = blockIdx.x * 64 + threadIdx.x;
  float v = roundf(in[t] * 8.f);                                // coarse values: many ties, the first index must win
  int i = threadIdx.x;
  float a = v; int ai = i;
  for (int o = width >> 1; o > 0; o >>= 1) { float ov = __shfl_xor(a, o, 64); int oi = __shfl_xor(ai, o, 64); argmax_combine(a, ai, ov, oi); }
  vs[t] = a; is[t] = ai;
  wave_group_argmax(v, i, width);
  vd[t] = v; id[t] = i;
}
__global__ void kt(const float* in, float* out, int n, int reps, int mode) {   // throughput: 12 values per lane reduced per iteration
  float v[12];
  for (int q = 0; q < 12; ++q) v[q] = in[(blockIdx.x * 64 + threadIdx.x + q) % n];
  float acc = 0.f;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      float a = v[q] + r;
      if (mode == 0) { for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64); }
      else a = wave_group_sum(a, 64);
      acc += a;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  const int N = 64 * 64;
  float *in, *a, *b;
  hipMalloc(&in, N * 4); hipMalloc(&a, 2048 * 256 * 4); hipMalloc(&b, N * 4);   // a also receives kt's 2048 x 256 outputs
  float* h = (float*)malloc(N * 4);
  for (int i = 0; i < N; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, N * 4, hipMemcpyHostToDevice);
  float* ha = (float*)malloc(N * 4); float* hb = (float*)malloc(N * 4);
  int bad = 0;
  for (int w = 1; w <= 64; w *= 2) {
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, in, a, b, w);
    hipMemcpy(ha, a, N * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, N * 4, hipMemcpyDeviceToHost);
    double e = 0;
    for (int i = 0; i < N; ++i) e = fmax(e, fabs(ha[i] - hb[i]));
    printf("width %2d max |shfl - dpp| = %.3g %s\n", w, e, e < 1e-5 ? "ok" : "MISMATCH");
    bad += e >= 1e-5;
  }
  int *ia, *ib; hipMalloc(&ia, N * 4); hipMalloc(&ib, N * 4);
  int* hia = (int*)malloc(N * 4); int* hib = (int*)malloc(N * 4);
  for (int w = 1; w <= 64; w *= 2) {
    hipLaunchKernelGGL(ka, dim3(64), dim3(64), 0, 0, in, a, ia, b, ib, w);
    hipMemcpy(ha, a, N * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hia, ia, N * 4, hipMemcpyDeviceToHost); hipMemcpy(hib, ib, N * 4, hipMemcpyDeviceToHost);
    int mism = 0;
    for (int i = 0; i < N; ++i) mism += (ha[i] != hb[i]) || (hia[i] != hib[i]);
    printf("argmax width %2d mismatches = %d %s\n", w, mism, mism ? "MISMATCH" : "ok");
    bad += mism != 0;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(kt, dim3(2048), dim3(256), 0, 0, in, a, N, 10, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kt, dim3(2048), dim3(256), 0, 0, in, a, N, 200, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.1f us for 2048 WGs x 4 waves x 200 x 12 wave-sums\n", mode ? "dpp " : "shfl", ms * 1e3);
  }
  printf(bad ? "FAILED\n" : "PASS\n");
  return bad;
}
