// Ablation of the k_pool structure on the P3 shape (B=32, C=64, HW=6400, fp32): which ingredient costs what.
//   hipcc -O3 --offload-arch=gfx950 tools/poolbench.hip -o /tmp/poolbench && /tmp/poolbench
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CPT, int LEVEL>   // LEVEL 0: x only; 1: + mask load; 2: + sigmoid & sxs; 3: + argmax; 4: + block reduction epilogue
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out,
                                         int B, int C, int HW, int TX, int S = 1) {
  __shared__ float red[64];
  const int tid = threadIdx.x;
  const int lt = 31 - __clz(TX);
  const int tx = tid & (TX - 1), ty = tid >> lt, TY = 256 >> lt;
  const int CPB = TY * CPT, ncg = C / CPB;
  const int part = blockIdx.x % S, bq = blockIdx.x / S;       // S-way split of the H*W sweep over workgroups (partials combined later)
  const int b = bq / ncg, cg = bq % ncg;
  const int c0 = cg * CPB + ty * CPT;
  const int nvt = HW / 4, nvp = (nvt + S - 1) / S;
  const int v0 = part * nvp, nv = min(nvt, v0 + nvp);
  const float4* xr[CPT];
  for (int j = 0; j < CPT; ++j) xr[j] = reinterpret_cast<const float4*>(x + ((size_t)b * C + c0 + j) * HW);
  const float4* mb = reinterpret_cast<const float4*>(mask + (size_t)b * HW);
  float sx[CPT], sxs[CPT], vmax[CPT]; int imax[CPT];
  for (int j = 0; j < CPT; ++j) { sx[j] = 0; sxs[j] = 0; vmax[j] = -FLT_MAX; imax[j] = 0; }
  float ssum = 0;
  for (int i = v0 + tx; i < nv; i += TX) {
    float s[4] = {1, 1, 1, 1}; bool sel[4] = {true, true, true, true};
    if (LEVEL >= 1) {
      float4 m = mb[i]; float mm[4] = {m.x, m.y, m.z, m.w};
      for (int e = 0; e < 4; ++e) {
        s[e] = LEVEL >= 2 ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * mm[e])) : mm[e];
        sel[e] = s[e] > 0.5f; ssum += s[e];
      }
    }
    float4 xv[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) xv[j] = xr[j][i];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      float v[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sx[j] += v[e];
        if (LEVEL >= 2) sxs[j] += v[e] * s[e];
        if (LEVEL >= 3) { if (sel[e] && v[e] > vmax[j]) { vmax[j] = v[e]; imax[j] = i * 4 + e; } }
      }
    }
  }
  if (LEVEL >= 4) {
    for (int j = 0; j < CPT; ++j) {
      for (int o = (TX < 64 ? TX : 64) >> 1; o > 0; o >>= 1) {
        sx[j] += __shfl_xor(sx[j], o, 64); sxs[j] += __shfl_xor(sxs[j], o, 64);
        float ov = __shfl_xor(vmax[j], o, 64); int oi = __shfl_xor(imax[j], o, 64);
        if (ov > vmax[j] || (ov == vmax[j] && oi < imax[j])) { vmax[j] = ov; imax[j] = oi; }
      }
    }
    if (TX > 64) {
      __syncthreads();
      if ((tid & 63) == 0) for (int j = 0; j < CPT; ++j) { red[(tid >> 6) * CPT + j] = sx[j]; red[32 + (tid >> 6) * CPT + j] = sxs[j]; }
      __syncthreads();
      if (tx == 0) for (int j = 0; j < CPT; ++j) for (int w = 1; w < TX / 64; ++w) { sx[j] += red[((tid >> 6) + w) * CPT + j]; sxs[j] += red[32 + ((tid >> 6) + w) * CPT + j]; }
    }
    if (tx == 0) for (int j = 0; j < CPT; ++j) { out[(((size_t)b * C + c0 + j) * S + part) * 4] = sx[j] + ssum; out[(((size_t)b * C + c0 + j) * S + part) * 4 + 1] = sxs[j]; out[(((size_t)b * C + c0 + j) * S + part) * 4 + 2] = vmax[j]; out[(((size_t)b * C + c0 + j) * S + part) * 4 + 3] = (float)imax[j]; }
  } else {
    float t = ssum;
    for (int j = 0; j < CPT; ++j) t += sx[j] + sxs[j] + vmax[j] + imax[j];
    if (t == 1.2345f) out[0] = t;
  }
}

int main(int argc, char** argv) {
  const int B = argc > 3 ? atoi(argv[1]) : 32, C = argc > 3 ? atoi(argv[2]) : 64, HW = argc > 3 ? atoi(argv[3]) : 6400;
  printf("B=%d C=%d HW=%d (%.1f MB)\n", B, C, HW, B * (double)C * HW * 4 / 1e6);
  const size_t n = (size_t)B * C * HW;
  float *x, *m, *o;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&m, (size_t)B * HW * 4)); CK(hipMalloc(&o, (size_t)B * C * 16 * 8));
  CK(hipMemset(x, 0, n * 4)); CK(hipMemset(m, 0, (size_t)B * HW * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 30; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %7.2f us  %7.1f GB/s\n", name, ms * 1e3 / 30, n * 4.0 * 30 / (ms * 1e-3) / 1e9);
  };
#define RUN(CPT, LEVEL, TX) run("cpt" #CPT " level" #LEVEL " tx" #TX, [&] { hipLaunchKernelGGL((k<CPT, LEVEL>), dim3(B * C / ((256 / TX) * CPT)), dim3(256), 0, 0, x, m, o, B, C, HW, TX); })
  RUN(2, 0, 256); RUN(2, 1, 256); RUN(2, 2, 256); RUN(2, 3, 256); RUN(2, 4, 256);
  RUN(1, 0, 256); RUN(1, 4, 256); RUN(4, 0, 256); RUN(4, 1, 256); RUN(4, 2, 256); RUN(4, 3, 256); RUN(4, 4, 256);
  RUN(2, 0, 64); RUN(2, 4, 64); RUN(1, 0, 64); RUN(1, 4, 64); RUN(4, 4, 64);
  RUN(2, 0, 128); RUN(2, 4, 128);
#define RUNS(CPT, LEVEL, TX, S) run("cpt" #CPT " level" #LEVEL " tx" #TX " split" #S, [&] { hipLaunchKernelGGL((k<CPT, LEVEL>), dim3(S * B * C / ((256 / TX) * CPT)), dim3(256), 0, 0, x, m, o, B, C, HW, TX, S); })
  RUNS(2, 0, 256, 2); RUNS(2, 4, 256, 2); RUNS(2, 0, 256, 4); RUNS(2, 4, 256, 4); RUNS(2, 4, 256, 8); RUNS(1, 4, 256, 2); RUNS(1, 4, 256, 4); RUNS(4, 4, 256, 4); RUNS(4, 4, 256, 8);
  RUNS(2, 4, 64, 2); RUNS(2, 4, 64, 4); RUNS(2, 4, 128, 4);
  return 0;
}
