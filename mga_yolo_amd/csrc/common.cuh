// Device helpers shared by the forward and backward kernels (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <float.h>
#include <stdint.h>

#include "args.cuh"

#ifndef MGACBAM_POOL_PF
#define MGACBAM_POOL_PF 1   // H*W positions per lane per memory round in the sweep kernels (k_pool, k_bwd_reduce2)
#endif

namespace mgacbam {

constexpr int kBlock = 256;        // every kernel uses 256-thread workgroups = 4 waves
constexpr int kWave = 64;
using bf16_t = __hip_bfloat16;

// ---------------------------------------------------------------------------------------------
// element I/O: T in {float, __half, __hip_bfloat16}; arithmetic is always fp32
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return __bfloat162float(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return __float2bfloat16(v); }

template <typename T, int VEC> struct alignas(sizeof(T) * VEC < 16 ? sizeof(T) * VEC : 16) Pack { T v[VEC]; };

// one VEC-wide (4 x fp32 = 16 B, 4 x half = 8 B) coalesced load / store per lane
template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, float (&out)[VEC]) {
  Pack<T, VEC> r = *reinterpret_cast<const Pack<T, VEC>*>(p);
#pragma unroll
  for (int e = 0; e < VEC; ++e) out[e] = to_f32<T>(r.v[e]);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* __restrict__ p, const float (&in)[VEC]) {
  Pack<T, VEC> r;
#pragma unroll
  for (int e = 0; e < VEC; ++e) r.v[e] = from_f32<T>(in[e]);
  *reinterpret_cast<Pack<T, VEC>*>(p) = r;
}
// write-once streams (y, gx): non-temporal, so they do not displace x / gy (re-read by later kernels) from L2 / Infinity Cache
template <typename T, int VEC>
__device__ __forceinline__ void store_vec_stream(T* __restrict__ p, const float (&in)[VEC], bool nt) {
  Pack<T, VEC> r;
#pragma unroll
  for (int e = 0; e < VEC; ++e) r.v[e] = from_f32<T>(in[e]);
  if (nt) {
    if constexpr (sizeof(T) * VEC == 16) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(*reinterpret_cast<v4f*>(&r), reinterpret_cast<v4f*>(p));
    } else if constexpr (sizeof(T) * VEC == 8) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      __builtin_nontemporal_store(*reinterpret_cast<v2f*>(&r), reinterpret_cast<v2f*>(p));
    } else if constexpr (sizeof(T) * VEC == 4) {
      __builtin_nontemporal_store(*reinterpret_cast<float*>(&r), reinterpret_cast<float*>(p));
    } else {
      __builtin_nontemporal_store(*reinterpret_cast<unsigned short*>(&r), reinterpret_cast<unsigned short*>(p));
    }
  } else {
    *reinterpret_cast<Pack<T, VEC>*>(p) = r;
  }
}
template <int VEC>
__device__ __forceinline__ void load_ivec(const int* __restrict__ p, int (&out)[VEC]) {
  Pack<int, VEC> r = *reinterpret_cast<const Pack<int, VEC>*>(p);
#pragma unroll
  for (int e = 0; e < VEC; ++e) out[e] = r.v[e];
}
template <int VEC>
__device__ __forceinline__ void store_ivec(int* __restrict__ p, const int (&in)[VEC]) {
  Pack<int, VEC> r;
#pragma unroll
  for (int e = 0; e < VEC; ++e) r.v[e] = in[e];
  *reinterpret_cast<Pack<int, VEC>*>(p) = r;
}

// ---------------------------------------------------------------------------------------------
// scalar math
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
// per-pixel sigmoid on the streaming path: v_exp_f32 + v_rcp_f32 (relative error ~1e-6, two transcendental issues);
// the accurate expf costs ~10x the VALU slots and made k_pool VALU-bound instead of HBM-bound
__device__ __forceinline__ float sigmoid_fast(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
// torch.nn.functional.softplus(beta=1, threshold=20)  (masked_cbam.py:150-152)
__device__ __forceinline__ float softplusf_(float v) { return v > 20.0f ? v : log1pf(expf(v)); }
// torch.isclose(a, b) with the default rtol=1e-5, atol=1e-8 (masked_cbam.py:120)
__device__ __forceinline__ bool isclosef_(float a, float b) { return fabsf(a - b) <= 1e-8f + 1e-5f * fabsf(b); }

// ---------------------------------------------------------------------------------------------
// reductions.  Threads of a workgroup are laid out as TY rows of TX lanes, TX a power of two in
// [1,256], row = tid / TX.  A "row reduction" combines the TX lanes of one row: xor-shuffles inside a
// wave (groups are TX-aligned so xor offsets < TX never leave the group) and, when a row spans
// several waves (TX = 128, 256), a second step through LDS.
// ---------------------------------------------------------------------------------------------
// Sums inside aligned groups of `width` lanes (power of two, 1..64); every lane of the group gets the group's sum.
// DPP forms, not __shfl_xor: a shuffle compiles to ds_bpermute_b32, i.e. one trip through the LDS crossbar per step and value -- in the
// kernels that reduce per channel (k_bwd_reduce1: 2 sums x log2(TX) steps per channel and wave; the mask head's backward: 12 sums per
// channel) that traffic, shared by every wave of the CU, costs more than the arithmetic.  quad_perm / row_mirror / row_bcast run in the
// VALU.  After the two quad steps every quad holds its sum in all four lanes, so mirroring within 8 and within 16 lanes pairs each
// half with the other one (same effect as xor 4 / xor 8); across the four rows of 16: row_bcast15 / row_bcast31 (gfx9 DPP) leave
// S0+S1 in lane 31 and the wave's total in lane 63, which v_readlane hands to every lane.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
  // lanes whose row is masked out (or whose source is invalid) receive `old` = 0.  Full row mask (the in-row steps: every lane has a
  // valid source): bound_ctrl set, which is what lets the compiler fold the move into ONE v_add_f32_dpp -- with bound_ctrl clear it
  // emits v_mov 0 + v_mov_dpp + v_add per step (GCNDPPCombine needs an undefined `old` for float adds)
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF));
}
// a value every lane reads from the same address, kept in a scalar register from here on (the compiler does not use scalar loads for
// memory the kernel may also write, and re-issues vector loads wherever a store could alias: request such constants once, up front)
__device__ __forceinline__ float uniform_load(const float* p) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *p)));
}
// a wave-uniform pointer, pinned in scalar registers: address = SGPR base + a lane's 32-bit byte offset is then ONE VGPR per load (left
// alone the compiler folds the lane's part into the base first and keeps a 64-bit VGPR address per row: 2 registers per load in flight)
template <typename P>
__device__ __forceinline__ const P* uniform_ptr(const P* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u >> 32));
  return reinterpret_cast<const P*>((static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ float wave_group_sum(float v, int width) {
  if (width >= 2) v = dpp_add<0xB1>(v);                         // quad_perm [1,0,3,2]
  if (width >= 4) v = dpp_add<0x4E>(v);                         // quad_perm [2,3,0,1]
  if (width >= 8) v = dpp_add<0x141>(v);                        // row_half_mirror
  if (width >= 16) v = dpp_add<0x140>(v);                       // row_mirror: every lane of a row of 16 holds the row's sum
  if (width >= 32) {
    v = dpp_add<0x142, 0xA>(v);                                 // row_bcast15 into rows 1 and 3: lanes 16..31 = S0+S1, 48..63 = S2+S3
    if (width >= 64) {
      v = dpp_add<0x143, 0xC>(v);                               // row_bcast31 into rows 2 and 3: lane 63 = S0+S1+S2+S3
      v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    } else {
      const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
      const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
      v = (threadIdx.x & 32) ? hi : lo;
    }
  }
  return v;
}
// (value, index) arg-max with first-index tie break: larger value wins, equal values -> smaller index
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
// the same DPP steps for (value, index) pairs; lanes a step does not reach combine with themselves (old = own value: a no-op)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ void dpp_argmax(float& v, int& i) {
  const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
  const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xF, false);
  argmax_combine(v, i, ov, oi);
}
__device__ __forceinline__ void wave_group_argmax(float& v, int& i, int width) {
  if (width >= 2) dpp_argmax<0xB1>(v, i);
  if (width >= 4) dpp_argmax<0x4E>(v, i);
  if (width >= 8) dpp_argmax<0x141>(v, i);
  if (width >= 16) dpp_argmax<0x140>(v, i);
  if (width >= 32) {
    dpp_argmax<0x142, 0xA>(v, i);
    if (width >= 64) {
      dpp_argmax<0x143, 0xC>(v, i);
      v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
      i = __builtin_amdgcn_readlane(i, 63);
    } else {
      const float vlo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
      const float vhi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
      const int ilo = __builtin_amdgcn_readlane(i, 31), ihi = __builtin_amdgcn_readlane(i, 63);
      const bool up = (threadIdx.x & 32) != 0;
      v = up ? vhi : vlo; i = up ? ihi : ilo;
    }
  }
}

// Row-sum of n values per thread.  `red` = LDS scratch of at least n * (kBlock / kWave) floats, used only
// when TX > 64.  The result is valid in the first lane of each row (tx == 0).  All threads must call.
template <int N>
__device__ __forceinline__ void row_sum(float (&v)[N], int TX, int tid, float* red) {
#pragma unroll
  for (int n = 0; n < N; ++n) v[n] = wave_group_sum(v[n], TX);
  if (TX > kWave) {                       // uniform per launch
    const int wave = tid >> 6, lane = tid & 63;
    __syncthreads();                       // red may still be in use by a previous call
    if (lane == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) red[wave * N + n] = v[n];
    }
    __syncthreads();
    const int waves_per_row = TX >> 6;
    if ((tid & (TX - 1)) == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        float a = red[wave * N + n];
        for (int w = 1; w < waves_per_row; ++w) a += red[(wave + w) * N + n];
        v[n] = a;
      }
    }
  }
}
template <int N>
__device__ __forceinline__ void row_argmax(float (&v)[N], int (&idx)[N], int TX, int tid, float* red) {
#pragma unroll
  for (int n = 0; n < N; ++n) wave_group_argmax(v[n], idx[n], TX);
  if (TX > kWave) {
    const int wave = tid >> 6, lane = tid & 63;
    int* redi = reinterpret_cast<int*>(red) + N * (kBlock / kWave);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) { red[wave * N + n] = v[n]; redi[wave * N + n] = idx[n]; }
    }
    __syncthreads();
    const int waves_per_row = TX >> 6;
    if ((tid & (TX - 1)) == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        for (int w = 1; w < waves_per_row; ++w) argmax_combine(v[n], idx[n], red[(wave + w) * N + n], redi[(wave + w) * N + n]);
      }
    }
  }
}

// whole-workgroup sum (result valid in thread 0); red >= kBlock/kWave floats
__device__ __forceinline__ float block_sum(float v, int tid, float* red) {
  v = wave_group_sum(v, kWave);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) { v = red[0]; for (int w = 1; w < kBlock / kWave; ++w) v += red[w]; }
  return v;
}

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

// XCD-aware workgroup -> (sample, part) map.  MI355X deals workgroup ids round-robin over its 8 XCDs (id % 8 labels the
// workgroups that share an XCD and therefore an L2).  Every workgroup of sample b gets id % 8 == b % 8, so the per-sample
// planes / masks / gates that all parts of a sample re-read are fetched into ONE L2 instead of all eight (PMC before:
// +38 % read traffic in k_bwd_reduce2).  Grids are ceil(B/8)*8*P wide; ids that map past the last sample exit.  Placement
// is a speed assumption only -- results do not depend on it.
__device__ __forceinline__ bool xcd_sample_part(int bid, int B, int P, int& b, int& part) {
  const int x = bid & 7, q = bid >> 3;
  const int g8 = q / P;
  part = q - g8 * P;
  b = g8 * 8 + x;
  return b < B;
}

// ---------------------------------------------------------------------------------------------
// In-launch hand-off between workgroups (cdna_hip_programming.md Guideline 16), used by the x-resident kernels: a
// workgroup publishes a few hundred bytes (its plane rows, its partial sums) that OTHER workgroups of the same launch
// consume.  The handed-off words themselves are written and read with relaxed AGENT-scope atomics (sc1 accesses: written
// through / fetched past the non-coherent caches), so no whole-cache release/acquire is needed -- measured: an agent
// release+acquire pair per workgroup (buffer_wbl2 + buffer_inv) made a 80 us forward 220 us.
//   producer:  st_agent(...) -> EVERY wave s_waitcnt vmcnt(0) -> barrier -> one lane bumps the flag (relaxed agent add)
//   consumer:  lanes poll the flags (relaxed agent loads, s_sleep) -> s_waitcnt -> barrier -> ld_agent(...) of the data
// Flags are generation counters: every workgroup bumps its own flag exactly once per call, so after call n a flag reads n;
// nothing is reset (the region is zero-filled once by the caller) and graph replay needs no per-launch argument.
// Every spin is bounded: on time-out the status word is set, the workgroup poisons its outputs (NaN) and carries on, so the
// launch always drains and the failure cannot stay silent.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// all of this workgroup's st_agent stores are complete -> bump its flag (fire and forget: the generation this call will
// reach is known beforehand, own flag + 1, read at the top of the kernel where its latency is hidden)
__device__ __forceinline__ void handoff_publish(int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until flags[lo..hi] have all reached generation `gen`.  Returns true (uniformly over the workgroup) when a wait timed
// out: the caller then POISONS what it writes (NaN), so a hand-off that did not happen is loud in the loss, never a silently
// stale halo; the status word `err` is set as well (read by the host wherever it synchronises: mgacbam_ctx_layout_t.status).
__device__ __forceinline__ bool handoff_wait(const int* flags, int lo, int hi, int gen, int* err, unsigned spin_limit) {
  int timed_out = 0;
  for (int t = lo + static_cast<int>(threadIdx.x); t <= hi; t += kBlock) {
    unsigned spins = 0;
    // wrap-safe: the counters run for the life of ctx (2^31 calls = days of training), compare modulo 2^32
    while (static_cast<int>(static_cast<unsigned>(ld_agent(flags + t)) - static_cast<unsigned>(gen)) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > spin_limit) { st_agent(err, 1); timed_out = 1; break; }   // default 2^20 ~ 1 s: give up, flag it, let the launch drain
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return __syncthreads_or(timed_out) != 0;
}

// the same for a 2-D set of tiles: in each image row ra..rb the tiles (runs of TP pixels) that hold columns cl..ch
__device__ __forceinline__ bool handoff_wait_rows(const int* flags, int ra, int rb, int W, int TP, int cl, int ch, int gen, int* err,
                                                  unsigned spin_limit) {
  int timed_out = 0;
  const int per = (ch - cl) / TP + 2;                         // upper bound of the tiles one row's column range touches
  for (int i = static_cast<int>(threadIdx.x); i < (rb - ra + 1) * per; i += kBlock) {
    const int row = ra + i / per, j = i - (i / per) * per;
    const int t = (row * W + cl) / TP + j;
    if (t > (row * W + ch) / TP) continue;
    unsigned spins = 0;
    while (static_cast<int>(static_cast<unsigned>(ld_agent(flags + t)) - static_cast<unsigned>(gen)) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > spin_limit) { st_agent(err, 1); timed_out = 1; break; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return __syncthreads_or(timed_out) != 0;
}

// poor man's thread trace (builds with -DMGACBAM_TRACE, tools/trace_gate.py): thread 0 of a workgroup records the 100 MHz
// wall clock at phase boundaries, slot 15 holds the hardware id (XCD / SE / CU)
#ifdef MGACBAM_TRACE
#define TRACE_MARK(buf, gid, slot) do { if ((buf) && threadIdx.x == 0) (buf)[static_cast<size_t>(gid) * 16 + (slot)] = static_cast<long long>(wall_clock64()); } while (0)
#define TRACE_HWID(buf, gid) do { if ((buf) && threadIdx.x == 0) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); (buf)[static_cast<size_t>(gid) * 16 + 15] = (static_cast<long long>(xcc) << 32) | hw; } } while (0)
#else
#define TRACE_MARK(buf, gid, slot) do {} while (0)
#define TRACE_HWID(buf, gid) do {} while (0)
#endif

// level of a grouped launch that owns workgroup `bid`; returns the id relative to that level in `local`
template <typename G>
__device__ __forceinline__ int find_level(const G& g, int bid, int& local) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < kGroupMax; ++i)
    if (i < g.n && bid >= g.start[i]) l = i;
  local = bid - g.start[l];
  return l;
}

}  // namespace mgacbam
