// Shared device/host helpers for libttround_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/ttround_hip.h"

namespace ttr {

constexpr int kThreads = 256;  // every kernel here runs 4 wave64 per workgroup
constexpr int kWave = 64;

// ------------------------------------------------------------------ errors
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
#define TTR_HIP_CHECK(expr)                                  \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return ::ttr::hip_fail(_e, #expr); \
  } while (0)
#define TTR_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      ::ttr::set_error(__VA_ARGS__);  \
      return (code);                  \
    }                                 \
  } while (0)

// ------------------------------------------------------------------ profiling (HIP events per kernel kind)
struct ProfScope {
  int kind;
  hipStream_t stream;
  void* slot;
  ProfScope(int kind, hipStream_t s);
  ~ProfScope();
};

// ------------------------------------------------------------------ MFMA 16x16x4 wrappers
// A operand: lane l holds A[i = l & 15][k = l >> 4]; B operand: B[k = l >> 4][j = l & 15].
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Mfma;

template <>
struct Mfma<float> {
  using Acc = f32x4;
  static __device__ __forceinline__ Acc zero() { return Acc{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ Acc mma(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) * 4 + reg; }
};

template <>
struct Mfma<double> {
  using Acc = f64x4;
  static __device__ __forceinline__ Acc zero() { return Acc{0., 0., 0., 0.}; }
  static __device__ __forceinline__ Acc mma(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // f64 C/D: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};

// ------------------------------------------------------------------ LDS-only workgroup barrier
// __syncthreads() also drains vmcnt: every in-flight GLOBAL store / prefetch load has to land before any
// wave may pass.  In the sequential Householder steps that exposes the full HBM write latency of the
// fire-and-forget reflector stores once per step (measured: 55 % of wave cycles in SQ_WAIT_ANY, ~2800 cycles
// per step for ~100 VALU instructions), and in the apply kernel it kills the panel prefetch.  When only LDS
// traffic has to be ordered, waiting for lgkmcnt alone is sufficient.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------ wave reductions
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over the 64 lanes, result in every lane.  Row (16-lane) sums with four DPP steps -- VALU only, no
// LDS crossbar -- then the four row sums are combined through SGPRs (readlane).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ float lane_get(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double lane_get(double v, int l) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}

// Four independent wave sums at once: the DPP / readlane chains of the four values interleave.
template <typename T>
__device__ __forceinline__ void wave_sum_dpp4(T (&v)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] += dpp_mov<0xB1>(v[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] += dpp_mov<0x4E>(v[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] += dpp_mov<0x141>(v[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] += dpp_mov<0x140>(v[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = (lane_get(v[c], 0) + lane_get(v[c], 16)) + (lane_get(v[c], 32) + lane_get(v[c], 48));
}

// ------------------------------------------------------------------ 16-lane-row primitives (MFMA C/D layout helpers)
// In the 16x16 accumulator layout lane = (g = lane >> 4, cl = lane & 15) holds rows {4g..4g+3} (f32) of column cl.
// row_bcast<J>(v): every lane receives v of lane (g, J) of its own 16-lane row (DPP row_newbcast: VALU only).
template <int J>
__device__ __forceinline__ float row_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + J, 0xF, 0xF, false));
}
template <int J>
__device__ __forceinline__ double row_bcast(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), 0x150 + J, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + J, 0xF, 0xF, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
// Sum over the four lanes {cl, cl+16, cl+32, cl+48}; result in all four.  gfx950 v_permlane16_swap / v_permlane32_swap
// exchange whole 16-lane rows between two registers (VALU only); `safe` uses ds_bpermute shuffles instead.
__device__ __forceinline__ float xrow_sum(float v, bool safe) {
  if (safe) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
  }
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  unsigned a = __builtin_bit_cast(unsigned, v);
  u32x2 s = __builtin_amdgcn_permlane16_swap(a, a, false, false);
  v = __builtin_bit_cast(float, s[0]) + __builtin_bit_cast(float, s[1]);
  a = __builtin_bit_cast(unsigned, v);
  s = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  return __builtin_bit_cast(float, s[0]) + __builtin_bit_cast(float, s[1]);
}
__device__ __forceinline__ double xrow_sum(double v, bool) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <typename T>
struct Num;
template <>
struct Num<float> {
  static __device__ __forceinline__ float eps() { return 1.1920929e-07f; }
  static __device__ __forceinline__ float tiny() { return 1.17549435e-38f; }
  static __device__ __forceinline__ float big_theta() { return 1e18f; }
};
template <>
struct Num<double> {
  static __device__ __forceinline__ double eps() { return 2.220446049250313e-16; }
  static __device__ __forceinline__ double tiny() { return 2.2250738585072014e-308; }
  static __device__ __forceinline__ double big_theta() { return 1e150; }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

}  // namespace ttr
