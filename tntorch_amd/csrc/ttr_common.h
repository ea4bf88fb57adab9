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

// ------------------------------------------------------------------ executed-work census (ttr_prof_enable(2))
// The per-kind device times of the profiler say nothing about how much of a kernel's ALGORITHMIC work a launch performed: the
// kernels decide per item / per block, on the device, to skip work the input does not need (rank-skipped panels and packed rows
// of the QR blocks, `rows32` items of the Gram / projection kernels, pass-through items of the second Gram pass).  In census
// mode every instrumented dispatcher enqueues, OUTSIDE its timed scope, a tiny kernel that reads the same device-side decisions
// (taus, flags) and adds the flops / bytes the launch really executed to per-kind device counters (ttr_prof_collect_work).
bool work_census_on();
// adds fl[c] / by[c] for every item b of the batch, c = (f1 && f1[b] ? 1 : 0) + (f2 && f2[b] ? 2 : 0)
void work_items(int kind, const int32_t* f1, const int32_t* f2, int64_t batch, const double fl[4], const double by[4], hipStream_t s);
// QR blocks: `tau` holds nblk blocks of NP reflector scalars; a block's q live 16-column panels (any tau != 0) cost
//   factor (kc == 0): 2 r c^2 - 2 c^3 / 3 + 4 r c (n - c), c = 16 q (Householder panels + update of the other columns)
//   apply  (kc  > 0): 4 r c kc                               (W = V^T C and C -= V (T W))
// with r = the block's rows: rpb, the last block of an item (nb blocks per item, m rows) what is left
// bytes: live reflectors (r c) + R (n^2) written by a factoring block, + its r x n input when `reads_input` (every launch but
// the fused push, whose core / Rm reads are charged per item); an applying block reads its live reflectors and writes r x kc
void work_qr_taus(int kind, const void* tau, bool f64, int64_t nblk, int64_t nb_per_item, int NP, int64_t m, int64_t rpb, int n,
                  int kc, bool reads_input, hipStream_t s);

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

// ------------------------------------------------------------------ multi-value wave reductions
// A wave issues at most one VALU instruction per ~8 cycles (measured: 8.0 cycles per independent v_fma for a wave alone on
// its SIMD), so on the sequential Householder chain the number of issue slots of a reduction IS its latency.  Round 2's first
// attempt at folding several values into one register with gfx950's v_permlane32/16_swap went through the builtins (operand
// copies around every swap: ~27-32 slots for four values) and lost against interleaved DPP chains (14 / 28 slots for two / four
// values); written as in-place inline asm the folds need 9 / 14 (round 3, below).
//
// a, b <- their sums over the 64 lanes (wave-uniform).  Round 3: ONE half exchange folds both values into one register
// (v_permlane32_swap a, b leaves [a.lo | b.lo] in a and [a.hi | b.hi] in b: their sum holds a's 32 pair sums in lanes 0-31 and b's in
// lanes 32-63), then one DPP chain reduces both halves at once: 1 swap + 1 add + 5 DPP + 2 v_readlane = 9 issue slots instead of
// the 14 of two interleaved chains (12 DPP + 2 v_readlane) -- a lone wave issues one instruction per ~8 clocks, so on the
// Householder chain slots ARE latency.  (inline asm: hipcc 7.2 miscompiles the swap builtin when both operands hold the same
// value, and the asm keeps the swap in place, without operand copies.)
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
  float w = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1\n\ts_nop 1" : "+v"(w), "+v"(b));
  w += dpp_mov<0xB1>(w);   // quad_perm [1,0,3,2]
  w += dpp_mov<0x4E>(w);   // quad_perm [2,3,0,1]
  w += dpp_mov<0x141>(w);  // row_half_mirror
  w += dpp_mov<0x140>(w);  // row_mirror: every lane holds its row's sum
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1" : "+v"(w));  // rows 1, 3 += rows 0, 2
  a = lane_get(w, 31);
  b = lane_get(w, 63);
}
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
  a = wave_sum_dpp(a);
  b = wave_sum_dpp(b);
}
// v[0..3] <- their sums over the 64 lanes (wave-uniform).  Round 3: two half exchanges fold (v0, v1) and (v2, v3) into one register
// each ([V0 | V1], [V2 | V3] by halves), one row exchange (v_permlane16_swap: odd rows of the first with even rows of the second
// operand) folds those into ONE register whose four rows carry V0, V2, V1, V3, and a single 4-step DPP row reduction finishes all
// four: 3 swaps + 3 adds + 4 DPP + 4 v_readlane = 14 issue slots instead of 28 (24 DPP + 4 v_readlane).
__device__ __forceinline__ void wave_sum4(float (&v)[4]) {
  float x = v[0], y = v[2];
  asm volatile("s_nop 1\n\t"
               "v_permlane32_swap_b32 %0, %2\n\t"
               "v_permlane32_swap_b32 %1, %3\n\t"
               "v_add_f32 %0, %0, %2\n\t"
               "v_add_f32 %1, %1, %3\n\t"
               "s_nop 1\n\t"
               "v_permlane16_swap_b32 %0, %1\n\t"
               "v_add_f32 %0, %0, %1\n\t"
               "s_nop 1"   // (the compiler cannot see that the asm ends on a VALU write the DPP step below reads)
               : "+v"(x), "+v"(y), "+v"(v[1]), "+v"(v[3]));
  x += dpp_mov<0xB1>(x);
  x += dpp_mov<0x4E>(x);
  x += dpp_mov<0x141>(x);
  x += dpp_mov<0x140>(x);
  v[0] = lane_get(x, 0);
  v[2] = lane_get(x, 16);
  v[1] = lane_get(x, 32);
  v[3] = lane_get(x, 48);
}
__device__ __forceinline__ void wave_sum4(double (&v)[4]) { wave_sum_dpp4(v); }

template <typename T>
struct Num;
template <>
struct Num<float> {
  static __device__ __forceinline__ float eps() { return 1.1920929e-07f; }
  static __device__ __forceinline__ float tiny() { return 1.17549435e-38f; }
  static __device__ __forceinline__ float big_theta() { return 1e18f; }
  // Householder: a sub-column whose squared norm lies below 2^-100 of a block factored at the exponent of its largest entry is
  // treated as zero (H = I): 2^-50 of the block, far below eps -- and the hardware sqrt / rcp flush denormal arguments
  // (alpha^2 + ss ~ 1e-39 gave beta = 0, tau = inf: a rank-64 bond with sigma_j ~ 2^-j, SURVEY 8d's decaying variant)
  static __device__ __forceinline__ float larfg_floor() { return 7.8886091e-31f; }
};
template <>
struct Num<double> {
  static __device__ __forceinline__ double eps() { return 2.220446049250313e-16; }
  static __device__ __forceinline__ double tiny() { return 2.2250738585072014e-308; }
  static __device__ __forceinline__ double big_theta() { return 1e150; }
  static __device__ __forceinline__ double larfg_floor() { return 1e-290; }
};

// Rank rule of round.py:147-158 on singular values sorted decreasing: drop the longest tail whose energy is <= delta^2 (`<=`),
// rank = max(1, min(rmax, n - tail)); rank 0 = the zero guard of round.py:137-145.  `s` holds n_live values, entries
// n_live .. n_full - 1 are exact zeros that were never computed (zero-tail eigenproblems).  `noise_c` > 0
// (TTR_KNOB_RANK_NOISE_FLOOR): the rule sees every value at no less than noise_c eps sigma_0 -- what LAPACK's gesdd returns for
// the null directions of a rank-deficient unfolding (the reference's ranks depend on that noise).
template <typename T>
__device__ __forceinline__ int rank_rule(const T* s, int n_live, int n_full, int64_t rmax, int use_delta, T d2, int noise_c) {
  const int64_t cap = rmax < (int64_t)n_full ? rmax : (int64_t)n_full;
  if (s[0] < T(1e-13)) return 0;
  if (!use_delta) return (int)(cap < 1 ? 1 : cap);
  const T fl = noise_c > 0 ? T(noise_c) * Num<T>::eps() * s[0] : T(0);
  double acc = 0.0;
  int tail = 0;
  for (int k = n_full - 1; k >= 0; --k) {
    T v = k < n_live ? s[k] : T(0);
    if (v < fl) v = fl;
    acc += (double)(v * v);
    if ((T)acc <= d2) tail = n_full - k; else break;
  }
  int64_t rk = n_full - tail;
  if (rk > cap) rk = cap;
  if (rk < 1) rk = 1;
  return (int)rk;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

}  // namespace ttr
