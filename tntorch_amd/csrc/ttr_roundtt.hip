// ttr_round_tt: the WHOLE rounding of a (sub-)batch of tensor trains behind ONE C call -- the left-to-right orthogonalisation
// loop of tensor.py:1905-1906 (QR of the left unfolding, R pushed into the next core: tensor.py:1816-1832) followed by the
// right-to-left truncation loop of tensor.py:2053-2083 (truncated SVD of the right unfolding, round.py:52-187; U sigma pushed into
// the previous core).  The reference runs both loops in Python, one torch operator at a time; rounds 1-4 of this library did the
// same with one ctypes call per kernel (~80 per 64^8 train: 1.2 ms of host time per call at B = 1).  Here the chain is enqueued
// from C++: the SAME library entries in the SAME order as tntorch_amd/_hipops.py::_round_tt_sweep (the results are bit-identical,
// tests/test_gpu_parity.py pins that), every temporary carved out of ONE caller-owned workspace, no allocation, no host
// synchronisation, nothing read back.
//
// Envelope (what the fused kernels cover; everything else returns TTR_E_UNSUPPORTED from the planner and the Python host keeps
// its own loop): every TT rank <= 64 columns of a TSQR panel, every pushed factorisation supported (k I >= n), every bond a
// <= 64-row matrix with at least as many columns (the fused row-Gram / projection kernels).
#include <math.h>

#include <chrono>
#include <cstdlib>

#include "ttr_common.h"

namespace ttr {
namespace {

constexpr int kMaxCores = 64;
constexpr int64_t kNoCap = 2147483647;   // rank caps at or above this mean "none" (round.py:83-84)

struct SweepPlan {
  int dt = 0;
  int64_t es = 4, N = 0, B = 0;
  // left-to-right: factorisation mu = 0 .. N-2 of the (rows_k[mu] * I[mu]) x r1[mu] left unfolding of (R_{mu-1} x core mu)
  int64_t r0[kMaxCores], I[kMaxCores], r1[kMaxCores];
  int64_t rows_k[kMaxCores];   // rows of the R factor entering core mu (r0 of core 0)
  int64_t kq[kMaxCores];       // rows of the R factor leaving core mu
  int64_t qr_wsb[kMaxCores];
  int64_t flag_off[kMaxCores]; // byte offset of the rows32 flags inside the QR workspace (-1: none)
  // right-to-left: bond mu = N-1 .. 1
  int64_t bR[kMaxCores], bn[kMaxCores], bcap[kMaxCores], brq[kMaxCores];
  // workspace offsets (bytes)
  int64_t off_qr[kMaxCores], off_R[kMaxCores], off_Rn[kMaxCores];
  int64_t off_carry = 0, off_carryn = 0, off_expo = 0, off_escr = 0, off_r32 = 0;
  int64_t off_G = 0, off_G2 = 0, off_V1 = 0, off_V = 0, off_sig1 = 0, off_sig = 0, off_info1 = 0, off_info = 0, off_flat = 0;
  int64_t off_eigws = 0, eig_wsb = 0, off_gemmws = 0, gemm_wsb = 0;
  int64_t off_M = 0, off_left[3] = {0, 0, 0}, off_d2 = 0, off_nrm = 0, off_orth = 0, orth_wsb = 0;
  int64_t total = 0;
};

int64_t take(int64_t& off, int64_t bytes) {
  const int64_t at = off;
  off += align_up(bytes > 0 ? bytes : 1, 256);
  return at;
}

// Shapes + workspace layout.  Returns TTR_OK, or TTR_E_UNSUPPORTED (silently: no error text is needed for a capability probe,
// but one is set for callers that go on regardless).
int make_sweep_plan(int dtype, int64_t N, const int64_t* shapes, const int64_t* rcap, int64_t batch, int eps_mode, SweepPlan& p) {
  TTR_REQUIRE(dtype == TTR_F32 || dtype == TTR_F64, TTR_E_INVALID, "ttr_round_tt: bad dtype %d", dtype);
  TTR_REQUIRE(N >= 2 && N <= kMaxCores, TTR_E_UNSUPPORTED, "ttr_round_tt: %lld cores outside [2, %d]", (long long)N, kMaxCores);
  TTR_REQUIRE(batch >= 1 && batch <= 65535, TTR_E_UNSUPPORTED, "ttr_round_tt: batch %lld outside [1, 65535]", (long long)batch);
  TTR_REQUIRE(!eps_mode || batch == 1, TTR_E_UNSUPPORTED, "ttr_round_tt: the eps-mode sweep rounds ONE train (tensor.py:2039-2051)");
  p.dt = dtype; p.es = dtype == TTR_F64 ? 8 : 4; p.N = N; p.B = batch;
  const int64_t maxc = ttr_qr_max_cols(dtype);
  for (int64_t mu = 0; mu < N; ++mu) {
    p.r0[mu] = shapes[3 * mu]; p.I[mu] = shapes[3 * mu + 1]; p.r1[mu] = shapes[3 * mu + 2];
    TTR_REQUIRE(p.r0[mu] >= 1 && p.I[mu] >= 1 && p.r1[mu] >= 1, TTR_E_INVALID, "ttr_round_tt: empty core %lld", (long long)mu);
    TTR_REQUIRE(mu == 0 || p.r0[mu] == p.r1[mu - 1], TTR_E_INVALID, "ttr_round_tt: ranks of cores %lld and %lld do not chain",
                (long long)(mu - 1), (long long)mu);
  }
  int64_t off = 0;
  int64_t k = p.r0[0];
  for (int64_t mu = 0; mu + 1 < N; ++mu) {
    const int64_t n = p.r1[mu];
    TTR_REQUIRE(n <= maxc, TTR_E_UNSUPPORTED, "ttr_round_tt: TT rank %lld above the %lld columns of a TSQR panel", (long long)n,
                (long long)maxc);
    p.rows_k[mu] = k;
    const int64_t m = k * p.I[mu];
    p.flag_off[mu] = -1;
    if (mu == 0) {
      p.qr_wsb[mu] = ttr_qr_workspace_bytes(dtype, m, n, batch);
    } else {
      // (the fused push's own envelope, pushed_ok in ttr_qr.hip: all three conditions, so that a train outside it is declined HERE --
      // TTR_E_UNSUPPORTED from the planner, the caller keeps its loop over the per-kernel entries -- and not in the middle of the sweep)
      TTR_REQUIRE(k <= 64 && p.r0[mu] <= 64 && m >= n && p.r0[mu] * p.I[mu] * n < (int64_t(1) << 31), TTR_E_UNSUPPORTED,
                  "ttr_round_tt: core %lld outside the fused push (k = %lld, Rin = %lld, k I = %lld < n = %lld, or Rin I n >= 2^31)", (long long)mu,
                  (long long)k, (long long)p.r0[mu], (long long)m, (long long)n);
      p.qr_wsb[mu] = ttr_qr_pushed_workspace_bytes(dtype, p.I[mu], n, batch);
      if (k == 64) p.flag_off[mu] = ttr_qr_pushed_flag_offset(dtype, p.I[mu], n, batch);
    }
    p.kq[mu] = m < n ? m : n;
    p.off_qr[mu] = take(off, p.qr_wsb[mu]);
    p.off_R[mu] = take(off, batch * p.kq[mu] * n * p.es);
    p.off_Rn[mu] = p.off_R[mu];   // (rounds 1 - 4: a second buffer for the normalised R; the factor kernel normalises in place now)
    k = p.kq[mu];
  }
  p.rows_k[N - 1] = k;
  TTR_REQUIRE(k <= 64, TTR_E_UNSUPPORTED, "ttr_round_tt: the last carry has %lld rows (fused truncation: <= 64)", (long long)k);
  // right-to-left shapes
  int64_t rq = p.r1[N - 1];
  int64_t maxGR = 1, maxM = 1, maxleft = 1, maxparts = 1;
  for (int64_t mu = N - 1; mu >= 1; --mu) {
    const int64_t R = p.rows_k[mu], n = p.I[mu] * rq;
    TTR_REQUIRE(R <= 64 && R <= n, TTR_E_UNSUPPORTED, "ttr_round_tt: bond %lld is %lld x %lld (fused truncation: rows <= 64, rows <= columns)",
                (long long)mu, (long long)R, (long long)n);
    int64_t cap = rcap ? rcap[mu - 1] : kNoCap;
    if (cap > R) cap = R;
    if (cap < 1) cap = 1;
    p.bR[mu] = R; p.bn[mu] = n; p.bcap[mu] = cap; p.brq[mu] = rq;
    const int64_t parts = ttr_sweep_gram_parts(n, batch);
    if (parts > maxparts) maxparts = parts;
    if (R > maxGR) maxGR = R;
    if (mu < N - 1 && batch * R * n > maxM) maxM = batch * R * n;   // apply output [B, rows_k * I, rq]
    if (batch * R * cap > maxleft) maxleft = batch * R * cap;
    const int64_t ew = ttr_eigh_workspace_bytes(dtype, R, batch);
    if (ew > p.eig_wsb) p.eig_wsb = ew;
    const int64_t ow = ttr_orth_fixup_workspace_bytes(dtype, cap, n, batch, 1);
    if (ow > p.orth_wsb) p.orth_wsb = ow;
    rq = cap;
  }
  // the carry M = R_{N-2} x (last core)
  const int64_t cM = p.rows_k[N - 1], cN = p.I[N - 1] * p.r1[N - 1];
  p.off_carry = take(off, batch * cM * cN * p.es);
  p.off_carryn = dtype == TTR_F32 ? take(off, batch * cM * cN * p.es) : p.off_carry;
  p.gemm_wsb = ttr_gemm_workspace_bytes(dtype, cM, cN, p.r0[N - 1], batch);
  p.off_gemmws = take(off, p.gemm_wsb);
  p.off_expo = take(off, batch * 4);
  p.off_escr = take(off, batch * 4);
  p.off_r32 = take(off, batch * 4);
  p.off_G = take(off, batch * maxparts * maxGR * maxGR * p.es);
  p.off_G2 = take(off, batch * maxparts * maxGR * maxGR * p.es);
  p.off_V1 = take(off, batch * maxGR * maxGR * p.es);
  p.off_V = take(off, batch * maxGR * maxGR * p.es);
  p.off_sig1 = take(off, batch * maxGR * p.es);
  p.off_sig = take(off, batch * maxGR * p.es);
  p.off_info1 = take(off, batch * 4);
  p.off_info = take(off, batch * 4);
  p.off_flat = take(off, batch * 4);
  p.off_eigws = take(off, p.eig_wsb);
  p.off_M = take(off, maxM * p.es);
  for (int i = 0; i < 3; ++i) p.off_left[i] = take(off, maxleft * p.es);
  p.off_orth = take(off, p.orth_wsb);
  p.off_d2 = take(off, 8);
  p.off_nrm = take(off, 64 + 4096 * 8);   // [0, 64): the norm; behind it: up to 4096 chunk norms (norm_one)
  p.total = off;
  return TTR_OK;
}

template <typename T>
__global__ void delta2_kernel(const T* __restrict__ nrm, double factor, double* __restrict__ d2) {
  // tensor.py:2039-2051 without the `.item()`: delta^2 = (eps / max(1, sqrt(N - 1)))^2 ||last core||^2, in double
  const double v = (double)nrm[0];
  d2[0] = (v * v) * factor;
}

__global__ void izero_kernel(int32_t* __restrict__ x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) x[i] = 0;
}

// eps mode: the selected ranks, copied to a second (device-accessible, typically pinned host) array as soon as the last bond is decided
__global__ void icopy_sys_kernel(const int32_t* __restrict__ x, int n, int32_t* __restrict__ out) {
  if ((int)threadIdx.x < n) out[threadIdx.x] = x[threadIdx.x];
  __threadfence_system();
}

__global__ void imax_kernel(const int32_t* __restrict__ x, int64_t n, int32_t* __restrict__ out) {
  __shared__ int red[kThreads / kWave];
  int m = -2147483647 - 1;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) m = x[i] > m ? x[i] : m;
  for (int o = 32; o >= 1; o >>= 1) {
    const int v = __shfl_xor(m, o);
    m = v > m ? v : m;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / kWave; ++w) m = red[w] > m ? red[w] : m;
    out[0] = m;
    __threadfence_system();   // (`out` may be pinned host memory that the caller polls: ttr_round_tt's zero_flag_dev)
  }
}

// Frobenius norm of ONE contiguous item, chunked exactly as tntorch_amd/_hip.py::norm chunks it (long items stream over the whole
// chip; the second stage squares and sums in double)
int norm_one(int dt, int64_t count, const void* x, void* out, void* scratch, hipStream_t st) {
  if (count >= (1 << 20)) {
    int64_t k = 4096;
    while (k > 1 && count % k) k /= 2;
    if (k > 1) {
      const int rc = ttr_norm(dt, count / k, k, x, count / k, scratch, st);
      if (rc != TTR_OK) return rc;
      return ttr_norm(dt, k, 1, scratch, k, out, st);
    }
  }
  return ttr_norm(dt, count, 1, x, count, out, st);
}

// (TTR_SWEEP_PACE, diagnostics: 1 = hipStreamQuery after every library call, n > 1 = busy-wait n microseconds -- does the depth of the
// queue cost the device time?)
static int sweep_pace() {
  static const int v = [] { const char* e = getenv("TTR_SWEEP_PACE"); return e ? atoi(e) : 0; }();
  return v;
}
static void pace(hipStream_t st) {
  const int v = sweep_pace();
  if (v == 1) (void)hipStreamQuery(st);
  else if (v > 1) {
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < (double)v) {}
  }
}
#define TTR_TRY(expr)                \
  do {                               \
    const int _rc = (expr);          \
    if (_rc != TTR_OK) return _rc;   \
    if (sweep_pace()) pace(st);      \
  } while (0)

}  // namespace
}  // namespace ttr

using namespace ttr;

extern "C" {

int64_t ttr_round_tt_workspace_bytes(int dtype, int64_t N, const int64_t* shapes, const int64_t* rcap, int64_t batch, int eps_mode) {
  if (!shapes) return TTR_E_INVALID;
  SweepPlan p;
  const int rc = make_sweep_plan(dtype, N, shapes, rcap, batch, eps_mode, p);
  return rc == TTR_OK ? p.total : (int64_t)rc;
}

int ttr_round_tt(int dtype, int64_t N, const int64_t* shapes, int64_t batch, const void* const* cores_in, const int64_t* rcap,
                 int algorithm, int eps_mode, double eps, double flat_thr, int use_eigh_top, void* const* cores_out,
                 int32_t* ranks_dev, int32_t* zero_flag_dev, void* workspace, int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(shapes && cores_in && cores_out, TTR_E_INVALID, "ttr_round_tt: null pointer");
  TTR_REQUIRE(algorithm == TTR_ALG_SVD || algorithm == TTR_ALG_EIG, TTR_E_INVALID, "ttr_round_tt: algorithm %d", algorithm);
  SweepPlan p;
  hipStream_t st = (hipStream_t)stream;
  TTR_TRY(make_sweep_plan(dtype, N, shapes, rcap, batch, eps_mode, p));
  TTR_REQUIRE(workspace && workspace_bytes >= p.total, TTR_E_WORKSPACE, "ttr_round_tt: workspace %lld < %lld bytes",
              (long long)workspace_bytes, (long long)p.total);
  TTR_REQUIRE(!eps_mode || ranks_dev, TTR_E_INVALID, "ttr_round_tt: the eps-mode sweep needs ranks_dev[N - 1]");
  for (int64_t mu = 0; mu < N; ++mu)
    TTR_REQUIRE(cores_in[mu] && cores_out[mu], TTR_E_INVALID, "ttr_round_tt: null core %lld", (long long)mu);
  char* ws = (char*)workspace;
  const int dt = dtype;
  const int64_t es = p.es, B = batch;
  const bool f32 = dt == TTR_F32;
  const bool svd = algorithm == TTR_ALG_SVD;
  int32_t* expo = (int32_t*)(ws + p.off_expo);
  int32_t* escr = (int32_t*)(ws + p.off_escr);
  if (f32) {
    hipLaunchKernelGGL(izero_kernel, dim3((unsigned)ceil_div(B, kThreads)), dim3(kThreads), 0, st, expo, B);
    TTR_HIP_CHECK(hipGetLastError());
  }

  // ---------------------------------------------------------------- left to right (tensor.py:1905-1906; Q stays implicit)
  const void* Rprev = nullptr;   // [B, rows, r0 of the next core]
  for (int64_t mu = 0; mu + 1 < N; ++mu) {
    const int64_t n = p.r1[mu], k = p.rows_k[mu], kq = p.kq[mu];
    void* R = ws + p.off_R[mu];
    void* qws = ws + p.off_qr[mu];
    // fp32: every R back to O(1) by an exact power of two per item, exponents summed on the device (_round_tt_sweep) -- by the
    // factor kernel itself (ABI 11: no ttr_pow2_normalize launch per core)
    if (mu == 0) {
      const int64_t m = k * p.I[0];
      if (f32) TTR_TRY(ttr_qr_factor_expo(dt, m, n, B, cores_in[0], n, m * n, R, n, kq * n, qws, p.qr_wsb[0], expo, st));
      else TTR_TRY(ttr_qr_factor(dt, m, n, B, cores_in[0], n, m * n, R, n, kq * n, qws, p.qr_wsb[0], st));
    } else if (f32) {
      TTR_TRY(ttr_qr_factor_pushed_expo(dt, k, p.r0[mu], p.I[mu], n, B, Rprev, p.r0[mu], k * p.r0[mu], cores_in[mu],
                                        p.r0[mu] * p.I[mu] * n, R, n, kq * n, qws, p.qr_wsb[mu], expo, st));
    } else {
      TTR_TRY(ttr_qr_factor_pushed(dt, k, p.r0[mu], p.I[mu], n, B, Rprev, p.r0[mu], k * p.r0[mu], cores_in[mu],
                                   p.r0[mu] * p.I[mu] * n, R, n, kq * n, qws, p.qr_wsb[mu], st));
    }
    Rprev = R;
  }
  // the first truncation's carry M = R x (last core)
  const int64_t cM = p.rows_k[N - 1], cK = p.r0[N - 1], cN = p.I[N - 1] * p.r1[N - 1];
  void* carry = ws + p.off_carry;
  TTR_TRY(ttr_gemm(dt, 0, 0, cM, cN, cK, Rprev, cK, cM * cK, cores_in[N - 1], cN, cK * cN, carry, cN, cM * cN, nullptr, 0,
                   TTR_SCALE_NONE, nullptr, 0, TTR_SCALE_NONE, B, p.gemm_wsb > 0 ? (void*)(ws + p.off_gemmws) : nullptr, p.gemm_wsb, st));
  int32_t* r32_last = nullptr;
  if (cM == 64 && cN >= 64) {
    r32_last = (int32_t*)(ws + p.off_r32);
    TTR_TRY(ttr_carry_rows32(dt, cN, B, carry, cN, cM * cN, r32_last, st));
  }
  if (f32) {
    void* cn = ws + p.off_carryn;
    TTR_TRY(ttr_pow2_normalize(dt, cM * cN, B, carry, cM * cN, cn, cM * cN, escr, expo, st));
    carry = cn;
  }
  // eps mode: delta^2 on the device
  double* d2dev = nullptr;
  if (eps_mode) {
    d2dev = (double*)(ws + p.off_d2);
    void* nrm = ws + p.off_nrm;
    TTR_TRY(norm_one(dt, cM * cN, carry, nrm, (char*)nrm + 64, st));
    const double f = eps / fmax(1.0, sqrt((double)(N - 1)));
    if (f32) hipLaunchKernelGGL(delta2_kernel<float>, dim3(1), dim3(1), 0, st, (const float*)nrm, f * f, d2dev);
    else hipLaunchKernelGGL(delta2_kernel<double>, dim3(1), dim3(1), 0, st, (const double*)nrm, f * f, d2dev);
    TTR_HIP_CHECK(hipGetLastError());
  }

  // ---------------------------------------------------------------- right to left (tensor.py:2053-2083)
  const double epsT = f32 ? 1.1920928955078125e-07 : 2.220446049250313e-16;
  void* G = ws + p.off_G;
  void* G2 = ws + p.off_G2;
  void* V1 = ws + p.off_V1;
  void* V = ws + p.off_V;
  void* sig1 = ws + p.off_sig1;
  void* sig = ws + p.off_sig;
  int32_t* info1 = (int32_t*)(ws + p.off_info1);
  int32_t* flatb = (int32_t*)(ws + p.off_flat);
  void* eigws = p.eig_wsb > 0 ? (void*)(ws + p.off_eigws) : nullptr;
  const void* left = nullptr;   // (U sigma) of the bond to the right: [B, R, cap]
  int lslot = 0;
  for (int64_t mu = N - 1; mu >= 1; --mu) {
    const int64_t R = p.bR[mu], n = p.bn[mu], cap = p.bcap[mu], rq = p.brq[mu], I = p.I[mu];
    const void* M = carry;
    const int32_t* r32 = nullptr;
    if (mu < N - 1) {
      // core mu x (U sigma): the reflectors of factorisation mu applied to [U sigma; 0] (ttr_qr_apply*: tensor.py:2081-2083)
      void* Mo = ws + p.off_M;
      const bool rows32_ok = (R == 64 && n >= 64 && p.flag_off[mu] >= 0);
      // (mu >= 1 here: always a pushed factorisation)
      TTR_TRY(ttr_qr_apply_pushed(dt, R, I, p.r1[mu], B, ws + p.off_qr[mu], p.qr_wsb[mu], left, rq, p.r1[mu] * rq, rq, Mo, rq,
                                  R * I * rq, rows32_ok ? 1 : 0, st));
      M = Mo;
      if (rows32_ok) r32 = (const int32_t*)(ws + p.off_qr[mu] + p.flag_off[mu]);
    } else {
      r32 = (R == 64 && n >= 64) ? r32_last : nullptr;
    }
    int32_t* info = eps_mode ? ranks_dev + (mu - 1) : (int32_t*)(ws + p.off_info);
    const int64_t parts = ttr_sweep_gram_parts(n, B);
    const int64_t rcap_mu = rcap ? rcap[mu - 1] : kNoCap;
    const int64_t rmax_rule = rcap_mu < 1 ? 1 : (rcap_mu > kNoCap ? kNoCap : rcap_mu);
    TTR_TRY(ttr_rowgram(dt, R, n, B, M, n, R * n, G, parts, r32, st));
    const void* V1p = nullptr;
    if (svd) {
      const int32_t* flat = nullptr;
      // r of the top-r launch: the cap in batch mode (only with a cap), min(cap, 32) in eps mode (see _hipops.truncate)
      const int64_t r_top = eps_mode ? (cap < 32 ? cap : 32) : cap;
      if (use_eigh_top && flat_thr > 0.0 && (eps_mode || rcap_mu < kNoCap) && ttr_eigh_top_ok(R, r_top)) {
        // (eps mode: need_all -- the top-r path only where it computes every eigenpair; the certified flat test follows)
        TTR_TRY(ttr_eigh_top(dt, R, B, G, R, parts * R * R, parts, R * R, V1, R, R * R, sig1, R, info1, r_top, flat_thr, flatb,
                             eps_mode ? 1 : 0, st));
        if (eps_mode) {
          TTR_TRY(ttr_spectrum_flat(dt, R, B, sig1, R, cap, flat_thr, 1, 0.0, d2dev, flatb, r32, st));
        }
        flat = flatb;
      } else {
        TTR_TRY(ttr_eigh_trunc(dt, R, B, G, R, parts * R * R, parts, R * R, V1, R, R * R, sig1, R, info1, TTR_EIG_RAW, 0, 0.0, nullptr,
                               R, TTR_SOLVER_TRIDIAG, nullptr, nullptr, nullptr, 0, eigws, p.eig_wsb, st));
        if (flat_thr > 0.0) {
          TTR_TRY(ttr_spectrum_flat(dt, R, B, sig1, R, cap, flat_thr, eps_mode ? 1 : 0, 0.0, d2dev, flatb, r32, st));
          flat = flatb;
        }
      }
      TTR_TRY(ttr_rotgram(dt, R, n, B, M, n, R * n, V1, R, R * R, G2, parts, flat, r32, st));
      TTR_TRY(ttr_eigh_trunc(dt, R, B, G2, R, parts * R * R, parts, R * R, V, R, R * R, sig, R, info, TTR_EIG_RAW, eps_mode ? 1 : 0, 0.0,
                             d2dev, rmax_rule, TTR_SOLVER_JACOBI_LIVE, nullptr, flat, flat ? sig1 : nullptr, flat ? R : 0, eigws,
                             p.eig_wsb, st));
      V1p = V1;
    } else {
      TTR_TRY(ttr_eigh_trunc(dt, R, B, G, R, parts * R * R, parts, R * R, V, R, R * R, sig, R, info, TTR_EIG_REF, eps_mode ? 1 : 0, 0.0,
                             d2dev, rmax_rule, TTR_SOLVER_TRIDIAG, nullptr, nullptr, nullptr, 0, eigws, p.eig_wsb, st));
    }
    if (eps_mode && mu == 1 && zero_flag_dev && N - 1 <= 64) {   // every bond is decided: the caller's early copy of the ranks (see the header)
      hipLaunchKernelGGL(icopy_sys_kernel, dim3(1), dim3(64), 0, st, (const int32_t*)ranks_dev, (int)(N - 1), zero_flag_dev);
      TTR_HIP_CHECK(hipGetLastError());
    }
    if (!eps_mode && mu == N - 1 && zero_flag_dev) {   // zero guard of round.py:137-141 for the whole batch (read by the caller, later)
      hipLaunchKernelGGL(imax_kernel, dim3(1), dim3(kThreads), 0, st, (const int32_t*)info, B, zero_flag_dev);
      TTR_HIP_CHECK(hipGetLastError());
    }
    void* lnew = ws + p.off_left[lslot];
    lslot = (lslot + 1) % 3;
    TTR_TRY(ttr_project(dt, R, n, cap, B, M, n, R * n, V1p, R, R * R, V, R, R * R, sig, R, 1, cores_out[mu], n, cap * n, lnew, cap,
                        R * cap, r32, st));
    if (svd)
      TTR_TRY(ttr_orth_fixup(dt, cap, n, B, cores_out[mu], n, 1, cap * n, sig, R, (double)R * epsT, eps_mode ? info : nullptr,
                             p.orth_wsb > 0 ? (void*)(ws + p.off_orth) : nullptr, p.orth_wsb, st));
    if (eps_mode) TTR_TRY(ttr_mask_cols(dt, R, cap, B, lnew, cap, R * cap, info, st));
    left = lnew;
  }
  // ---------------------------------------------------------------- core 0 = Q_0 (U sigma), exponents given back (exact)
  const int64_t c1 = p.bcap[1], k0 = p.kq[0], m0 = p.rows_k[0] * p.I[0];
  if (f32) {
    void* ls = ws + p.off_left[lslot];
    TTR_TRY(ttr_scale_batch(dt, k0 * c1, B, left, k0 * c1, nullptr, 0, expo, +1, ls, k0 * c1, st));
    left = ls;
  }
  TTR_TRY(ttr_qr_apply(dt, m0, p.r1[0], B, ws + p.off_qr[0], p.qr_wsb[0], left, c1, k0 * c1, c1, cores_out[0], c1, m0 * c1, st));
  return TTR_OK;
}

}  // extern "C"
