// extern "C" surface of libttround_hip.so + the small streaming kernels (norm, column scaling),
// error reporting and the per-kernel HIP-event profiler used by bench.py.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "ttr_common.h"

namespace ttr {

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return TTR_E_HIP;
}

// ------------------------------------------------------------------ profiler
struct ProfRec {
  int kind;
  hipEvent_t start, stop;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;

ProfScope::ProfScope(int kind_, hipStream_t s) : kind(kind_), stream(s), slot(nullptr) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  r.kind = kind;
  if (!g_prof_pool.empty()) {
    r.start = g_prof_pool.back().first;
    r.stop = g_prof_pool.back().second;
    g_prof_pool.pop_back();
  } else {
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  }
  (void)hipEventRecord(r.start, stream);
  g_prof_recs.push_back(r);
  slot = (void*)(uintptr_t)g_prof_recs.size();  // index + 1
}

ProfScope::~ProfScope() {
  if (!slot) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const size_t idx = (size_t)(uintptr_t)slot - 1;
  if (idx < g_prof_recs.size()) (void)hipEventRecord(g_prof_recs[idx].stop, stream);
}

// ------------------------------------------------------------------ executed-work census
static int g_prof_level = 0;          // 0 off, 1 = times, 2 = times + executed-work census
static double* g_work_dev = nullptr;  // [2 * TTR_PROF_NKINDS]: flops per kind, then bytes per kind

bool work_census_on() { return g_prof_level >= 2 && g_work_dev != nullptr; }

__global__ void work_items_kernel(const int32_t* __restrict__ f1, const int32_t* __restrict__ f2, int64_t batch, double fl0, double fl1,
                                  double fl2, double fl3, double by0, double by1, double by2, double by3, double* __restrict__ out_fl,
                                  double* __restrict__ out_by) {
  __shared__ double red[2 * kThreads / kWave];
  double a = 0.0, c = 0.0;
  for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < batch; b += (int64_t)gridDim.x * kThreads) {
    const int k = ((f1 && f1[b] != 0) ? 1 : 0) + ((f2 && f2[b] != 0) ? 2 : 0);
    a += k == 0 ? fl0 : k == 1 ? fl1 : k == 2 ? fl2 : fl3;
    c += k == 0 ? by0 : k == 1 ? by1 : k == 2 ? by2 : by3;
  }
  a = wave_sum(a); c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[kThreads / kWave + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / kWave; ++w) { a += red[w]; c += red[kThreads / kWave + w]; }
    atomicAdd(out_fl, a);
    atomicAdd(out_by, c);
  }
}

void work_items(int kind, const int32_t* f1, const int32_t* f2, int64_t batch, const double fl[4], const double by[4], hipStream_t s) {
  if (!work_census_on() || batch <= 0) return;
  int64_t gx = ceil_div(batch, kThreads);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(work_items_kernel, dim3((unsigned)gx), dim3(kThreads), 0, s, f1, f2, batch, fl[0], fl[1], fl[2], fl[3], by[0], by[1],
                     by[2], by[3], g_work_dev + kind, g_work_dev + TTR_PROF_NKINDS + kind);
}

template <typename T>
__global__ void work_qr_taus_kernel(const T* __restrict__ tau, int64_t nblk, int64_t nb_per_item, int NP, int64_t m, int64_t rpb, int n,
                                    int kc, int reads_input, double* __restrict__ out_fl, double* __restrict__ out_by) {
  __shared__ double red[2 * kThreads / kWave];
  double a = 0.0, by = 0.0;
  const double s = (double)sizeof(T);
  for (int64_t blk = (int64_t)blockIdx.x * kThreads + threadIdx.x; blk < nblk; blk += (int64_t)gridDim.x * kThreads) {
    const T* __restrict__ t = tau + blk * NP;
    int q = 0;
    for (int pnl = 0; pnl < NP / 16; ++pnl) {
      bool live = false;
      for (int j = 0; j < 16; ++j) live = live || (t[16 * pnl + j] != T(0));
      q += live ? 1 : 0;
    }
    const int64_t b = blk % nb_per_item;
    double r = (double)(m - b * rpb < rpb ? m - b * rpb : rpb);
    if (r < 0) r = 0;
    double c = 16.0 * q;
    if (c > n) c = n;
    if (kc > 0) {
      a += 4.0 * r * c * kc;
      by += s * (r * c + (q > 0 ? r * kc : 0.0));                       // live reflectors read, the block's output rows written
    } else {
      a += 2.0 * r * c * c - 2.0 * c * c * c / 3.0 + 4.0 * r * c * (n - c);
      by += s * (r * c + (q > 0 ? (double)n * n : 0.0) + (reads_input ? r * n : 0.0));   // reflectors + R written (+ the block read)
    }
  }
  a = wave_sum(a); by = wave_sum(by);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[kThreads / kWave + (threadIdx.x >> 6)] = by; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / kWave; ++w) { a += red[w]; by += red[kThreads / kWave + w]; }
    atomicAdd(out_fl, a);
    atomicAdd(out_by, by);
  }
}

void work_qr_taus(int kind, const void* tau, bool f64, int64_t nblk, int64_t nb_per_item, int NP, int64_t m, int64_t rpb, int n,
                  int kc, bool reads_input, hipStream_t s) {
  if (!work_census_on() || nblk <= 0) return;
  int64_t gx = ceil_div(nblk, kThreads);
  if (gx > 512) gx = 512;
  double* fl = g_work_dev + kind;
  double* by = g_work_dev + TTR_PROF_NKINDS + kind;
  if (f64)
    hipLaunchKernelGGL(work_qr_taus_kernel<double>, dim3((unsigned)gx), dim3(kThreads), 0, s, (const double*)tau, nblk, nb_per_item, NP, m,
                       rpb, n, kc, reads_input ? 1 : 0, fl, by);
  else
    hipLaunchKernelGGL(work_qr_taus_kernel<float>, dim3((unsigned)gx), dim3(kThreads), 0, s, (const float*)tau, nblk, nb_per_item, NP, m,
                       rpb, n, kc, reads_input ? 1 : 0, fl, by);
}

// ------------------------------------------------------------------ small kernels
template <typename T>
__global__ __launch_bounds__(kThreads) void norm_kernel(const T* __restrict__ x, int64_t count, int64_t stride_x,
                                                        T* __restrict__ out) {
  __shared__ double red[kThreads / kWave];
  const int64_t b = blockIdx.x;
  const T* __restrict__ xb = x + b * stride_x;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < count; i += kThreads) {
    const double v = (double)xb[i];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int w = 0; w < kThreads / kWave; ++w) s += red[w];
    out[b] = (T)sqrt(s);
  }
}

// Large single vectors: two-stage (partials per workgroup, then a finishing workgroup).
template <typename T>
__global__ __launch_bounds__(kThreads) void sumsq_partial_kernel(const T* __restrict__ x, int64_t count,
                                                                 double* __restrict__ part) {
  __shared__ double red[kThreads / kWave];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (int64_t)gridDim.x * kThreads) {
    const double v = (double)x[i];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int w = 0; w < kThreads / kWave; ++w) s += red[w];
    part[blockIdx.x] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void scale_cols_kernel(int64_t rows, int64_t cols, const T* __restrict__ in,
                                                              int64_t ldi, int64_t stride_in, const T* __restrict__ s,
                                                              int64_t stride_s, int mode, T* __restrict__ out,
                                                              int64_t ldo, int64_t stride_out) {
  const int64_t b = blockIdx.y;
  const int64_t total = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
    const int64_t i = idx / cols, j = idx % cols;
    const T v = in[b * stride_in + i * ldi + j];
    const T sj = s[b * stride_s + j];
    T r;
    if (mode == TTR_SCALE_MUL) r = v * sj;
    else r = (fabs((double)sj) < (double)Num<T>::tiny()) ? T(0) : v / sj;
    out[b * stride_out + i * ldo + j] = r;
  }
}

// x[b][:, j] <- 0 for j >= keep[b] (in place): the device-side truncation of an eps-mode sweep that computes every bond at
// its rank CAP and never reads the selected rank back (keep = the eigensolver's info[b]; 0 = zero guard: everything goes)
template <typename T>
__global__ __launch_bounds__(kThreads) void mask_cols_kernel(int64_t rows, int64_t cols, T* __restrict__ x, int64_t ldx,
                                                             int64_t stride_x, const int32_t* __restrict__ keep) {
  const int64_t b = blockIdx.y;
  const int64_t k = keep[b];
  if (k >= cols) return;
  const int64_t w = cols - k, total = rows * w;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
    const int64_t i = idx / w, j = k + idx % w;
    x[b * stride_x + i * ldx + j] = T(0);
  }
}

// flat[b] = 1 when the `keep` largest singular values of item b (sigma sorted decreasing) lie within a factor 1 / thr of
// each other: sigma[keep - 1] >= thr * sigma[0] > 0
template <typename T>
__global__ void spectrum_flat_kernel(int64_t batch, int n, int keep, T thr, const T* __restrict__ sigma, int64_t stride_sigma,
                                     int use_delta, double delta2, const double* __restrict__ delta2_dev, int32_t* __restrict__ flat,
                                     int noise_c, const int32_t* __restrict__ rows32, int n_full) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  // `rows32` items (the carry of a packed bond: rows 32.. are exactly zero, ttr_rowgram): sigma[32..] are STRUCTURAL zeros -- exact
  // in pass 1 and in pass 2 alike -- so what the rule decides about them is certain and only the 32 computed values carry pass 1's
  // error.  The rule sees them at the noise floor c eps sigma_0 (TTR_KNOB_RANK_NOISE_FLOOR, rank_rule in ttr_common.h; 0 when the
  // floor is off): a fixed, known tail energy nstruct (c eps sigma_0)^2 that enters the sums below without an error bar.
  // (Round 6: with the floor on by default the 32 zeros used to be treated like computed values "within E of delta^2" -- no
  // rows32 item of an fp32 eps-mode sweep took the one-pass shortcut any more, 1.63 -> 1.85 ms for round_tt(eps=1e-4) of one 64^8 train.)
  const int nstruct = (rows32 && rows32[b] != 0 && n > 32) ? n - 32 : 0;
  n -= nstruct;                      // computed values: sigma[0 .. n)
  if (keep > n + nstruct) keep = n + nstruct;
  const T* __restrict__ sgr = sigma + b * stride_sigma;
  const T s0 = sgr[0];
  // (the rank rule's view of the spectrum: rank_rule in ttr_common.h -- with TTR_KNOB_RANK_NOISE_FLOOR nothing lies below c eps sigma_0)
  const T nfl = noise_c > 0 ? T(noise_c) * Num<T>::eps() * s0 : T(0);
  auto sg = [&](int k) { const T v = k < n ? sgr[k] : T(0); return v < nfl ? nfl : v; };   // (k >= n: a structural zero at the floor)
  int kp = keep;
  bool ok = s0 > T(0);
  const double d2 = use_delta ? (delta2_dev ? *delta2_dev : delta2) : 0.0;
  if (ok && d2 > 0.0) {
    // eps mode: the rank comes from the tail energies of pass 1's sigma, which carry an absolute error of up to E = 64 n eps sigma_1^2
    // (n values, each c eps sigma_1^2 off); the item only qualifies when the rule's decision is the same for every spectrum within E
    // of this one -- tail(r) <= delta^2 - E and tail(r - 1) > delta^2 + E at the selected rank r (rank cap binding: only the latter)
    const int nt = n + nstruct;
    const double E = 64.0 * n_full * (double)Num<T>::eps() * (double)s0 * (double)s0;
    double acc = 0.0, tail_r = 0.0;
    int tail = 0;
    for (int k = nt - 1; k >= 0; --k) {
      acc += (double)sg(k) * (double)sg(k);
      if (acc <= d2) { tail = nt - k; tail_r = acc; } else break;
    }
    int r = nt - tail;
    if (r < 1) r = 1;
    if (r > keep) {  // the cap decides as long as the rule cannot cut the keep-th value: tail(keep - 1) > delta^2 + E
      double tc = 0.0;
      for (int k = nt - 1; k >= keep - 1; --k) tc += (double)sg(k) * (double)sg(k);
      // (a keep-th value that is itself structural is known exactly: no error bar)
      ok = keep - 1 >= n ? tc > d2 : tc > d2 + E;
      kp = keep;
    } else {
      const int rr = nt - tail;  // the rule's rank before the ">= 1" clamp
      // what is cut stays cut: nothing computed is cut (only structural zeros: certain), or the computed tail keeps its distance
      const bool cut_safe = tail <= nstruct || tail_r <= d2 - E;
      bool keep_safe = true;                                 // the last kept value cannot be cut as well (rr = 0: rank 1 either way)
      if (rr >= 1) keep_safe = rr - 1 >= n ? (tail_r + (double)sg(rr - 1) * (double)sg(rr - 1) > d2)
                                           : (tail_r + (double)sg(rr - 1) * (double)sg(rr - 1) > d2 + E);
      ok = cut_safe && keep_safe;
      kp = r;
    }
  }
  flat[b] = (ok && sg(kp - 1) >= thr * s0) ? 1 : 0;
}

// Block-wide sum of doubles (256 threads), result in every thread.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();  // `red` may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0;
  for (int w = 0; w < kThreads / kWave; ++w) s += red[w];
  return s;
}

// flag[b] = 1 when rows 32.. of the 64-row matrix R[b] hold at most (c eps)^2 of its squared Frobenius norm -- the packing test of
// the fused push + factor kernel (ttr_qr.hip: `packed`), for the LAST core of a sweep, which no push follows
template <typename T>
__global__ __launch_bounds__(kThreads) void carry_rows32_kernel(int64_t cols, const T* __restrict__ R, int64_t ldr, int64_t strideR,
                                                                double ce2, int32_t* __restrict__ flag) {
  __shared__ double red[kThreads / kWave];
  const int64_t b = blockIdx.x;
  const T* __restrict__ Rb = R + b * strideR;
  double all = 0.0, low = 0.0;
  for (int64_t idx = threadIdx.x; idx < 64 * cols; idx += kThreads) {
    const int64_t i = idx / cols, j = idx - i * cols;
    const double v = (double)Rb[i * ldr + j];
    all += v * v;
    if (i >= 32) low += v * v;
  }
  all = block_sum(all, red);
  low = block_sum(low, red);
  if (threadIdx.x == 0) flag[b] = (ce2 > 0.0 && low <= ce2 * all) ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void pow2_normalize_kernel(const T* __restrict__ x, int64_t count, int64_t stride_x,
                                                                  T* __restrict__ out, int64_t stride_out,
                                                                  int32_t* __restrict__ e_out, int32_t* __restrict__ expo_acc) {
  __shared__ double red[kThreads / kWave];
  const int64_t b = blockIdx.x;
  const T* __restrict__ xb = x + b * stride_x;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < count; i += kThreads) {
    const double v = (double)xb[i];
    acc += v * v;
  }
  const double nrm = sqrt(block_sum(acc, red));
  int e = 0;
  if (nrm > 0.0 && nrm < 1e300) (void)frexp(nrm, &e);
  if (out) {
    T* __restrict__ ob = out + b * stride_out;
    for (int64_t i = threadIdx.x; i < count; i += kThreads) ob[i] = (T)ldexp((double)xb[i], -e);  // exact
  }
  if (threadIdx.x == 0) {
    if (e_out) e_out[b] = e;
    if (expo_acc) expo_acc[b] += e;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void scale_batch_kernel(const T* __restrict__ x, int64_t count, int64_t stride_x,
                                                               const T* __restrict__ scale, int64_t stride_scale,
                                                               const int32_t* __restrict__ expo, int expo_sign,
                                                               T* __restrict__ out, int64_t stride_out) {
  const int64_t b = blockIdx.y;
  const T sc = scale ? scale[b * stride_scale] : T(1);
  const int e = expo ? expo_sign * expo[b] : 0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (int64_t)gridDim.x * kThreads) {
    T v = x[b * stride_x + i] * sc;
    if (e != 0) v = (T)ldexp((double)v, e);
    out[b * stride_out + i] = v;
  }
}

// One workgroup per batch item; see ttr_orth_fixup in the header.
template <typename T>
__global__ __launch_bounds__(kThreads) void orth_fixup_kernel(int r, int64_t n, T* __restrict__ X, int64_t vs, int64_t es,
                                                              int64_t strideX, const T* __restrict__ sigma,
                                                              int64_t stride_sigma, double dead_rel,
                                                              const int32_t* __restrict__ rank_dev) {
  __shared__ double red[kThreads / kWave];
  const int64_t b = blockIdx.x;
  if (rank_dev) r = rank_dev[b] < r ? rank_dev[b] : r;  // vectors beyond the selected rank are cut away by the caller later
  const T* __restrict__ sg = sigma + b * stride_sigma;
  const double s0 = (double)sg[0];
  int first = r;
  for (int i = 0; i < r; ++i)
    if (!((double)sg[i] > dead_rel * s0)) { first = i; break; }  // (also catches NaN / zero sigma_0)
  if (first >= r) return;
  T* __restrict__ Xb = X + b * strideX;
  const int tid = threadIdx.x;
  for (int i = first; i < r; ++i) {
    T* __restrict__ xi = Xb + (int64_t)i * vs;
    for (int attempt = 0; attempt < 3; ++attempt) {
      double n0 = 0.0;
      for (int64_t k = tid; k < n; k += kThreads) { const double v = (double)xi[k * es]; n0 += v * v; }
      n0 = block_sum(n0, red);
      bool regenerate = !(n0 > 0.0) || !(n0 < 1e300);
      if (!regenerate) {
        for (int pass = 0; pass < 2; ++pass)
          for (int j = 0; j < i; ++j) {  // modified Gram-Schmidt against every finished vector
            const T* __restrict__ xj = Xb + (int64_t)j * vs;
            double d = 0.0;
            for (int64_t k = tid; k < n; k += kThreads) d += (double)xj[k * es] * (double)xi[k * es];
            d = block_sum(d, red);
            for (int64_t k = tid; k < n; k += kThreads) xi[k * es] = (T)((double)xi[k * es] - d * (double)xj[k * es]);
          }
        double n1 = 0.0;
        for (int64_t k = tid; k < n; k += kThreads) { const double v = (double)xi[k * es]; n1 += v * v; }
        n1 = block_sum(n1, red);
        if (n1 > 1e-6 * n0) {  // a genuine remainder: normalise and go on
          const double inv = 1.0 / sqrt(n1);
          for (int64_t k = tid; k < n; k += kThreads) xi[k * es] = (T)((double)xi[k * es] * inv);
          break;
        }
        regenerate = true;  // the vector lay in the span of the previous ones
      }
      if (regenerate) {  // hashed pseudo-random replacement (deterministic), orthogonalised by the next attempt
        for (int64_t k = tid; k < n; k += kThreads) {
          uint32_t h = (uint32_t)(k * 2654435761u) ^ (uint32_t)((i + 1) * 40503u) ^ (uint32_t)((attempt + 1) * 97u);
          h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
          xi[k * es] = (T)((double)(h >> 8) * (1.0 / 8388608.0) - 1.0);
        }
        __syncthreads();
      }
    }
  }
}

// Block variant of the kernel above for r <= 64 vectors (every bond of the rounding sweeps): the sequential modified
// Gram-Schmidt of the kernel above walks the whole vector two times per (dead vector, earlier vector) pair with a block
// reduction each -- 51 ms per metric step on a decaying-spectrum batch whose kept directions 17 .. 31 of every bond lie below
// the resolution (SURVEY 8d's second input variant, measured).  Here one ROUND is: the Gram matrix S = X X^T of all r vectors
// (one pass, column chunks through LDS, products and sums in double), Gram-Schmidt of the dead
// rows IN COEFFICIENT SPACE against everything before them (r x r matrices in LDS, double; the live rows are orthonormal
// already and stay untouched), then X_dead <- W X in a second pass.  A dead vector whose remainder collapses (below 1 % of its
// norm, or a zero / non-finite vector) is replaced by a hashed pseudo-random one, which the next round orthogonalises.  A second
// round when a remainder lost more than half of its squared norm ("twice is enough": the second sees a Gram matrix within
// rounding of the identity), a third / fourth only after a replacement.  Same semantics as the kernel above: genuine remainders are kept, live vectors are not touched.
constexpr int kOfMaxE = 32;        // tile elements per thread: r x CW <= 32 x 256 (CW = 256 up to 32 vectors, 128 above)

inline size_t orth_fixup_lds_bytes(int r, int cw, size_t es) {
  const int r4 = (r + 15) & ~15;   // whole 16 x 16 MFMA tiles
  return (size_t)r4 * (cw + 4) * es + 2 * (size_t)r4 * (r4 + 1) * 8 + 2 * 64 * 8 + 64 * 4 + 16;
}

// kOfCW: columns per chunk (a chunk = one global round trip + two barriers: 64 columns left the kernel latency-bound on
// them); kOfLd: tile row stride
// V2 (round 5, the CW = 256 instance, i.e. up to 32 vectors -- every bond of a rank-32 rounding): the same two passes per round with
// their inner loops rebuilt around the LDS.  Round 4's loops issued one ds_read_b32 per MFMA operand and waited for it (184 VGPRs:
// two waves per SIMD, nothing to hide the latency with): 12 us per 32 x 256 chunk and pass, measured (profiles/r05_decay_probe.txt:
// 0.76 ms per launch and round at B = 2048) against ~1 us of MFMA time.  Here (a) the Gram pass reads its operands as ds_read_b128
// with the K index permuted (lane (i, q) takes columns 16 g + 4 q + j for the j-th MFMA of column group g: any K order is a valid
// sum), computes only the tile rows that hold dead vectors, and the four waves split the chunk's 16-column groups (partials added
// through S in wave order: deterministic); (b) the apply pass keeps its W operands in registers for the whole pass and reads the
// tile with a K permutation that spreads a wave's four K rows over all 64 banks.
template <typename T, int kOfCW, bool V2>
__global__ __launch_bounds__(kThreads, V2 ? 2 : 1) void orth_fixup_block_kernel(int r, int64_t n, T* __restrict__ X, int64_t vs, int64_t es,
                                                                    int64_t strideX, const T* __restrict__ sigma,
                                                                    int64_t stride_sigma, double dead_rel,
                                                                    const int32_t* __restrict__ rank_dev, int max_rounds,
                                                                    double* __restrict__ census, long long* __restrict__ dbg,
                                                                    const int32_t* __restrict__ skip_items, int round0) {
  constexpr int kOfLd = kOfCW + 4;
  // (`skip_items` / `round0`: this launch finishes what the three-launch rounds left over -- items flagged done return at once, the
  // others continue with round number round0, which only enters the hashed replacement vectors)
  if (skip_items && skip_items[blockIdx.x] != 0) return;
  // (diagnostics, ttr_debug_set_qr_stamps with TTR_KNOB_QR_STAMP_BX = -1: item 0 stamps its phases -- start, then per round: Gram
  // pass done, coefficients done, apply pass done)
  int dbgi = 0;
  auto ostamp = [&]() { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[dbgi++] = (long long)clock64(); };
  ostamp();
  extern __shared__ __attribute__((aligned(16))) unsigned char of_smem[];
  const int r_launch = r;
  const int64_t b = blockIdx.x;
  if (rank_dev) r = rank_dev[b] < r ? rank_dev[b] : r;
  const T* __restrict__ sg = sigma + b * stride_sigma;
  const double s0 = (double)sg[0];
  int first = r;
  for (int i = 0; i < r; ++i)
    if (!((double)sg[i] > dead_rel * s0)) { first = i; break; }  // (also catches NaN / zero sigma_0)
  if (first >= r) return;
  // carve the dynamic LDS (sized for the launch's r, rounded up to whole 16 x 16 MFMA tiles)
  const int r4 = (r_launch + 15) & ~15;
  const int ls = r4 + 1;                                   // row stride of S / W
  double* S = reinterpret_cast<double*>(of_smem);          // [r4][ls]
  double* W = S + (size_t)r4 * ls;
  double* tv = W + (size_t)r4 * ls;                        // [64]
  double* pj = tv + 64;
  int* regen = reinterpret_cast<int*>(pj + 64);            // [64]
  int* any_regen = regen + 64;
  T* tile = reinterpret_cast<T*>(any_regen + 4);           // [r4][kOfLd]
  T* __restrict__ Xb = X + b * strideX;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int nt = (r + 15) >> 4;              // 16-row tiles that hold vectors
  const bool vec_contig = es == 1;           // vectors are rows of a row-major matrix (else: columns, es = row stride)
  const int ne = (r * kOfCW + kThreads - 1) / kThreads;    // tile elements per thread (<= kOfMaxE)
  // rows of the tile beyond r are read by the 16 x 16 MFMA tiles: keep them zero
  for (int idx = r * kOfLd + tid; idx < r4 * kOfLd; idx += kThreads) tile[idx] = T(0);

  // chunk c0 .. c0 + 63 of all r vectors: global -> registers (issued one chunk ahead), registers -> LDS
  auto fetch = [&](int64_t c0, T* reg) {
    const int cw = (int)((n - c0) < kOfCW ? (n - c0) : kOfCW);
#pragma unroll
    for (int e = 0; e < kOfMaxE; ++e) {
      if (e >= ne) break;
      const int idx = tid + e * kThreads;
      int i, c;
      if (vec_contig) { i = idx / kOfCW; c = idx - i * kOfCW; } else { c = idx / r; i = idx - c * r; }
      reg[e] = (i < r && c < cw) ? Xb[(int64_t)i * vs + (c0 + c) * es] : T(0);
    }
  };
  auto stage = [&](const T* reg) {
#pragma unroll
    for (int e = 0; e < kOfMaxE; ++e) {
      if (e >= ne) break;
      const int idx = tid + e * kThreads;
      int i, c;
      if (vec_contig) { i = idx / kOfCW; c = idx - i * kOfCW; } else { c = idx / r; i = idx - c * r; }
      if (i < r && c < kOfCW) {
        const T v = reg[e];
        tile[i * kOfLd + c] = (v - v == T(0)) ? v : T(0);   // (non-finite entries of a dead vector count as zero: 0 x NaN would poison the products)
      }
    }
  };

  // Chunk order.  Every item's vectors lie 4 n bytes apart (8 KB at the metric's bonds), so the 32 row pieces of chunk c of EVERY
  // item share their address bits 10 .. 12: workgroups that walk their chunks in step keep hitting the same eighth of the HBM
  // channels (measured, round 5: 13.5 us per 32 KB chunk and workgroup = 1.2 TB/s chip-wide with the loads of a whole chunk in
  // flight per workgroup; profiles/r05_orth_stamps.txt).  V2: item b starts at chunk b mod nch and wraps around -- at any moment
  // the resident workgroups cover all chunk phases.  (The Gram sums are then added in an item-dependent order: double
  // accumulation across chunks, so an item's result depends on its position in the batch at the 1e-16 level of S only.)
  const int nch = (int)((n + kOfCW - 1) / kOfCW);
  const int rot = V2 ? (int)(b % nch) : 0;
  auto chunk_c0 = [&](int tq) { int c = tq + rot; if (c >= nch) c -= nch; return (int64_t)c * kOfCW; };
  if (census && round0 == 0 && tid == 0) atomicAdd(census + TTR_PROF_NKINDS + TTR_PROF_MISC, 1.0);   // census: items with dead rows ...
  if (round0 > 0) census = nullptr;   // (the item and its first rounds were counted by the three-launch rounds)
  for (int round = round0; round < max_rounds; ++round) {
    if (census && tid == 0) atomicAdd(census + TTR_PROF_MISC, 1.0);                   // ... and the rounds they took
    // ---- S = X X^T on the matrix cores: wave w owns the 16-row tile w of S (all column tiles); fp32 accumulators are
    // flushed into double sums after every chunk (64 products per entry), fp64 accumulates in place
    double sacc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) sacc[u][v] = 0.0;
    T reg[kOfMaxE];
    fetch(chunk_c0(0), reg);
    for (int tq = 0; tq < nch; ++tq) {
      __syncthreads();
      stage(reg);
      __syncthreads();
      if (tq + 1 < nch) fetch(chunk_c0(tq + 1), reg);   // the next chunk's loads fly under this chunk's products
      if constexpr (V2) {
        // nt <= 2 tile rows; sacc[2 tr + v] = tile (tr, v) of S for the tile rows tr >= tr0 that hold dead vectors
        typename Mfma<T>::Acc acc[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = Mfma<T>::zero();
        const int tr0 = first >> 4;
        const T* __restrict__ xr = tile + (lane & 15) * kOfLd + 4 * (lane >> 4);
#pragma unroll
        for (int g = 0; g < kOfCW / 64; ++g) {
          const int k0 = 16 * (wv + 4 * g);
          typedef T tv4 __attribute__((ext_vector_type(4)));
          const tv4 x0 = *reinterpret_cast<const tv4*>(xr + k0);
          const tv4 x1 = nt > 1 ? *reinterpret_cast<const tv4*>(xr + 16 * kOfLd + k0) : tv4{0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (tr0 == 0) {
              acc[0] = Mfma<T>::mma(x0[j], x0[j], acc[0]);
              if (nt > 1) acc[1] = Mfma<T>::mma(x0[j], x1[j], acc[1]);
            }
            if (nt > 1) {
              acc[2] = Mfma<T>::mma(x1[j], x0[j], acc[2]);
              acc[3] = Mfma<T>::mma(x1[j], x1[j], acc[3]);
            }
          }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int u = 0; u < 4; ++u) sacc[v][u] += (double)acc[v][u];
      } else {
      if (wv < nt) {
        typename Mfma<T>::Acc acc[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = Mfma<T>::zero();
        const T* __restrict__ arow = tile + (16 * wv + (lane & 15)) * kOfLd + (lane >> 4);
#pragma unroll 4
        for (int k0 = 0; k0 < kOfCW; k0 += 4) {
          const T a = arow[k0];
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (v < nt) acc[v] = Mfma<T>::mma(a, tile[(16 * v + (lane & 15)) * kOfLd + k0 + (lane >> 4)], acc[v]);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int u = 0; u < 4; ++u) sacc[v][u] += (double)acc[v][u];
      }
      }
    }
    __syncthreads();
    if constexpr (V2) {
      // the four waves' partial sums, added in wave order (deterministic); tile (tr, v) lives in sacc[2 tr + v]
      for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
          for (int tv = 0; tv < 4; ++tv) {
            const int tr = tv >> 1, v = tv & 1;
            if (tr < (first >> 4) || tr >= nt || v >= nt) continue;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              double* dst = S + (16 * tr + Mfma<T>::row(lane, u)) * ls + 16 * v + (lane & 15);
              *dst = (w == 0 ? 0.0 : *dst) + sacc[tv][u];
            }
          }
        }
        __syncthreads();
      }
    } else {
    if (wv < nt) {
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (v < nt) {
#pragma unroll
          for (int u = 0; u < 4; ++u) S[(16 * wv + Mfma<T>::row(lane, u)) * ls + 16 * v + (lane & 15)] = sacc[v][u];
        }
    }
    }
    if (tid < 64) regen[tid] = 0;
    if (tid == 0) { any_regen[0] = 0; any_regen[1] = 0; }
    __syncthreads();
    ostamp();
    // ---- the dead rows against everything before them, in coefficient space (double).  The live rows are orthonormal
    // (S_LL = I to rounding), so the remainders x_d - S_dL x_L have the Gram matrix C = S_DD - S_DL S_LD; with C = L L^T
    // (Cholesky) the rows of  L^-1 [-S_DL, I]  are the coefficients of the orthonormalised dead vectors.  A pivot below 1e-4
    // of the vector's own squared norm (or a zero vector) = the remainder collapsed: that row is replaced.
    const int nd = r - first;
    double* Cm = W;                                      // nd x nd, stride ls
    for (int idx = tid; idx < nd * nd; idx += kThreads) {
      const int ia = idx / nd, ib = idx - ia * nd;
      const double* __restrict__ sa = S + (size_t)(first + ia) * ls;
      const double* __restrict__ sb = S + (size_t)(first + ib) * ls;
      double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
      int k = 0;
      for (; k + 3 < first; k += 4) { c0 += sa[k] * sb[k]; c1 += sa[k + 1] * sb[k + 1]; c2 += sa[k + 2] * sb[k + 2]; c3 += sa[k + 3] * sb[k + 3]; }
      for (; k < first; ++k) c0 += sa[k] * sb[k];
      Cm[ia * ls + ib] = sa[first + ib] - ((c0 + c1) + (c2 + c3));
    }
    __syncthreads();
    for (int a = 0; a < nd; ++a) {                       // right-looking Cholesky, lower triangle in place
      const double piv = Cm[a * ls + a], saa = S[(first + a) * ls + first + a];
      const bool bad = !(saa > 0.0) || !(saa < 1e300) || !(piv > 1e-4 * saa);
      const double dinv = bad ? 0.0 : 1.0 / sqrt(piv);
      __syncthreads();                                   // (everybody has read the pivot)
      if (tid > a && tid < nd) Cm[tid * ls + a] *= dinv; // (collapsed: the column is removed)
      if (tid == a) Cm[a * ls + a] = bad ? 1.0 : piv * dinv;
      if (tid == 0 && bad) { regen[first + a] = 1; any_regen[0] = 1; }
      if (tid == 0 && !(piv > 0.5 * saa)) any_regen[1] = 1;   // lost more than half of its squared norm: orthogonalise twice
      __syncthreads();
      if (!bad) {
        const int rem = nd - a - 1;
        for (int idx = tid; idx < rem * rem; idx += kThreads) {
          const int i = a + 1 + idx / rem, j = a + 1 + idx % rem;
          if (j <= i) Cm[i * ls + j] -= Cm[i * ls + a] * Cm[j * ls + a];
        }
      }
      __syncthreads();
    }
    // Z = [-S_DL, I] in place over the dead rows of S, then the forward substitution L W_D = Z row by row (S is not needed any more)
    for (int idx = tid; idx < nd * r4; idx += kThreads) {
      const int ia = idx / r4, k = idx - ia * r4;
      double* __restrict__ zr = S + (size_t)(first + ia) * ls;
      zr[k] = k < first ? -zr[k] : (k == first + ia ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int a = 0; a < nd; ++a) {
      if (tid < r4) {
        double* __restrict__ za = S + (size_t)(first + a) * ls;
        double w = 0.0;
        if (!regen[first + a]) {
          double w0 = za[tid], w1 = 0.0, w2 = 0.0, w3 = 0.0;
          const double* __restrict__ la = Cm + (size_t)a * ls;
          int bq = 0;
          for (; bq + 3 < a; bq += 4) {
            w0 -= la[bq] * S[(size_t)(first + bq) * ls + tid];
            w1 -= la[bq + 1] * S[(size_t)(first + bq + 1) * ls + tid];
            w2 -= la[bq + 2] * S[(size_t)(first + bq + 2) * ls + tid];
            w3 -= la[bq + 3] * S[(size_t)(first + bq + 3) * ls + tid];
          }
          for (; bq < a; ++bq) w0 -= la[bq] * S[(size_t)(first + bq) * ls + tid];
          w = ((w0 + w1) + (w2 + w3)) / la[a];
        }
        za[tid] = w;
      }
      __syncthreads();
    }
    ostamp();
    // ---- X_dead <- W X  (regenerated rows: hashed pseudo-random values, orthogonalised by the next round)
    fetch(chunk_c0(0), reg);
    for (int tq = 0; tq < nch; ++tq) {
      const int64_t c0 = chunk_c0(tq);
      const int cw = (int)((n - c0) < kOfCW ? (n - c0) : kOfCW);
      __syncthreads();
      stage(reg);
      __syncthreads();
      if (tq + 1 < nch) fetch(chunk_c0(tq + 1), reg);
      // wave w: the 16-column blocks w, w + 4, .. of the chunk, every 16-row tile that holds dead rows; W (double in LDS) is the A
      // operand in the matrix precision -- the second round sees W = I + O(first round's error), which restores full accuracy
      for (int cb = wv; cb < kOfCW / 16; cb += 4)
      for (int it = first >> 4; it < nt; ++it) {
        typename Mfma<T>::Acc acc = Mfma<T>::zero();
        const double* __restrict__ wrow = S + (size_t)(16 * it + (lane & 15)) * ls + (lane >> 4);   // (the dead rows of S hold W_D now)
        const T* __restrict__ bcol = tile + (lane >> 4) * kOfLd + 16 * cb + (lane & 15);
        for (int k0 = 0; k0 < 16 * nt; k0 += 4) acc = Mfma<T>::mma((T)wrow[k0], bcol[k0 * kOfLd], acc);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int d = 16 * it + Mfma<T>::row(lane, u), c = 16 * cb + (lane & 15);
          if (d < first || d >= r || c >= cw) continue;
          T out = acc[u];
          if (regen[d]) {
            const int64_t k = c0 + c;
            uint32_t h = (uint32_t)(k * 2654435761u) ^ (uint32_t)((d + 1) * 40503u) ^ (uint32_t)((round + 1) * 97u);
            h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
            out = (T)((double)(h >> 8) * (1.0 / 8388608.0) - 1.0);
          }
          Xb[(int64_t)d * vs + (c0 + c) * es] = out;
        }
      }
    }
    __syncthreads();
    // "twice is enough" (Kahan / Parlett): a remainder that kept at least 1 / sqrt(2) of its vector's norm is orthogonal to
    // the others to ~1.4 eps already -- the second round is only run when some vector lost more (or was replaced).  (Measured,
    // round 4: on the decaying-spectrum batch every item needs it -- the dead rows are mostly leakage of the live ones.  Starting
    // the dead rows from hashed vectors instead, one round: 11.3 -> 6.7 ms per step, but the approximation error of that batch
    // rose from 8.6e-6 to 2.4e-5 -- the dead rows do carry part of the tail; not taken.)
    ostamp();
    if (!any_regen[0] && (round >= 1 || !any_regen[1])) break;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// ttr_orth_fixup for LARGE batches (round 5): the same rounds -- Gram matrix, coefficients, X_dead <- W X -- as three launches per
// round instead of one workgroup per item.  Why: the single-workgroup kernel above keeps one 32 KB chunk in flight per workgroup
// and walks an item's 256 KB four times; even with the staggered chunk order a pass costs 8 us per chunk (cycle stamps,
// profiles/r05_orth_stamps.txt), i.e. ~2 TB/s chip-wide.  The Gram pass IS ttr_rowgram on the r x n matrix of vectors (split-K
// partials, 16-byte loads, two slabs in flight per wave: 4 TB/s class), and the apply pass is a streaming kernel of the same
// build (below).  Per-item control flow lives in flag arrays: skip[round][b] != 0 = item b takes no part in that round.
// Small batches keep the single launch (three launches per round and bond would add ~40 dependent launches to a B = 1 call).
constexpr int kOrthSplitRounds = 2;
struct OrthSplitWs {
  int64_t off_g, off_w, off_regen, off_skip, total;
  int parts;    // Gram partials of a ttr_rowgram launch
  int nsplit;   // column splits of the apply launch (grid x)
  int fparts;   // Gram partials written by a FUSED apply launch (one per wave and column split)
};
static OrthSplitWs orth_split_layout(int64_t r, int64_t n, int64_t batch, int64_t es, int max_rounds) {
  OrthSplitWs w{};
  const int64_t r4 = (r + 15) & ~15LL;
  w.parts = n >= 2048 ? 4 : (n >= 1024 ? 2 : 1);   // short fp32 accumulation chains: the partials are summed in double
  w.nsplit = (int)ceil_div(2048, batch);           // aim at >= 2048 workgroups, >= 8 slabs per wave
  const int64_t slabs = (n + 15) / 16;
  if (w.nsplit > slabs / 32) w.nsplit = (int)(slabs / 32);
  if (w.nsplit < 1) w.nsplit = 1;
  w.fparts = 4 * w.nsplit;
  int64_t off = 0;
  w.off_g = off; off += align_up(batch * (w.parts > w.fparts ? w.parts : w.fparts) * r * r * es, 256);
  w.off_w = off; off += align_up(batch * r4 * r4 * es, 256);
  w.off_regen = off; off += align_up(batch * 8, 256);                       // one 64-bit mask per item
  w.off_skip = off; off += align_up((int64_t)(max_rounds + 1) * batch * 4, 256);
  w.total = off;
  return w;
}

template <typename T>
__global__ void orth_init_kernel(int r, int64_t batch, const T* __restrict__ sigma, int64_t stride_sigma, double dead_rel,
                                 const int32_t* __restrict__ rank_dev, int32_t* __restrict__ skip0) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  if (rank_dev) r = rank_dev[b] < r ? rank_dev[b] : r;
  const T* __restrict__ sg = sigma + b * stride_sigma;
  const double s0 = (double)sg[0];
  int first = r;
  for (int i = 0; i < r; ++i)
    if (!((double)sg[i] > dead_rel * s0)) { first = i; break; }
  skip0[b] = first >= r ? 1 : 0;
}

// Coefficients of one round (the middle section of orth_fixup_block_kernel, verbatim): S = sum of the Gram partials (double),
// Cholesky of the dead rows' Schur complement, W_D = L^-1 [-S_DL, I].  Writes W (matrix precision), the regeneration mask and
// the next round's skip flag.
template <typename T>
__global__ __launch_bounds__(kThreads) void orth_coef_kernel(int r, const T* __restrict__ Gp, int parts, T* __restrict__ Wg,
                                                             unsigned long long* __restrict__ regen_mask, const int32_t* __restrict__ skip_now,
                                                             int32_t* __restrict__ skip_next, const T* __restrict__ sigma,
                                                             int64_t stride_sigma, double dead_rel, const int32_t* __restrict__ rank_dev,
                                                             int round, double* __restrict__ census) {
  extern __shared__ __attribute__((aligned(16))) unsigned char oc_smem[];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  if (skip_now[b] != 0) { if (tid == 0) skip_next[b] = 1; return; }
  const int r_launch = r;
  if (rank_dev) r = rank_dev[b] < r ? rank_dev[b] : r;
  const T* __restrict__ sg = sigma + b * stride_sigma;
  const double s0 = (double)sg[0];
  int first = r;
  for (int i = 0; i < r; ++i)
    if (!((double)sg[i] > dead_rel * s0)) { first = i; break; }
  const int r4 = (r_launch + 15) & ~15;
  const int ls = r4 + 1;
  double* S = reinterpret_cast<double*>(oc_smem);   // [r4][ls]
  double* W = S + (size_t)r4 * ls;
  int* regen = reinterpret_cast<int*>(W + (size_t)r4 * ls);   // [64]
  int* any_regen = regen + 64;
  if (census && tid == 0) {
    if (round == 0) atomicAdd(census + TTR_PROF_NKINDS + TTR_PROF_MISC, 1.0);
    atomicAdd(census + TTR_PROF_MISC, 1.0);
  }
  const T* __restrict__ G = Gp + b * (int64_t)parts * r_launch * r_launch;
  for (int idx = tid; idx < r4 * r4; idx += kThreads) {
    const int i = idx / r4, j = idx - i * r4;
    double v = 0.0;
    if (i < r_launch && j < r_launch)
      for (int pt = 0; pt < parts; ++pt) v += (double)G[(int64_t)pt * r_launch * r_launch + i * r_launch + j];
    S[i * ls + j] = v;
  }
  if (tid < 64) regen[tid] = 0;
  if (tid == 0) { any_regen[0] = 0; any_regen[1] = 0; }
  __syncthreads();
  // a dead vector with a non-finite entry has a non-finite diagonal: it counts as the zero vector (replaced below); its row and
  // column must not poison the others (the single-launch kernel zeroes such entries when it stages the vectors)
  for (int d = first; d < r; ++d) {
    const double sdd = S[d * ls + d];
    if (!(sdd - sdd == 0.0)) {
      __syncthreads();
      for (int k = tid; k < r4; k += kThreads) { S[d * ls + k] = 0.0; S[k * ls + d] = 0.0; }
      __syncthreads();
    }
  }
  for (int idx = tid; idx < (r - first) * r4; idx += kThreads) {   // (other non-finite entries of dead rows: treated as zero)
    const int d = first + idx / r4, k = idx % r4;
    const double v = S[d * ls + k];
    if (!(v - v == 0.0)) { S[d * ls + k] = 0.0; S[k * ls + d] = 0.0; }
  }
  __syncthreads();
  const int nd = r - first;
  double* Cm = W;
  for (int idx = tid; idx < nd * nd; idx += kThreads) {
    const int ia = idx / nd, ib = idx - ia * nd;
    const double* __restrict__ sa = S + (size_t)(first + ia) * ls;
    const double* __restrict__ sb = S + (size_t)(first + ib) * ls;
    double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
    int k = 0;
    for (; k + 3 < first; k += 4) { c0 += sa[k] * sb[k]; c1 += sa[k + 1] * sb[k + 1]; c2 += sa[k + 2] * sb[k + 2]; c3 += sa[k + 3] * sb[k + 3]; }
    for (; k < first; ++k) c0 += sa[k] * sb[k];
    Cm[ia * ls + ib] = sa[first + ib] - ((c0 + c1) + (c2 + c3));
  }
  __syncthreads();
  for (int a = 0; a < nd; ++a) {
    const double piv = Cm[a * ls + a], saa = S[(first + a) * ls + first + a];
    const bool bad = !(saa > 0.0) || !(saa < 1e300) || !(piv > 1e-4 * saa);
    const double dinv = bad ? 0.0 : 1.0 / sqrt(piv);
    __syncthreads();
    if (tid > a && tid < nd) Cm[tid * ls + a] *= dinv;
    if (tid == a) Cm[a * ls + a] = bad ? 1.0 : piv * dinv;
    if (tid == 0 && bad) { regen[first + a] = 1; any_regen[0] = 1; }
    if (tid == 0 && !(piv > 0.5 * saa)) any_regen[1] = 1;
    __syncthreads();
    if (!bad) {
      const int rem = nd - a - 1;
      for (int idx = tid; idx < rem * rem; idx += kThreads) {
        const int i = a + 1 + idx / rem, j = a + 1 + idx % rem;
        if (j <= i) Cm[i * ls + j] -= Cm[i * ls + a] * Cm[j * ls + a];
      }
    }
    __syncthreads();
  }
  for (int idx = tid; idx < nd * r4; idx += kThreads) {
    const int ia = idx / r4, k = idx - ia * r4;
    double* __restrict__ zr = S + (size_t)(first + ia) * ls;
    zr[k] = k < first ? -zr[k] : (k == first + ia ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int a = 0; a < nd; ++a) {
    if (tid < r4) {
      double* __restrict__ za = S + (size_t)(first + a) * ls;
      double w = 0.0;
      if (!regen[first + a]) {
        double w0 = za[tid], w1 = 0.0, w2 = 0.0, w3 = 0.0;
        const double* __restrict__ la = Cm + (size_t)a * ls;
        int bq = 0;
        for (; bq + 3 < a; bq += 4) {
          w0 -= la[bq] * S[(size_t)(first + bq) * ls + tid];
          w1 -= la[bq + 1] * S[(size_t)(first + bq + 1) * ls + tid];
          w2 -= la[bq + 2] * S[(size_t)(first + bq + 2) * ls + tid];
          w3 -= la[bq + 3] * S[(size_t)(first + bq + 3) * ls + tid];
        }
        for (; bq < a; ++bq) w0 -= la[bq] * S[(size_t)(first + bq) * ls + tid];
        w = ((w0 + w1) + (w2 + w3)) / la[a];
      }
      za[tid] = w;
    }
    __syncthreads();
  }
  // W_D in the matrix precision (rows below `first` are never applied: zero), mask, next round's flag
  T* __restrict__ Wb = Wg + b * (int64_t)r4 * r4;
  for (int idx = tid; idx < r4 * r4; idx += kThreads) {
    const int i = idx / r4, k = idx - i * r4;
    Wb[idx] = (i >= first && i < r) ? (T)S[(size_t)i * ls + k] : T(0);
  }
  if (tid == 0) {
    unsigned long long m = 0ull;
    for (int d = first; d < r; ++d) if (regen[d]) m |= 1ull << d;
    regen_mask[b] = m;
    skip_next[b] = (any_regen[0] || (round == 0 && any_regen[1])) ? 0 : 1;
  }
}

// X_dead <- W X, streamed: grid (column splits, items); a wave walks 16-column slabs (every item starts at another slab: the
// columns are independent), the vectors of a slab are the B operand straight from global memory (four 64-byte row segments per
// load, the next slab's loads in flight under this slab's products), W is the A operand from an LDS image.
//
// GRAM (at most 32 vectors): the launch also leaves the Gram matrix of the vectors AS IT WROTE THEM -- what the next round's
// coefficients are computed from -- so that the next round does not read the item again for it (ttr_rowgram on 32 x 2048 items:
// 0.35 ms per launch at B = 4096, a quarter of the round).  The MFMA products of the apply pass have the columns on the lanes'
// N index; the Gram product contracts over the columns, so a slab's 32 x 16 tile (live rows as loaded, dead rows as computed)
// goes through a per-wave LDS tile and comes back with the columns on K.  Only the dead row tiles' rows of the Gram matrix are
// formed (orth_coef_kernel reads nothing else); every wave writes its own partial: Gnext[b][4 split + wave][r][r], the others
// zero.
constexpr int kOrthXtLd = 17;
template <typename T, bool GRAM>
__global__ __launch_bounds__(kThreads) void orth_apply_kernel(int r, int64_t n, T* __restrict__ X, int64_t vs, int64_t strideX,
                                                              const T* __restrict__ Wg, const unsigned long long* __restrict__ regen_mask,
                                                              const int32_t* __restrict__ skip_now, const T* __restrict__ sigma,
                                                              int64_t stride_sigma, double dead_rel, const int32_t* __restrict__ rank_dev,
                                                              int round, int nsplit, T* __restrict__ Gnext) {
  using M = Mfma<T>;
  __shared__ T Wl[64 * 65];
  __shared__ T Xt[GRAM ? 4 * 32 * kOrthXtLd : 1];
  const int64_t b = blockIdx.y;
  if (skip_now[b] != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 15, g = lane >> 4;
  const int r_launch = r;
  if (rank_dev) r = rank_dev[b] < r ? rank_dev[b] : r;
  const T* __restrict__ sg = sigma + b * stride_sigma;
  const double s0 = (double)sg[0];
  int first = r;
  for (int i = 0; i < r; ++i)
    if (!((double)sg[i] > dead_rel * s0)) { first = i; break; }
  const int r4 = (r_launch + 15) & ~15;
  const int nt = (r + 15) >> 4, it0 = first >> 4;
  const T* __restrict__ Wb = Wg + b * (int64_t)r4 * r4;
  for (int idx = tid; idx < r4 * r4; idx += kThreads) Wl[(idx / r4) * 65 + idx % r4] = Wb[idx];
  const unsigned long long rm = regen_mask[b];
  __syncthreads();
  T* __restrict__ Xb = X + b * strideX;
  const int64_t slabs = (n + 15) / 16, per = (slabs + nsplit - 1) / nsplit;
  const int64_t cb = (int64_t)blockIdx.x * per, ce = cb + per < slabs ? cb + per : slabs;
  const int64_t nsteps = ce > cb + wave ? (ce - cb - wave + 3) / 4 : 0;
  const int64_t rot = nsteps > 1 ? (int64_t)(b % nsteps) : 0;
  auto step_c = [&](int64_t sidx) { int64_t sq = sidx + rot; if (sq >= nsteps) sq -= nsteps; return cb + wave + 4 * sq; };
  const int nks = r4 >> 2;
  constexpr int KS = GRAM ? 8 : 16;   // k-steps of four vectors held per slab (GRAM: at most 32 vectors)
  auto load_cols = [&](int64_t c, T (&a)[KS]) {
    const int64_t col = c * 16 + cl;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + g;
      T v = (ks < nks && k < r_launch && col < n) ? Xb[(int64_t)k * vs + col] : T(0);
      a[ks] = (v - v == T(0)) ? v : T(0);   // (non-finite entries of a dead vector count as zero)
    }
  };
  T a[KS], an[KS];
  T* const xt = Xt + (GRAM ? wave * 32 * kOrthXtLd : 0);
  typename M::Acc Gacc[2][2] = {{M::zero(), M::zero()}, {M::zero(), M::zero()}};
  if (nsteps > 0) load_cols(step_c(0), a);
  for (int64_t sidx = 0; sidx < nsteps; ++sidx) {
    const int64_t c = step_c(sidx);
    if (sidx + 1 < nsteps) load_cols(step_c(sidx + 1), an);
    if constexpr (GRAM) {   // the slab as loaded (columns beyond n: zeros); the dead rows are overwritten below
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (ks < nks) xt[(4 * ks + g) * kOrthXtLd + cl] = a[ks];
    }
    for (int it = it0; it < nt; ++it) {
      typename M::Acc acc = M::zero();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (ks < nks) acc = M::mma(Wl[(16 * it + cl) * 65 + 4 * ks + g], a[ks], acc);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int d = 16 * it + M::row(lane, u);
        const int64_t col = c * 16 + cl;
        if (d < first || d >= r || col >= n) continue;
        T out = acc[u];
        if ((rm >> d) & 1ull) {
          uint32_t h = (uint32_t)(col * 2654435761u) ^ (uint32_t)((d + 1) * 40503u) ^ (uint32_t)((round + 1) * 97u);
          h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
          out = (T)((double)(h >> 8) * (1.0 / 8388608.0) - 1.0);
        }
        Xb[(int64_t)d * vs + col] = out;
        if constexpr (GRAM) xt[d * kOrthXtLd + cl] = out;
      }
    }
    if constexpr (GRAM) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave, in-order LDS: the tile is complete)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if (it >= it0 && it < nt) {   // (wave-uniform)
          T pa[4];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) pa[s4] = xt[(16 * it + cl) * kOrthXtLd + 4 * s4 + g];
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            if (jt < nt) {
#pragma unroll
              for (int s4 = 0; s4 < 4; ++s4) Gacc[it][jt] = M::mma(pa[s4], xt[(16 * jt + cl) * kOrthXtLd + 4 * s4 + g], Gacc[it][jt]);
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the reads are done before the next slab's tile is written)
    }
    if (sidx + 1 < nsteps) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[ks] = an[ks];
    }
  }
  if constexpr (GRAM) {
    T* __restrict__ Gp = Gnext + ((b * nsplit + blockIdx.x) * 4 + wave) * (int64_t)r_launch * r_launch;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = 16 * it + M::row(lane, u), j = 16 * jt + cl;
          if (i < r_launch && j < r_launch) Gp[i * r_launch + j] = Gacc[it][jt][u];
        }
  }
}

// implemented in the other translation units
int gemm_dispatch(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                  int64_t strideA, const void* B, int64_t ldb, int64_t strideB, void* C, int64_t ldc, int64_t strideC,
                  const void* rs, int64_t stride_rs, int rs_mode, const void* cs, int64_t stride_cs, int cs_mode,
                  int64_t batch, void* ws, int64_t ws_bytes, hipStream_t stream, int axpby = 0, double alpha = 1.0,
                  double beta = 0.0);
int64_t gemm_workspace_bytes(int dtype, int64_t M, int64_t N, int64_t K, int64_t batch);
int krp_contract_dispatch(int dtype, int64_t P, int64_t J, int64_t Q, int64_t R, const void* Tn, const void* B,
                          int64_t ldb, void* out, hipStream_t stream);
int hadamard_dispatch(int dtype, int64_t count, const void* a, const void* b, void* out, hipStream_t stream);
int core_kron_dispatch(int dtype, int64_t B, int64_t R1, int64_t S1, int64_t I, int64_t R2, int64_t S2, const void* a,
                       const void* c, void* out, hipStream_t stream);
int qr_factor_dispatch(int dtype, int64_t m, int64_t n, int64_t batch, const void* A, int64_t lda, int64_t strideA,
                       void* R, int64_t ldr, int64_t strideR, void* ws, int64_t ws_bytes, hipStream_t stream, int64_t a_cs = 1,
                       int32_t* expo_acc = nullptr);
int qr_apply_dispatch(int dtype, int64_t m, int64_t n, int64_t batch, void* ws, int64_t ws_bytes, const void* C,
                      int64_t ldc, int64_t strideC, int64_t kc, void* Out, int64_t ldo, int64_t strideO,
                      hipStream_t stream, int64_t o_cs = 1);
int64_t qr_workspace_bytes(int dtype, int64_t m, int64_t n, int64_t batch);
int64_t qr_pushed_workspace_bytes(int dtype, int64_t I, int64_t n, int64_t batch);
int qr_factor_pushed_dispatch(int dtype, int64_t k, int64_t Rin, int64_t I, int64_t n, int64_t batch, const void* Rm,
                              int64_t ldrm, int64_t strideRm, const void* Cn, int64_t strideCn, void* R, int64_t ldr,
                              int64_t strideR, void* ws, int64_t ws_bytes, hipStream_t stream, int32_t* expo_acc = nullptr);
int qr_factor_pushed_sum_dispatch(int dtype, int64_t k, int64_t I, int64_t batch, const void* Rm, int64_t ldrm,
                                  int64_t strideRm, const void* Ca, int64_t ra, int64_t ca, int64_t strideCa,
                                  const void* Cb, int64_t rb, int64_t cb, int64_t strideCb, void* R, int64_t ldr,
                                  int64_t strideR, void* ws, int64_t ws_bytes, hipStream_t stream);
int qr_apply_pushed_dispatch(int dtype, int64_t k, int64_t I, int64_t n, int64_t batch, void* ws, int64_t ws_bytes,
                             const void* C, int64_t ldc, int64_t strideC, int64_t kc, void* Out, int64_t ldo,
                             int64_t strideO, void* G, hipStream_t stream, int skip_zero_rows = 0);
int64_t qr_apply_pushed_gram_parts(int dtype, int64_t k, int64_t I, int64_t n, int64_t kc);
int qr_max_cols(int dtype);
int eigh_dispatch(int dtype, int64_t n, int64_t batch, const void* G, int64_t ldg, int64_t strideG, int64_t gparts,
                  int64_t stride_gpart, void* V, int64_t ldv, int64_t strideV, void* sigma, int64_t stride_sigma, int32_t* info, int eig_mode,
                  int use_delta, double delta2, int64_t rmax, int abs_floor, int32_t* sweeps, void* ws,
                  int64_t ws_bytes, hipStream_t stream, const double* delta2_dev = nullptr, const int32_t* skip_items = nullptr,
                  const void* sigma_in = nullptr, int64_t stride_sigma_in = 0);
int eigh_top_dispatch(int dtype, int64_t n, int64_t batch, const void* G, int64_t ldg, int64_t strideG, int64_t gparts,
                      int64_t stride_gpart, void* V, int64_t ldv, int64_t strideV, void* sigma, int64_t stride_sigma, int32_t* info,
                      int64_t r, double thr, int32_t* flat, hipStream_t stream, int need_all);
int eigh_pairs_dispatch(int dtype, int64_t b, int64_t npairs, int64_t items, const void* G, int64_t ldg, int64_t strideG,
                        const int32_t* pair_tab, void* W, void* scratch, const int32_t* skip_flag, int32_t* rot_count,
                        hipStream_t stream);
int bj_apply_dispatch(int dtype, int64_t b, int64_t npairs, int64_t items, void* G, int64_t ldg, int64_t strideG, void* V,
                      int64_t ldv, int64_t strideV, const int32_t* pair_tab, const void* W, const int32_t* ctrl, double* offsq,
                      hipStream_t stream);
int bj_control_dispatch(int dtype, int64_t items, int32_t* ctrl, double* state, const void* gnorm, int relative, double tol,
                        hipStream_t stream);
int64_t eigsel_scratch_bytes(int dtype, int64_t n, int64_t batch);
int eigsel_max_n();
int64_t tridiag_workspace_bytes(int dtype, int64_t n, int64_t batch);
int tridiag_dispatch(int dtype, int64_t n, int64_t batch, void* A, int64_t lda, int64_t strideA, void* d, void* e, void* tau,
                     void* ws, hipStream_t stream);
int eigsel_dispatch(int dtype, int64_t n, int64_t batch, int64_t k, const void* d, const void* e, void* lam, void* Z, void* scratch,
                    hipStream_t stream);
int tridiag_back_dispatch(int dtype, int64_t n, int64_t batch, int64_t k, const void* A, int64_t lda, int64_t strideA, const void* tau,
                          void* Z, hipStream_t stream);
int64_t eigh_workspace_bytes(int dtype, int64_t n, int64_t batch);
int eigh_max_n(int dtype);
int eigh_max_n_lds(int dtype);

int sweep_gram_parts(int64_t n, int64_t batch);
int sweep_gram_dispatch(int dtype, int64_t R, int64_t n, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM,
                        const void* V1, int64_t ldv1, int64_t strideV1, void* G, int64_t nsplit, hipStream_t stream,
                        const int32_t* skip = nullptr, const int32_t* rows32 = nullptr);
int sweep_project_dispatch(int dtype, int64_t R, int64_t n, int64_t ro, int64_t batch, const void* Mx, int64_t ldm,
                           int64_t strideM, const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2,
                           int64_t strideV2, const void* sigma, int64_t stride_sigma, int scale_right, void* right,
                           int64_t ldr, int64_t strideR, void* left, int64_t ldl, int64_t strideL, hipStream_t stream,
                           const int32_t* rows32 = nullptr);
int64_t qr_pushed_flag_offset(int dtype, int64_t I, int64_t n, int64_t batch);

int64_t colgram_workspace_bytes(int dtype, int64_t rows, int64_t n, int64_t batch);
int colgram_dispatch(int dtype, int64_t rows, int64_t n, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM,
                     const void* V1, int64_t ldv1, int64_t strideV1, void* G, void* ws, int64_t ws_bytes, hipStream_t stream,
                     const int32_t* skip = nullptr);
int colproject_dispatch(int dtype, int64_t rows, int64_t n, int64_t ro, int64_t batch, const void* Mx, int64_t ldm,
                        int64_t strideM, const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2,
                        int64_t strideV2, const void* sigma, int64_t stride_sigma, int left_ortho, void* left, int64_t ldl,
                        int64_t strideL, void* right, int64_t ldr, int64_t strideR, hipStream_t stream);

extern long long* g_qr_dbg;
extern int g_qr_variant;
extern int g_bj_inner_sweeps;
extern int g_gemm_big;
extern int g_qr_dbg_bx, g_qr_dbg_by;
extern int g_qr_f64_nw4;
extern int g_rank_skip_c;
extern int g_qr_pack;
extern int g_qr_interleave;
extern int g_qr_stagger;
extern int g_qr_pack_pre;
extern int g_qr_l1_idle;
extern int g_eigh_big_occ;
extern int g_sweep_stagger;
extern int g_rank_noise_c;
extern int g_jacobi_live_wave;
extern int g_eigh_small;

static bool dtype_ok(int dtype) { return dtype == TTR_F32 || dtype == TTR_F64; }

}  // namespace ttr

using namespace ttr;

static int g_orth_rounds = 4;   // ttr_debug_set_knob(TTR_KNOB_ORTH_ROUNDS): rounds of the block orthonormal completion (diagnostics)
static int g_orth_v2 = 2;       // ttr_debug_set_knob(TTR_KNOB_ORTH_V2): 0 = round 4's inner loops, 1 = round 5's, 2 = + the three-launch rounds' fused Gram (A/B)
// ttr_debug_set_knob(TTR_KNOB_ORTH_SPLIT): batches from this size (per launch, i.e. per sub-batch stream) take the three-launch
// rounds; 0 = never.  Where most kept directions lie below the resolution (sigma ~ 2^-j) the rounds cost 6.0 instead of 8.4 ms per
// 2048-train step (B = 4096: step 50.4 -> 44.5 ms, 650 k -> 736 k cores/s); a batch WITHOUT dead directions pays for 13 launches
// per bond that exit at once instead of one: nothing measurable from 2048 items per launch (headline at B = 4096, four
// alternations: 25.515 vs 25.524 ms), 0.3 - 2 % at 1024 (profiles/r05_orth_split_ab.txt, r05_orth_split_headline_ab.txt) --
// hence the threshold.
static int g_orth_split = 2048;

// (vectors as rows of a row-major matrix, at most 64 of them, at least 512 elements each, a batch that fills the chip)
static bool orth_split_ok(int64_t r, int64_t n, int64_t batch, int64_t elem_stride) {
  return g_orth_split > 0 && batch >= g_orth_split && batch <= 65535 && elem_stride == 1 && r >= 2 && r <= 64 && n >= 512;
}

template <typename T>
static int orth_split_run(int64_t r, int64_t n, int64_t batch, T* X, int64_t vs, int64_t strideX, const T* sigma, int64_t stride_sigma,
                          double dead_rel, const int32_t* rank_dev, char* ws, hipStream_t s) {
  const int dtype = sizeof(T) == 4 ? TTR_F32 : TTR_F64;
  const OrthSplitWs L = orth_split_layout(r, n, batch, sizeof(T), g_orth_rounds);
  T* G = (T*)(ws + L.off_g);
  T* Wg = (T*)(ws + L.off_w);
  unsigned long long* regen = (unsigned long long*)(ws + L.off_regen);
  int32_t* skip = (int32_t*)(ws + L.off_skip);
  double* census = work_census_on() ? g_work_dev : nullptr;
  {
    ProfScope prof(TTR_PROF_MISC, s);
    hipLaunchKernelGGL(orth_init_kernel<T>, dim3((unsigned)ceil_div(batch, kThreads)), dim3(kThreads), 0, s, (int)r, batch, sigma, stride_sigma,
                       dead_rel, rank_dev, skip);
  }
  const int r4 = ((int)r + 15) & ~15;
  const size_t lds = 2 * (size_t)r4 * (r4 + 1) * 8 + 64 * 4 + 16;
  const int nsplit = L.nsplit;
  // at most 32 vectors: round k's apply launch leaves the Gram matrix round k + 1 starts from (orth_apply_kernel<T, true>)
  const bool fuse = g_orth_v2 >= 2 && r4 <= 32;
  // Two rounds ("twice is enough"); what is left after them -- items whose remainders collapsed and were replaced by hashed
  // vectors: rare -- is finished by the single-launch kernel (one launch that exits at once where nothing is left, instead of
  // six more launches per bond that do).
  const int nrounds = g_orth_rounds < kOrthSplitRounds ? g_orth_rounds : kOrthSplitRounds;
  for (int round = 0; round < nrounds; ++round) {
    const int32_t* sk = skip + (int64_t)round * batch;
    TTR_HIP_CHECK(hipGetLastError());
    const bool have_gram = fuse && round > 0;   // left by the previous round's apply launch
    if (!have_gram) {
      const int rc = sweep_gram_dispatch(dtype, r, n, batch, X, vs, strideX, nullptr, 0, 0, G, L.parts, s, sk, nullptr);
      if (rc != TTR_OK) return rc;
    }
    ProfScope prof(TTR_PROF_MISC, s);
    auto kern = orth_coef_kernel<T>;
    if (lds > 64 * 1024) TTR_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(kThreads), lds, s, (int)r, (const T*)G, have_gram ? L.fparts : L.parts, Wg, regen,
                       sk, skip + (int64_t)(round + 1) * batch, sigma, stride_sigma, dead_rel, rank_dev, round, census);
    if (fuse && round + 1 < nrounds)
      hipLaunchKernelGGL((orth_apply_kernel<T, true>), dim3((unsigned)nsplit, (unsigned)batch), dim3(kThreads), 0, s, (int)r, n, X, vs,
                         strideX, (const T*)Wg, (const unsigned long long*)regen, sk, sigma, stride_sigma, dead_rel, rank_dev, round,
                         nsplit, G);
    else
      hipLaunchKernelGGL((orth_apply_kernel<T, false>), dim3((unsigned)nsplit, (unsigned)batch), dim3(kThreads), 0, s, (int)r, n, X, vs,
                         strideX, (const T*)Wg, (const unsigned long long*)regen, sk, sigma, stride_sigma, dead_rel, rank_dev, round,
                         nsplit, (T*)nullptr);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}


extern "C" {

int ttr_version(void) { return TTR_ABI_VERSION; }

const char* ttr_last_error(void) { return g_err.c_str(); }

int ttr_qr_max_cols(int dtype) { return qr_max_cols(dtype); }
int ttr_eigh_max_n_lds(int dtype) { return eigh_max_n_lds(dtype); }
int ttr_eigh_max_n(int dtype) { return eigh_max_n(dtype); }

int64_t ttr_gemm_workspace_bytes(int dtype, int64_t M, int64_t N, int64_t K, int64_t batch) {
  return gemm_workspace_bytes(dtype, M, N, K, batch);
}

int ttr_gemm(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
             int64_t strideA, const void* B, int64_t ldb, int64_t strideB, void* C, int64_t ldc, int64_t strideC,
             const void* rowscale, int64_t stride_rs, int rowscale_mode, const void* colscale, int64_t stride_cs,
             int colscale_mode, int64_t batch, void* workspace, int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_gemm: bad dtype %d", dtype);
  TTR_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, TTR_E_INVALID, "ttr_gemm: negative dimension");
  if (M == 0 || N == 0 || batch == 0) return TTR_OK;
  TTR_REQUIRE(K >= 1, TTR_E_INVALID, "ttr_gemm: K must be >= 1");
  TTR_REQUIRE(A && B && C, TTR_E_INVALID, "ttr_gemm: null operand");
  return gemm_dispatch(dtype, transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, rowscale,
                       stride_rs, rowscale_mode, colscale, stride_cs, colscale_mode, batch, workspace, workspace_bytes,
                       (hipStream_t)stream);
}

int ttr_gemm_axpby(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                   int64_t strideA, const void* B, int64_t ldb, int64_t strideB, void* C, int64_t ldc, int64_t strideC,
                   double alpha, double beta, int64_t batch, void* workspace, int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_gemm_axpby: bad dtype %d", dtype);
  TTR_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, TTR_E_INVALID, "ttr_gemm_axpby: negative dimension");
  if (M == 0 || N == 0 || batch == 0) return TTR_OK;
  TTR_REQUIRE(K >= 1, TTR_E_INVALID, "ttr_gemm_axpby: K must be >= 1");
  TTR_REQUIRE(A && B && C, TTR_E_INVALID, "ttr_gemm_axpby: null operand");
  return gemm_dispatch(dtype, transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, nullptr, 0,
                       TTR_SCALE_NONE, nullptr, 0, TTR_SCALE_NONE, batch, workspace, workspace_bytes,
                       (hipStream_t)stream, 1, alpha, beta);
}

int64_t ttr_qr_workspace_bytes(int dtype, int64_t m, int64_t n, int64_t batch) {
  return qr_workspace_bytes(dtype, m, n, batch);
}

int ttr_qr_factor(int dtype, int64_t m, int64_t n, int64_t batch, const void* A, int64_t lda, int64_t strideA, void* R,
                  int64_t ldr, int64_t strideR, void* workspace, int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_qr_factor: bad dtype %d", dtype);
  TTR_REQUIRE(m >= 1 && n >= 1 && batch >= 0, TTR_E_INVALID, "ttr_qr_factor: bad shape %lld x %lld", (long long)m,
              (long long)n);
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(A && R && workspace, TTR_E_INVALID, "ttr_qr_factor: null pointer");
  return qr_factor_dispatch(dtype, m, n, batch, A, lda, strideA, R, ldr, strideR, workspace, workspace_bytes,
                            (hipStream_t)stream);
}

int ttr_qr_factor_expo(int dtype, int64_t m, int64_t n, int64_t batch, const void* A, int64_t lda, int64_t strideA, void* R,
                       int64_t ldr, int64_t strideR, void* workspace, int64_t workspace_bytes, int32_t* expo_acc, void* stream) {
  TTR_REQUIRE(dtype == TTR_F32, TTR_E_UNSUPPORTED, "ttr_qr_factor_expo: fp32 only (dtype %d)", dtype);
  TTR_REQUIRE(m >= 1 && n >= 1 && batch >= 0, TTR_E_INVALID, "ttr_qr_factor_expo: bad shape %lld x %lld", (long long)m,
              (long long)n);
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(A && R && workspace && expo_acc, TTR_E_INVALID, "ttr_qr_factor_expo: null pointer");
  return qr_factor_dispatch(dtype, m, n, batch, A, lda, strideA, R, ldr, strideR, workspace, workspace_bytes,
                            (hipStream_t)stream, 1, expo_acc);
}

int ttr_qr_apply(int dtype, int64_t m, int64_t n, int64_t batch, void* workspace, int64_t workspace_bytes,
                 const void* C, int64_t ldc, int64_t strideC, int64_t kcols, void* Out, int64_t ldo, int64_t strideO,
                 void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_qr_apply: bad dtype %d", dtype);
  TTR_REQUIRE(m >= 1 && n >= 1 && batch >= 0 && kcols >= 1, TTR_E_INVALID, "ttr_qr_apply: bad shape");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(Out && workspace, TTR_E_INVALID, "ttr_qr_apply: null pointer");
  TTR_REQUIRE(kcols <= (m < n ? m : n), TTR_E_INVALID, "ttr_qr_apply: kcols %lld > min(m, n)", (long long)kcols);
  return qr_apply_dispatch(dtype, m, n, batch, workspace, workspace_bytes, C, ldc, strideC, kcols, Out, ldo, strideO,
                           (hipStream_t)stream);
}

int ttr_qr(int dtype, int64_t m, int64_t n, int64_t batch, const void* A, int64_t lda, int64_t strideA, void* Q,
           int64_t ldq, int64_t strideQ, void* R, int64_t ldr, int64_t strideR, void* workspace,
           int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(Q != nullptr || batch == 0, TTR_E_INVALID, "ttr_qr: null pointer");
  int rc = ttr_qr_factor(dtype, m, n, batch, A, lda, strideA, R, ldr, strideR, workspace, workspace_bytes, stream);
  if (rc != TTR_OK || batch == 0) return rc;
  return ttr_qr_apply(dtype, m, n, batch, workspace, workspace_bytes, nullptr, 0, 0, m < n ? m : n, Q, ldq, strideQ,
                      stream);
}

int ttr_qr_t(int dtype, int64_t m, int64_t n, int64_t batch, const void* At, int64_t ldat, int64_t strideAt, void* Qt,
             int64_t ldqt, int64_t strideQt, void* R, int64_t ldr, int64_t strideR, void* workspace,
             int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_qr_t: bad dtype %d", dtype);
  TTR_REQUIRE(m >= 1 && n >= 1 && batch >= 0, TTR_E_INVALID, "ttr_qr_t: bad shape %lld x %lld", (long long)m, (long long)n);
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(At && Qt && R && workspace, TTR_E_INVALID, "ttr_qr_t: null pointer");
  TTR_REQUIRE(ldat >= m && ldqt >= m, TTR_E_INVALID, "ttr_qr_t: leading dimensions below m");
  // the factored matrix is A = At^T: element (row, col) at At[col * ldat + row]; Q^T goes out the same way
  int rc = qr_factor_dispatch(dtype, m, n, batch, At, 1, strideAt, R, ldr, strideR, workspace, workspace_bytes, (hipStream_t)stream, ldat);
  if (rc != TTR_OK) return rc;
  return qr_apply_dispatch(dtype, m, n, batch, workspace, workspace_bytes, nullptr, 0, 0, m < n ? m : n, Qt, 1, strideQt,
                           (hipStream_t)stream, ldqt);
}

int64_t ttr_qr_pushed_workspace_bytes(int dtype, int64_t I, int64_t n, int64_t batch) {
  return qr_pushed_workspace_bytes(dtype, I, n, batch);
}

int ttr_qr_factor_pushed(int dtype, int64_t k, int64_t Rin, int64_t I, int64_t n, int64_t batch, const void* Rm,
                         int64_t ldrm, int64_t strideRm, const void* core, int64_t stride_core, void* R, int64_t ldr,
                         int64_t strideR, void* workspace, int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_qr_factor_pushed: bad dtype %d", dtype);
  TTR_REQUIRE(batch >= 0, TTR_E_INVALID, "ttr_qr_factor_pushed: negative batch");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(Rm && core && R && workspace, TTR_E_INVALID, "ttr_qr_factor_pushed: null pointer");
  return qr_factor_pushed_dispatch(dtype, k, Rin, I, n, batch, Rm, ldrm, strideRm, core, stride_core, R, ldr, strideR,
                                   workspace, workspace_bytes, (hipStream_t)stream);
}

int ttr_qr_factor_pushed_expo(int dtype, int64_t k, int64_t Rin, int64_t I, int64_t n, int64_t batch, const void* Rm,
                              int64_t ldrm, int64_t strideRm, const void* core, int64_t stride_core, void* R, int64_t ldr,
                              int64_t strideR, void* workspace, int64_t workspace_bytes, int32_t* expo_acc, void* stream) {
  TTR_REQUIRE(dtype == TTR_F32, TTR_E_UNSUPPORTED, "ttr_qr_factor_pushed_expo: fp32 only (dtype %d)", dtype);
  TTR_REQUIRE(batch >= 0, TTR_E_INVALID, "ttr_qr_factor_pushed_expo: negative batch");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(Rm && core && R && workspace && expo_acc, TTR_E_INVALID, "ttr_qr_factor_pushed_expo: null pointer");
  return qr_factor_pushed_dispatch(dtype, k, Rin, I, n, batch, Rm, ldrm, strideRm, core, stride_core, R, ldr, strideR,
                                   workspace, workspace_bytes, (hipStream_t)stream, expo_acc);
}

int ttr_qr_factor_pushed_sum(int dtype, int64_t k, int64_t I, int64_t batch, const void* Rm, int64_t ldrm,
                             int64_t strideRm, const void* core_a, int64_t ra, int64_t ca, int64_t stride_a,
                             const void* core_b, int64_t rb, int64_t cb, int64_t stride_b, void* R, int64_t ldr,
                             int64_t strideR, void* workspace, int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_qr_factor_pushed_sum: bad dtype %d", dtype);
  TTR_REQUIRE(batch >= 0, TTR_E_INVALID, "ttr_qr_factor_pushed_sum: negative batch");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(Rm && core_a && core_b && R && workspace, TTR_E_INVALID, "ttr_qr_factor_pushed_sum: null pointer");
  return qr_factor_pushed_sum_dispatch(dtype, k, I, batch, Rm, ldrm, strideRm, core_a, ra, ca, stride_a, core_b, rb, cb,
                                       stride_b, R, ldr, strideR, workspace, workspace_bytes, (hipStream_t)stream);
}

int ttr_qr_apply_pushed(int dtype, int64_t k, int64_t I, int64_t n, int64_t batch, void* workspace,
                        int64_t workspace_bytes, const void* C, int64_t ldc, int64_t strideC, int64_t kcols, void* Out,
                        int64_t ldo, int64_t strideO, int skip_zero_rows, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_qr_apply_pushed: bad dtype %d", dtype);
  TTR_REQUIRE(batch >= 0 && kcols >= 1 && kcols <= n, TTR_E_INVALID, "ttr_qr_apply_pushed: bad arguments");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(Out && workspace, TTR_E_INVALID, "ttr_qr_apply_pushed: null pointer");
  return qr_apply_pushed_dispatch(dtype, k, I, n, batch, workspace, workspace_bytes, C, ldc, strideC, kcols, Out, ldo,
                                  strideO, nullptr, (hipStream_t)stream, skip_zero_rows ? 1 : 0);
}

int64_t ttr_qr_apply_pushed_gram_parts(int dtype, int64_t k, int64_t I, int64_t n, int64_t kcols) {
  if (!dtype_ok(dtype)) return 0;
  return qr_apply_pushed_gram_parts(dtype, k, I, n, kcols);
}

int ttr_qr_apply_pushed_gram(int dtype, int64_t k, int64_t I, int64_t n, int64_t batch, void* workspace,
                             int64_t workspace_bytes, const void* C, int64_t ldc, int64_t strideC, int64_t kcols,
                             void* Out, int64_t ldo, int64_t strideO, void* G, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_qr_apply_pushed_gram: bad dtype %d", dtype);
  TTR_REQUIRE(batch >= 0 && kcols >= 1 && kcols <= n, TTR_E_INVALID, "ttr_qr_apply_pushed_gram: bad arguments");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(Out && workspace && G, TTR_E_INVALID, "ttr_qr_apply_pushed_gram: null pointer");
  return qr_apply_pushed_dispatch(dtype, k, I, n, batch, workspace, workspace_bytes, C, ldc, strideC, kcols, Out, ldo,
                                  strideO, G, (hipStream_t)stream);
}

int64_t ttr_eigh_workspace_bytes(int dtype, int64_t n, int64_t batch) { return eigh_workspace_bytes(dtype, n, batch); }

int ttr_eigh_trunc(int dtype, int64_t n, int64_t batch, const void* G, int64_t ldg, int64_t strideG, int64_t gparts,
                   int64_t stride_gpart, void* V, int64_t ldv, int64_t strideV, void* sigma, int64_t stride_sigma,
                   int32_t* info, int eig_mode, int use_delta, double delta2, const double* delta2_dev, int64_t rmax,
                   int abs_floor, int32_t* sweeps, const int32_t* skip_items, const void* sigma_in, int64_t stride_sigma_in,
                   void* workspace, int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_eigh_trunc: bad dtype %d", dtype);
  TTR_REQUIRE(n >= 1 && batch >= 0 && rmax >= 1 && gparts >= 1, TTR_E_INVALID, "ttr_eigh_trunc: bad arguments");
  TTR_REQUIRE(abs_floor >= TTR_SOLVER_JACOBI_REL && abs_floor <= TTR_SOLVER_JACOBI_LIVE, TTR_E_INVALID,
              "ttr_eigh_trunc: bad solver %d", abs_floor);
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(G && V && sigma && info, TTR_E_INVALID, "ttr_eigh_trunc: null pointer");
  TTR_REQUIRE(eig_mode >= TTR_EIG_RAW && eig_mode <= TTR_EIG_MATCH_DIAG, TTR_E_INVALID, "ttr_eigh_trunc: bad eig_mode %d",
              eig_mode);
  return eigh_dispatch(dtype, n, batch, G, ldg, strideG, gparts, stride_gpart, V, ldv, strideV, sigma, stride_sigma, info,
                       eig_mode, use_delta, delta2, rmax, abs_floor, sweeps, workspace, workspace_bytes,
                       (hipStream_t)stream, delta2_dev, skip_items, sigma_in, stride_sigma_in);
}

int ttr_eigh_top_ok(int64_t n, int64_t r) { return n >= 40 && n <= 64 && r >= 1 && r <= 32 && r < n; }

int ttr_eigh_top(int dtype, int64_t n, int64_t batch, const void* G, int64_t ldg, int64_t strideG, int64_t gparts,
                 int64_t stride_gpart, void* V, int64_t ldv, int64_t strideV, void* sigma, int64_t stride_sigma, int32_t* info,
                 int64_t r, double thr, int32_t* flat, int need_all, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_eigh_top: bad dtype %d", dtype);
  TTR_REQUIRE(batch >= 0 && gparts >= 1 && thr > 0.0 && thr <= 1.0, TTR_E_INVALID, "ttr_eigh_top: bad arguments");
  TTR_REQUIRE(ttr_eigh_top_ok(n, r), TTR_E_UNSUPPORTED, "ttr_eigh_top: n = %lld, r = %lld outside 40 <= n <= 64, r <= 32, r < n",
              (long long)n, (long long)r);
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(G && V && sigma && info, TTR_E_INVALID, "ttr_eigh_top: null pointer");
  return eigh_top_dispatch(dtype, n, batch, G, ldg, strideG, gparts, stride_gpart, V, ldv, strideV, sigma, stride_sigma, info, r, thr,
                           flat, (hipStream_t)stream, need_all);
}

int64_t ttr_bj_scratch_bytes(int dtype, int64_t b, int64_t npairs, int64_t items) {
  const int64_t elem = dtype == TTR_F64 ? 8 : 4;
  return items * npairs * (2 * b + 1) * elem;
}

static int bj_ok(const char* who, int dtype, int64_t b, int64_t npairs, int64_t items) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "%s: bad dtype %d", who, dtype);
  TTR_REQUIRE(b >= 1 && b <= 32 && npairs >= 1 && npairs <= 65535 && items >= 0, TTR_E_UNSUPPORTED,
              "%s: block width %lld outside [1, 32] or %lld pairs outside [1, 65535]", who, (long long)b, (long long)npairs);
  return TTR_OK;
}

int ttr_bj_solve(int dtype, int64_t b, int64_t npairs, int64_t items, const void* G, int64_t ldg, int64_t strideG,
                 const int32_t* pair_tab, void* W, void* scratch, int32_t* ctrl, void* stream) {
  const int rc = bj_ok("ttr_bj_solve", dtype, b, npairs, items);
  if (rc != TTR_OK) return rc;
  if (items == 0) return TTR_OK;
  TTR_REQUIRE(G && pair_tab && W && scratch && ctrl, TTR_E_INVALID, "ttr_bj_solve: null pointer");
  TTR_REQUIRE(items * npairs < (int64_t(1) << 31), TTR_E_UNSUPPORTED, "ttr_bj_solve: %lld pair problems exceed the grid",
              (long long)(items * npairs));
  return eigh_pairs_dispatch(dtype, b, npairs, items, G, ldg, strideG, pair_tab, W, scratch, ctrl, ctrl + 1, (hipStream_t)stream);
}

int ttr_bj_apply(int dtype, int64_t b, int64_t npairs, int64_t items, void* G, int64_t ldg, int64_t strideG, void* V,
                 int64_t ldv, int64_t strideV, const int32_t* pair_tab, const void* W, const int32_t* ctrl, double* offsq,
                 void* stream) {
  const int rc = bj_ok("ttr_bj_apply", dtype, b, npairs, items);
  if (rc != TTR_OK) return rc;
  if (items == 0) return TTR_OK;
  TTR_REQUIRE(G && V && pair_tab && W && ctrl, TTR_E_INVALID, "ttr_bj_apply: null pointer");
  return bj_apply_dispatch(dtype, b, npairs, items, G, ldg, strideG, V, ldv, strideV, pair_tab, W, ctrl, offsq, (hipStream_t)stream);
}

int ttr_bj_control(int dtype, int64_t items, int32_t* ctrl, double* state, const void* gnorm, int relative, double tol,
                   void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_bj_control: bad dtype %d", dtype);
  TTR_REQUIRE(ctrl && state && (relative || gnorm), TTR_E_INVALID, "ttr_bj_control: null pointer");
  return bj_control_dispatch(dtype, items, ctrl, state, gnorm, relative, tol, (hipStream_t)stream);
}

int ttr_norm(int dtype, int64_t count, int64_t batch, const void* x, int64_t stride_x, void* out, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_norm: bad dtype %d", dtype);
  TTR_REQUIRE(count >= 0 && batch >= 0, TTR_E_INVALID, "ttr_norm: negative size");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(x && out, TTR_E_INVALID, "ttr_norm: null pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(TTR_PROF_MISC, s);
  if (dtype == TTR_F32)
    hipLaunchKernelGGL(norm_kernel<float>, dim3((unsigned)batch), dim3(kThreads), 0, s, (const float*)x, count,
                       stride_x, (float*)out);
  else
    hipLaunchKernelGGL(norm_kernel<double>, dim3((unsigned)batch), dim3(kThreads), 0, s, (const double*)x, count,
                       stride_x, (double*)out);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int ttr_scale_cols(int dtype, int64_t rows, int64_t cols, int64_t batch, const void* in, int64_t ldi,
                   int64_t stride_in, const void* sc, int64_t stride_s, int mode, void* out, int64_t ldo,
                   int64_t stride_out, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_scale_cols: bad dtype %d", dtype);
  TTR_REQUIRE(mode == TTR_SCALE_MUL || mode == TTR_SCALE_DIV, TTR_E_INVALID, "ttr_scale_cols: bad mode");
  if (rows <= 0 || cols <= 0 || batch <= 0) return TTR_OK;
  TTR_REQUIRE(in && sc && out, TTR_E_INVALID, "ttr_scale_cols: null pointer");
  if (batch > 65535) {  // the batch is a grid dimension: slices
    const int64_t elem = dtype == TTR_F64 ? 8 : 4;
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
      const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
      const int rc = ttr_scale_cols(dtype, rows, cols, nb, (const char*)in + b0 * stride_in * elem, ldi, stride_in,
                                    (const char*)sc + b0 * stride_s * elem, stride_s, mode, (char*)out + b0 * stride_out * elem, ldo,
                                    stride_out, stream);
      if (rc != TTR_OK) return rc;
    }
    return TTR_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  int64_t gx = ceil_div(rows * cols, kThreads);
  if (gx > 2048) gx = 2048;
  ProfScope prof(TTR_PROF_MISC, s);
  if (dtype == TTR_F32)
    hipLaunchKernelGGL(scale_cols_kernel<float>, dim3((unsigned)gx, (unsigned)batch), dim3(kThreads), 0, s, rows, cols,
                       (const float*)in, ldi, stride_in, (const float*)sc, stride_s, mode, (float*)out, ldo, stride_out);
  else
    hipLaunchKernelGGL(scale_cols_kernel<double>, dim3((unsigned)gx, (unsigned)batch), dim3(kThreads), 0, s, rows,
                       cols, (const double*)in, ldi, stride_in, (const double*)sc, stride_s, mode, (double*)out, ldo,
                       stride_out);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int ttr_mask_cols(int dtype, int64_t rows, int64_t cols, int64_t batch, void* x, int64_t ldx, int64_t stride_x,
                  const int32_t* keep, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_mask_cols: bad dtype %d", dtype);
  if (rows <= 0 || cols <= 0 || batch <= 0) return TTR_OK;
  TTR_REQUIRE(x && keep, TTR_E_INVALID, "ttr_mask_cols: null pointer");
  TTR_REQUIRE(batch <= 65535, TTR_E_UNSUPPORTED, "ttr_mask_cols: batch %lld > 65535", (long long)batch);
  hipStream_t s = (hipStream_t)stream;
  int64_t gx = ceil_div(rows * cols, kThreads);
  if (gx > 1024) gx = 1024;
  ProfScope prof(TTR_PROF_MISC, s);
  if (dtype == TTR_F32)
    hipLaunchKernelGGL(mask_cols_kernel<float>, dim3((unsigned)gx, (unsigned)batch), dim3(kThreads), 0, s, rows, cols, (float*)x,
                       ldx, stride_x, keep);
  else
    hipLaunchKernelGGL(mask_cols_kernel<double>, dim3((unsigned)gx, (unsigned)batch), dim3(kThreads), 0, s, rows, cols,
                       (double*)x, ldx, stride_x, keep);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int64_t ttr_sweep_gram_parts(int64_t n, int64_t batch) {
  if (n <= 0 || batch <= 0) return 1;
  return sweep_gram_parts(n, batch);
}

int ttr_rowgram(int dtype, int64_t R, int64_t n, int64_t batch, const void* M, int64_t ldm, int64_t strideM, void* G,
                int64_t nparts, const int32_t* rows32, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_rowgram: bad dtype %d", dtype);
  TTR_REQUIRE(R >= 1 && n >= 1 && batch >= 0, TTR_E_INVALID, "ttr_rowgram: bad shape");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(M && G, TTR_E_INVALID, "ttr_rowgram: null pointer");
  return sweep_gram_dispatch(dtype, R, n, batch, M, ldm, strideM, nullptr, 0, 0, G, nparts, (hipStream_t)stream, nullptr, rows32);
}

int64_t ttr_qr_pushed_flag_offset(int dtype, int64_t I, int64_t n, int64_t batch) {
  if (!dtype_ok(dtype) || I < 1 || n < 1 || batch < 1) return -1;
  return qr_pushed_flag_offset(dtype, I, n, batch);
}

int ttr_carry_rows32(int dtype, int64_t cols, int64_t batch, const void* R, int64_t ldr, int64_t strideR, int32_t* flag,
                     void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_carry_rows32: bad dtype %d", dtype);
  TTR_REQUIRE(cols >= 1 && batch >= 0 && ldr >= cols, TTR_E_INVALID, "ttr_carry_rows32: bad shape");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(R && flag, TTR_E_INVALID, "ttr_carry_rows32: null pointer");
  hipStream_t s = (hipStream_t)stream;
  // the thresholds of the packing test (TTR_KNOB_QR_RANK_SKIP, TTR_KNOB_QR_PACK = 0: never)
  const double ce = (g_qr_pack ? (double)g_rank_skip_c : 0.0) * (dtype == TTR_F32 ? 1.1920929e-07 : 2.220446049250313e-16);
  ProfScope prof(TTR_PROF_MISC, s);
  if (dtype == TTR_F32)
    hipLaunchKernelGGL(carry_rows32_kernel<float>, dim3((unsigned)batch), dim3(kThreads), 0, s, cols, (const float*)R, ldr, strideR,
                       ce * ce, flag);
  else
    hipLaunchKernelGGL(carry_rows32_kernel<double>, dim3((unsigned)batch), dim3(kThreads), 0, s, cols, (const double*)R, ldr, strideR,
                       ce * ce, flag);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int ttr_rotgram(int dtype, int64_t R, int64_t n, int64_t batch, const void* M, int64_t ldm, int64_t strideM,
                const void* V1, int64_t ldv1, int64_t strideV1, void* G, int64_t nparts, const int32_t* skip,
                const int32_t* rows32, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_rotgram: bad dtype %d", dtype);
  TTR_REQUIRE(R >= 1 && n >= 1 && batch >= 0, TTR_E_INVALID, "ttr_rotgram: bad shape");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(M && G && V1, TTR_E_INVALID, "ttr_rotgram: null pointer");
  return sweep_gram_dispatch(dtype, R, n, batch, M, ldm, strideM, V1, ldv1, strideV1, G, nparts, (hipStream_t)stream, skip, rows32);
}

int ttr_eigsel_max_n(void) { return eigsel_max_n(); }

int64_t ttr_eigsel_scratch_bytes(int dtype, int64_t n, int64_t batch) {
  if (!dtype_ok(dtype) || n < 1 || batch < 0) return -1;
  return eigsel_scratch_bytes(dtype, n, batch);
}

int64_t ttr_tridiag_workspace_bytes(int dtype, int64_t n, int64_t batch) {
  if (!dtype_ok(dtype) || n < 1 || batch < 0) return -1;
  return tridiag_workspace_bytes(dtype, n, batch);
}

int ttr_tridiag(int dtype, int64_t n, int64_t batch, void* A, int64_t lda, int64_t strideA, void* d, void* e, void* tau,
                void* workspace, int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_tridiag: bad dtype %d", dtype);
  TTR_REQUIRE(n >= 2 && n <= eigsel_max_n(), TTR_E_UNSUPPORTED, "ttr_tridiag: n = %lld outside [2, %d]", (long long)n, eigsel_max_n());
  TTR_REQUIRE(batch >= 0 && lda >= n, TTR_E_INVALID, "ttr_tridiag: bad shape");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(A && d && e && tau, TTR_E_INVALID, "ttr_tridiag: null pointer");
  TTR_REQUIRE(!workspace || workspace_bytes >= tridiag_workspace_bytes(dtype, n, batch), TTR_E_WORKSPACE,
              "ttr_tridiag: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)tridiag_workspace_bytes(dtype, n, batch));
  return tridiag_dispatch(dtype, n, batch, A, lda, strideA, d, e, tau, workspace, (hipStream_t)stream);
}

int ttr_tri_eigsel(int dtype, int64_t n, int64_t batch, int64_t k, const void* d, const void* e, void* lam, void* Z, void* scratch,
                   int64_t scratch_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_tri_eigsel: bad dtype %d", dtype);
  TTR_REQUIRE(n >= 2 && n <= eigsel_max_n() && k >= 1 && k <= 64 && k <= n, TTR_E_UNSUPPORTED,
              "ttr_tri_eigsel: n = %lld, k = %lld outside n in [2, %d], k in [1, min(64, n)]", (long long)n, (long long)k, eigsel_max_n());
  TTR_REQUIRE(batch >= 0, TTR_E_INVALID, "ttr_tri_eigsel: bad batch");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(d && e && lam && Z && scratch, TTR_E_INVALID, "ttr_tri_eigsel: null pointer");
  TTR_REQUIRE(scratch_bytes >= eigsel_scratch_bytes(dtype, n, batch), TTR_E_WORKSPACE, "ttr_tri_eigsel: scratch %lld < %lld bytes",
              (long long)scratch_bytes, (long long)eigsel_scratch_bytes(dtype, n, batch));
  return eigsel_dispatch(dtype, n, batch, k, d, e, lam, Z, scratch, (hipStream_t)stream);
}

int ttr_tridiag_back(int dtype, int64_t n, int64_t batch, int64_t k, const void* A, int64_t lda, int64_t strideA, const void* tau,
                     void* Z, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_tridiag_back: bad dtype %d", dtype);
  TTR_REQUIRE(n >= 2 && n <= eigsel_max_n() && k >= 1 && k <= 64, TTR_E_UNSUPPORTED, "ttr_tridiag_back: n = %lld, k = %lld unsupported",
              (long long)n, (long long)k);
  TTR_REQUIRE(batch >= 0 && lda >= n, TTR_E_INVALID, "ttr_tridiag_back: bad shape");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(A && tau && Z, TTR_E_INVALID, "ttr_tridiag_back: null pointer");
  return tridiag_back_dispatch(dtype, n, batch, k, A, lda, strideA, tau, Z, (hipStream_t)stream);
}

int ttr_spectrum_flat(int dtype, int64_t n, int64_t batch, const void* sigma, int64_t stride_sigma, int64_t keep, double thr,
                      int use_delta, double delta2, const double* delta2_dev, int32_t* flat, const int32_t* rows32,
                      void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_spectrum_flat: bad dtype %d", dtype);
  TTR_REQUIRE(n >= 1 && keep >= 1 && keep <= n && batch >= 0 && thr > 0.0 && thr <= 1.0 && delta2 >= 0.0, TTR_E_INVALID,
              "ttr_spectrum_flat: bad arguments");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(sigma && flat, TTR_E_INVALID, "ttr_spectrum_flat: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const unsigned gx = (unsigned)ceil_div(batch, kThreads);
  ProfScope prof(TTR_PROF_MISC, s);
  if (dtype == TTR_F32)
    hipLaunchKernelGGL(spectrum_flat_kernel<float>, dim3(gx), dim3(kThreads), 0, s, batch, (int)n, (int)keep, (float)thr,
                       (const float*)sigma, stride_sigma, use_delta, delta2, delta2_dev, flat, g_rank_noise_c, rows32, (int)n);
  else
    hipLaunchKernelGGL(spectrum_flat_kernel<double>, dim3(gx), dim3(kThreads), 0, s, batch, (int)n, (int)keep, thr,
                       (const double*)sigma, stride_sigma, use_delta, delta2, delta2_dev, flat, g_rank_noise_c, rows32, (int)n);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int ttr_project(int dtype, int64_t R, int64_t n, int64_t ro, int64_t batch, const void* M, int64_t ldm, int64_t strideM,
                const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2, int64_t strideV2,
                const void* sigma, int64_t stride_sigma, int scale_right, void* right, int64_t ldr, int64_t strideR,
                void* left, int64_t ldl, int64_t strideL, const int32_t* rows32, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_project: bad dtype %d", dtype);
  TTR_REQUIRE(R >= 1 && n >= 1 && ro >= 1 && batch >= 0, TTR_E_INVALID, "ttr_project: bad shape");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(M && V2 && right, TTR_E_INVALID, "ttr_project: null pointer");
  TTR_REQUIRE(!scale_right || sigma, TTR_E_INVALID, "ttr_project: scale_right needs sigma");
  return sweep_project_dispatch(dtype, R, n, ro, batch, M, ldm, strideM, V1, ldv1, strideV1, V2, ldv2, strideV2, sigma,
                                stride_sigma, scale_right, right, ldr, strideR, left, ldl, strideL, (hipStream_t)stream, rows32);
}

int64_t ttr_colgram_workspace_bytes(int dtype, int64_t rows, int64_t n, int64_t batch) {
  if (rows <= 0 || n <= 0 || batch <= 0) return 0;
  return colgram_workspace_bytes(dtype, rows, n, batch);
}

int ttr_colgram(int dtype, int64_t rows, int64_t n, int64_t batch, const void* M, int64_t ldm, int64_t strideM,
                const void* V1, int64_t ldv1, int64_t strideV1, void* G, void* workspace, int64_t workspace_bytes,
                const int32_t* skip, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_colgram: bad dtype %d", dtype);
  TTR_REQUIRE(rows >= 1 && n >= 1 && batch >= 0, TTR_E_INVALID, "ttr_colgram: bad shape");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(M && G, TTR_E_INVALID, "ttr_colgram: null pointer");
  return colgram_dispatch(dtype, rows, n, batch, M, ldm, strideM, V1, ldv1, strideV1, G, workspace, workspace_bytes,
                          (hipStream_t)stream, skip);
}

int ttr_colproject(int dtype, int64_t rows, int64_t n, int64_t ro, int64_t batch, const void* M, int64_t ldm,
                   int64_t strideM, const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2,
                   int64_t strideV2, const void* sigma, int64_t stride_sigma, int left_ortho, void* left, int64_t ldl,
                   int64_t strideL, void* right, int64_t ldr, int64_t strideR, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_colproject: bad dtype %d", dtype);
  TTR_REQUIRE(rows >= 1 && n >= 1 && ro >= 1 && batch >= 0, TTR_E_INVALID, "ttr_colproject: bad shape");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(M && V2 && left, TTR_E_INVALID, "ttr_colproject: null pointer");
  TTR_REQUIRE(!left_ortho || sigma, TTR_E_INVALID, "ttr_colproject: left_ortho needs sigma");
  return colproject_dispatch(dtype, rows, n, ro, batch, M, ldm, strideM, V1, ldv1, strideV1, V2, ldv2, strideV2, sigma,
                             stride_sigma, left_ortho, left, ldl, strideL, right, ldr, strideR, (hipStream_t)stream);
}

int ttr_pow2_normalize(int dtype, int64_t count, int64_t batch, const void* x, int64_t stride_x, void* out,
                       int64_t stride_out, int32_t* e_out, int32_t* expo_acc, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_pow2_normalize: bad dtype %d", dtype);
  TTR_REQUIRE(count >= 0 && batch >= 0, TTR_E_INVALID, "ttr_pow2_normalize: negative size");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(x && (out || e_out), TTR_E_INVALID, "ttr_pow2_normalize: null pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(TTR_PROF_MISC, s);
  if (dtype == TTR_F32)
    hipLaunchKernelGGL(pow2_normalize_kernel<float>, dim3((unsigned)batch), dim3(kThreads), 0, s, (const float*)x, count,
                       stride_x, (float*)out, stride_out, e_out, expo_acc);
  else
    hipLaunchKernelGGL(pow2_normalize_kernel<double>, dim3((unsigned)batch), dim3(kThreads), 0, s, (const double*)x,
                       count, stride_x, (double*)out, stride_out, e_out, expo_acc);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int ttr_scale_batch(int dtype, int64_t count, int64_t batch, const void* x, int64_t stride_x, const void* scale,
                    int64_t stride_scale, const int32_t* expo, int expo_sign, void* out, int64_t stride_out,
                    void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_scale_batch: bad dtype %d", dtype);
  TTR_REQUIRE(count >= 0 && batch >= 0, TTR_E_INVALID, "ttr_scale_batch: negative size");
  if (batch == 0 || count == 0) return TTR_OK;
  TTR_REQUIRE(x && out, TTR_E_INVALID, "ttr_scale_batch: null pointer");
  hipStream_t s = (hipStream_t)stream;
  int64_t gx = ceil_div(count, kThreads * 4);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  ProfScope prof(TTR_PROF_MISC, s);
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {  // gridDim.y limit
    const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
    const int64_t so = dtype == TTR_F32 ? 4 : 8;
    const char* xs = (const char*)x + b0 * stride_x * so;
    char* os = (char*)out + b0 * stride_out * so;
    const char* ss = scale ? (const char*)scale + b0 * stride_scale * so : nullptr;
    const int32_t* es = expo ? expo + b0 : nullptr;
    if (dtype == TTR_F32)
      hipLaunchKernelGGL(scale_batch_kernel<float>, dim3((unsigned)gx, (unsigned)nb), dim3(kThreads), 0, s, (const float*)xs,
                         count, stride_x, (const float*)ss, stride_scale, es, expo_sign, (float*)os, stride_out);
    else
      hipLaunchKernelGGL(scale_batch_kernel<double>, dim3((unsigned)gx, (unsigned)nb), dim3(kThreads), 0, s,
                         (const double*)xs, count, stride_x, (const double*)ss, stride_scale, es, expo_sign, (double*)os,
                         stride_out);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}


int64_t ttr_orth_fixup_workspace_bytes(int dtype, int64_t r, int64_t n, int64_t batch, int64_t elem_stride) {
  if (!dtype_ok(dtype) || !orth_split_ok(r, n, batch, elem_stride)) return 0;
  return orth_split_layout(r, n, batch, dtype == TTR_F32 ? 4 : 8, g_orth_rounds).total;
}

int ttr_orth_fixup(int dtype, int64_t r, int64_t n, int64_t batch, void* X, int64_t vec_stride, int64_t elem_stride,
                   int64_t strideX, const void* sigma, int64_t stride_sigma, double dead_rel, const int32_t* rank_dev,
                   void* workspace, int64_t workspace_bytes, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_orth_fixup: bad dtype %d", dtype);
  TTR_REQUIRE(r >= 0 && n >= 0 && batch >= 0 && r <= 2147483647LL, TTR_E_INVALID, "ttr_orth_fixup: bad sizes");
  if (batch == 0 || r == 0 || n == 0) return TTR_OK;
  TTR_REQUIRE(X && sigma, TTR_E_INVALID, "ttr_orth_fixup: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int32_t* of_skip = nullptr;   // (set when the three-launch rounds ran first: what they left over goes to the single launch)
  int of_round0 = 0;
  if (workspace && orth_split_ok(r, n, batch, elem_stride) &&
      workspace_bytes >= orth_split_layout(r, n, batch, dtype == TTR_F32 ? 4 : 8, g_orth_rounds).total) {
    const int rc = dtype == TTR_F32
                       ? orth_split_run<float>(r, n, batch, (float*)X, vec_stride, strideX, (const float*)sigma, stride_sigma, dead_rel,
                                               rank_dev, (char*)workspace, s)
                       : orth_split_run<double>(r, n, batch, (double*)X, vec_stride, strideX, (const double*)sigma, stride_sigma, dead_rel,
                                                rank_dev, (char*)workspace, s);
    if (rc != TTR_OK) return rc;
    of_round0 = g_orth_rounds < kOrthSplitRounds ? g_orth_rounds : kOrthSplitRounds;
    if (of_round0 >= g_orth_rounds) return TTR_OK;
    of_skip = (const int32_t*)((char*)workspace + orth_split_layout(r, n, batch, dtype == TTR_F32 ? 4 : 8, g_orth_rounds).off_skip) +
              (int64_t)of_round0 * batch;
  }
  ProfScope prof(TTR_PROF_MISC, s);
  if (r <= 64) {  // the block variant (Gram matrix + coefficient-space Gram-Schmidt + one small product per round)
    const int cw = r <= 32 ? 256 : 128;
    const size_t lds = orth_fixup_lds_bytes((int)r, cw, dtype == TTR_F32 ? 4 : 8);
#define TTR_OF_LAUNCH(T_, CW_, V2_)                                                                                             \
    do {                                                                                                                          \
      auto kern = orth_fixup_block_kernel<T_, CW_, V2_>;                                                                          \
      if (lds > 64 * 1024) TTR_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(kThreads), lds, s, (int)r, n, (T_*)X, vec_stride, elem_stride, strideX, \
                         (const T_*)sigma, stride_sigma, dead_rel, rank_dev, g_orth_rounds, work_census_on() ? g_work_dev : nullptr, \
                         g_qr_dbg_bx == -1 ? g_qr_dbg : nullptr, of_skip, of_round0);                                             \
    } while (0)
    if (dtype == TTR_F32) {
      if (cw == 256 && g_orth_v2) TTR_OF_LAUNCH(float, 256, true);
      else if (cw == 256) TTR_OF_LAUNCH(float, 256, false);
      else TTR_OF_LAUNCH(float, 128, false);
    } else {   // (fp64: round 4's loops -- the V2 instance would spill at two waves per SIMD)
      if (cw == 256) TTR_OF_LAUNCH(double, 256, false);
      else TTR_OF_LAUNCH(double, 128, false);
    }
#undef TTR_OF_LAUNCH
  } else if (dtype == TTR_F32)
    hipLaunchKernelGGL(orth_fixup_kernel<float>, dim3((unsigned)batch), dim3(kThreads), 0, s, (int)r, n, (float*)X,
                       vec_stride, elem_stride, strideX, (const float*)sigma, stride_sigma, dead_rel, rank_dev);
  else
    hipLaunchKernelGGL(orth_fixup_kernel<double>, dim3((unsigned)batch), dim3(kThreads), 0, s, (int)r, n, (double*)X,
                       vec_stride, elem_stride, strideX, (const double*)sigma, stride_sigma, dead_rel, rank_dev);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int ttr_krp_contract(int dtype, int64_t P, int64_t J, int64_t Q, int64_t R, const void* T, const void* B, int64_t ldb,
                     void* out, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_krp_contract: bad dtype %d", dtype);
  TTR_REQUIRE(P >= 0 && J >= 1 && Q >= 0 && R >= 1 && ldb >= R, TTR_E_INVALID, "ttr_krp_contract: bad sizes");
  if (P == 0 || Q == 0) return TTR_OK;
  TTR_REQUIRE(T && B && out, TTR_E_INVALID, "ttr_krp_contract: null pointer");
  return krp_contract_dispatch(dtype, P, J, Q, R, T, B, ldb, out, (hipStream_t)stream);
}

int ttr_hadamard(int dtype, int64_t count, const void* a, const void* b, void* out, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_hadamard: bad dtype %d", dtype);
  TTR_REQUIRE(count >= 0, TTR_E_INVALID, "ttr_hadamard: negative count");
  if (count == 0) return TTR_OK;
  TTR_REQUIRE(a && b && out, TTR_E_INVALID, "ttr_hadamard: null pointer");
  return hadamard_dispatch(dtype, count, a, b, out, (hipStream_t)stream);
}

int ttr_core_kron(int dtype, int64_t batch, int64_t R1, int64_t S1, int64_t I, int64_t R2, int64_t S2, const void* a,
                  const void* c, void* out, void* stream) {
  TTR_REQUIRE(dtype_ok(dtype), TTR_E_INVALID, "ttr_core_kron: bad dtype %d", dtype);
  TTR_REQUIRE(batch >= 0 && R1 >= 1 && S1 >= 1 && I >= 1 && R2 >= 1 && S2 >= 1, TTR_E_INVALID, "ttr_core_kron: bad sizes");
  if (batch == 0) return TTR_OK;
  TTR_REQUIRE(a && c && out, TTR_E_INVALID, "ttr_core_kron: null pointer");
  return core_kron_dispatch(dtype, batch, R1, S1, I, R2, S2, a, c, out, (hipStream_t)stream);
}

int ttr_debug_set_qr_stamps(void* device_buffer) {
  g_qr_dbg = (long long*)device_buffer;
  return TTR_OK;
}

int ttr_debug_set_knob(int knob, int value) {
  switch (knob) {
    case TTR_KNOB_QR_STAMP_BX:
      g_qr_dbg_bx = value;
      return TTR_OK;
    case TTR_KNOB_QR_STAMP_BY:
      g_qr_dbg_by = value;
      return TTR_OK;
    case TTR_KNOB_EIGH_SMALL:
      TTR_REQUIRE(value >= 0 && value <= 3, TTR_E_INVALID, "ttr_debug_set_knob: small-eigensolver switch %d outside [0, 3]", value);
      g_eigh_small = value;
      return TTR_OK;
    case TTR_KNOB_ORTH_SPLIT:
      TTR_REQUIRE(value >= 0 && value <= 65535, TTR_E_INVALID, "ttr_debug_set_knob: orth split threshold %d outside [0, 65535]", value);
      g_orth_split = value;
      return TTR_OK;
    case TTR_KNOB_SWEEP_STAGGER:
      TTR_REQUIRE(value >= 0 && value <= 2, TTR_E_INVALID, "ttr_debug_set_knob: stagger mode %d outside [0, 2]", value);
      g_sweep_stagger = value;
      return TTR_OK;
    case TTR_KNOB_EIGH_BIG_OCC:
      TTR_REQUIRE(value == 0 || value == 2 || value == 3, TTR_E_INVALID, "ttr_debug_set_knob: eigensolver occupancy %d not in {0, 2, 3}", value);
      g_eigh_big_occ = value;
      return TTR_OK;
    case TTR_KNOB_QR_PACK_PRE:
      TTR_REQUIRE(value >= 0 && value <= 3, TTR_E_INVALID, "ttr_debug_set_knob: pack-flag switch %d outside [0, 3]", value);
      g_qr_pack_pre = value & 1;
      g_qr_l1_idle = (value & 2) ? 0 : 1;
      return TTR_OK;
    case TTR_KNOB_QR_STAGGER:
      TTR_REQUIRE(value >= 0 && value <= 1024, TTR_E_INVALID, "ttr_debug_set_knob: stagger of %d kilo-cycles outside [0, 1024]", value);
      g_qr_stagger = value;
      return TTR_OK;
    case TTR_KNOB_QR_INTERLEAVE:
      TTR_REQUIRE(value >= 0 && value <= 1, TTR_E_INVALID, "ttr_debug_set_knob: interleave switch %d outside [0, 1]", value);
      g_qr_interleave = value;
      return TTR_OK;
    case TTR_KNOB_ORTH_V2:
      TTR_REQUIRE(value >= 0 && value <= 2, TTR_E_INVALID, "ttr_debug_set_knob: orth_fixup variant %d outside [0, 2]", value);
      g_orth_v2 = value;
      return TTR_OK;
    case TTR_KNOB_JACOBI_LIVE_WAVE:
      TTR_REQUIRE(value >= 0 && value <= 1, TTR_E_INVALID, "ttr_debug_set_knob: pass-2 Jacobi switch %d outside [0, 1]", value);
      g_jacobi_live_wave = value;
      return TTR_OK;
    case TTR_KNOB_ORTH_ROUNDS:
      TTR_REQUIRE(value >= 1 && value <= 4, TTR_E_INVALID, "ttr_debug_set_knob: %d rounds outside [1, 4]", value);
      g_orth_rounds = value;
      return TTR_OK;
    case TTR_KNOB_RANK_NOISE_FLOOR:
      TTR_REQUIRE(value >= 0 && value <= 1024, TTR_E_INVALID, "ttr_debug_set_knob: rank-rule noise floor %d outside [0, 1024]", value);
      g_rank_noise_c = value;
      return TTR_OK;
    case TTR_KNOB_QR_PACK:
      TTR_REQUIRE(value >= 0 && value <= 3, TTR_E_INVALID, "ttr_debug_set_knob: packing switch %d outside [0, 3]", value);
      g_qr_pack = value;
      return TTR_OK;
    case TTR_KNOB_QR_RANK_SKIP:
      TTR_REQUIRE(value >= 0 && value <= 4096, TTR_E_INVALID, "ttr_debug_set_knob: rank-skip factor %d outside [0, 4096]", value);
      g_rank_skip_c = value;
      return TTR_OK;
    case TTR_KNOB_QR_F64_NW4:
      TTR_REQUIRE(value >= 0 && value <= 7, TTR_E_INVALID, "ttr_debug_set_knob: 4-wave block switch %d outside [0, 7]", value);
      g_qr_f64_nw4 = value;
      return TTR_OK;
    case TTR_KNOB_GEMM_BIG:
      TTR_REQUIRE(value >= 0 && value <= 1, TTR_E_INVALID, "ttr_debug_set_knob: big-tile GEMM switch %d outside [0, 1]", value);
      g_gemm_big = value;
      return TTR_OK;
    case TTR_KNOB_BJ_INNER_SWEEPS:
      TTR_REQUIRE(value >= 0 && value <= 64, TTR_E_INVALID, "ttr_debug_set_knob: %d inner sweeps outside [0, 64]", value);
      g_bj_inner_sweeps = value;
      return TTR_OK;
    case TTR_KNOB_QR_PANEL:
      TTR_REQUIRE(value >= 0 && value <= 1, TTR_E_INVALID, "ttr_debug_set_knob: QR panel variant %d outside [0, 1]", value);
      g_qr_variant = value;
      return TTR_OK;
    default:
      TTR_REQUIRE(false, TTR_E_INVALID, "ttr_debug_set_knob: unknown knob %d", knob);
  }
}

int ttr_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  g_prof_level = on;
  if (on >= 2) {   // census mode: per-kind device counters (the only allocation this library ever makes; profiling runs only)
    if (!g_work_dev) TTR_HIP_CHECK(hipMalloc((void**)&g_work_dev, 2 * TTR_PROF_NKINDS * sizeof(double)));
    TTR_HIP_CHECK(hipMemset(g_work_dev, 0, 2 * TTR_PROF_NKINDS * sizeof(double)));
  }
  return TTR_OK;
}

int ttr_prof_collect(double* ms, int64_t* launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int k = 0; k < TTR_PROF_NKINDS; ++k) {
    if (ms) ms[k] = 0.0;
    if (launches) launches[k] = 0;
  }
  for (auto& r : g_prof_recs) {
    if (hipEventSynchronize(r.stop) == hipSuccess) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.start, r.stop) == hipSuccess && r.kind >= 0 && r.kind < TTR_PROF_NKINDS) {
        if (ms) ms[r.kind] += (double)t;
        if (launches) launches[r.kind] += 1;
      }
    }
    g_prof_pool.push_back({r.start, r.stop});
  }
  g_prof_recs.clear();
  return TTR_OK;
}

int ttr_prof_collect_work(double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double host[2 * TTR_PROF_NKINDS] = {0};
  if (g_work_dev) {
    TTR_HIP_CHECK(hipDeviceSynchronize());
    TTR_HIP_CHECK(hipMemcpy(host, g_work_dev, sizeof(host), hipMemcpyDeviceToHost));
    TTR_HIP_CHECK(hipMemset(g_work_dev, 0, sizeof(host)));
  }
  for (int k = 0; k < TTR_PROF_NKINDS; ++k) {
    if (flops) flops[k] = host[k];
    if (bytes) bytes[k] = host[TTR_PROF_NKINDS + k];
  }
  return TTR_OK;
}

}  // extern "C"
