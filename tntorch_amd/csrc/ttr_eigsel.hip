// Selected eigenpairs of symmetric matrices above the single-workgroup eigensolvers (gfx950): the k LARGEST eigenvalues and
// their eigenvectors of G (n x n, 64 < n <= 1024), one workgroup per matrix.
//
// Where it is used: the Gram matrix of a dense TT-SVD bond (round.py:104-115 at n = I r: BASELINE config C3 has n = 256,
// rank cap 8) in batch mode, where the rank is the cap and only the top of the spectrum is ever looked at.  The block-Jacobi
// driver diagonalises the WHOLE matrix to the rounding floor (~12 sweeps x 7 rounds x 2 launches for n = 256); the classical
// reduction does 4/3 n^3 flops once and then works on a tridiagonal matrix (LAPACK syevx class):
//   ttr_tridiag        Householder reduction G = Q T Q^T, Q = H_0 ... H_{n-2}: per step a symmetric matrix-vector product and a
//                      rank-2 update of the trailing block, which stays in global memory (L2-resident: 256 KB per matrix at
//                      n = 256) and is walked row-wise by 8 waves (lanes across the columns: coalesced; the full square is kept
//                      symmetric, so the reflector's source column is read as a ROW).  Reflector k overwrites row k (columns
//                      k+1 ..), d / e / tau go to side arrays.
//   ttr_tri_eigsel     the k largest eigenvalues of T by Sturm-count MULTISECTION (one wave: 64 / k shifts per eigenvalue and
//                      round, every lane runs the n-step count recurrence for its own shift) and their eigenvectors from the
//                      twisted factorisation of T - lambda I (one lane per vector: stationary qd transforms from both ends, the
//                      twist index where |gamma| is smallest -- the getvec kernel of MRRR), normalised.
//   ttr_tridiag_back   X = Q Z: the reflectors applied to the k columns in reverse order, each column held in the REGISTERS of
//                      one wave (lane = row mod 64): no barriers.
// The host shim orthonormalises Z with the TSQR kernel between the last two (close eigenvalues leave the twisted vectors
// only nearly orthogonal) and checks the diagonal of that R for collapsed columns (clusters / multiple eigenvalues: the caller
// falls back to the block-Jacobi driver).
#include "ttr_common.h"

namespace ttr {

constexpr int kTdThreads = 1024;  // 16 waves
constexpr int kTdWaves = kTdThreads / kWave;
constexpr int UR = 8;            // rows of the trailing block a wave handles per trip

template <typename T>
struct TridiagArgs {
  int n;
  T* A;  // [batch][n][lda] symmetric, overwritten: row k, columns k+1 .. hold reflector k (leading 1 stored)
  int64_t lda, strideA;
  T* d;    // [batch][n]
  T* e;    // [batch][n]  e[k] couples k, k+1; e[n-1] = 0
  T* tau;  // [batch][n]
};

// sum over the 512 threads, result in every thread (red: kTdWaves elements; the leading barrier protects its reuse)
template <typename T>
__device__ __forceinline__ T td_block_sum(T v, T* red) {
  v = wave_sum_dpp(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  T s = T(0);
#pragma unroll
  for (int w = 0; w < kTdWaves; ++w) s += red[w];
  return s;
}

// One pass over the trailing block per step: the rank-2 update of step k and the matrix-vector product of step k + 1 are
// fused -- row k + 1 is updated first (one round trip), the next reflector is built from it, and the pass that updates the
// rest multiplies every updated element with that reflector on the fly.  A separate product pass doubled the L2 round trips:
// the step is bound by their latency (one workgroup per matrix; 1.94 ms for n = 256 unfused, measured), not by bytes.
template <typename T>
__global__ __launch_bounds__(kTdThreads) void tridiag_big_kernel(TridiagArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = p.n;
  T* buf = reinterpret_cast<T*>(smem_raw);  // v / w of the current step and of the next one: [4][n]
  T* red = buf + 4 * n;                     // [kTdWaves + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t lda = p.lda;
  T* __restrict__ A = p.A + (int64_t)blockIdx.x * p.strideA;
  T* __restrict__ dd = p.d + (int64_t)blockIdx.x * n;
  T* __restrict__ ee = p.e + (int64_t)blockIdx.x * n;
  T* __restrict__ tt = p.tau + (int64_t)blockIdx.x * n;
  // reflector of step k from x (thread t holds x[t], t < m): returns tau, writes v to vdst[0 .. m) and to row k of A, e[k], tau[k]
  auto build = [&](int k, int m, T x, T* vdst) -> T {
    if (tid == 0) red[kTdWaves] = x;
    const T ss = td_block_sum((tid >= 1 && tid < m) ? x * x : T(0), red);
    const T alpha = red[kTdWaves];
    T beta = alpha, tau = T(0), scale = T(0);
    if (ss != T(0)) {  // LAPACK larfg
      beta = -copysign(sqrt(alpha * alpha + ss), alpha);
      tau = (beta - alpha) / beta;
      scale = T(1) / (alpha - beta);
    }
    const T v = tid == 0 ? T(1) : x * scale;
    if (tid < m) {
      vdst[tid] = v;
      A[(int64_t)k * lda + (k + 1) + tid] = v;
    }
    if (tid == 0) { ee[k] = beta; tt[k] = tau; }
    return tau;
  };
  // w = p - tau/2 (p^T v) v from p = tau A v in wdst[0 .. m)
  auto finish_w = [&](int m, T tau, const T* vsrc, T* wdst) {
    __syncthreads();  // p complete
    const T pv = tid < m ? wdst[tid] : T(0);
    const T dot = td_block_sum(tid < m ? pv * vsrc[tid] : T(0), red);
    if (tid < m) wdst[tid] = pv - T(0.5) * tau * dot * vsrc[tid];
    __syncthreads();
  };
  T* vs = buf; T* ws = buf + n; T* vn = buf + 2 * n; T* wn = buf + 3 * n;
  if (n == 1) {
    if (tid == 0) { dd[0] = A[0]; ee[0] = T(0); tt[0] = T(0); }
    return;
  }
  // ---- step 0: reflector from row 0, plain product pass
  {
    const int m = n - 1;
    if (tid == 0) dd[0] = A[0];
    const T tau = build(0, m, tid < m ? A[1 + tid] : T(0), vs);
    __syncthreads();
    T* A22 = A + lda + 1;
    for (int r0 = wave * UR; r0 < m; r0 += kTdWaves * UR) {
      T acc[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) acc[u] = T(0);
      for (int c = lane; c < m; c += kWave) {
        const T vc = vs[c];
        T a[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) a[u] = (r0 + u < m) ? A22[(int64_t)(r0 + u) * lda + c] : T(0);
#pragma unroll
        for (int u = 0; u < UR; ++u) acc[u] += a[u] * vc;
      }
#pragma unroll
      for (int u = 0; u < UR; u += 4) {
        T q4[4] = {acc[u], acc[u + 1], acc[u + 2], acc[u + 3]};
        wave_sum4(q4);
        if (lane == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (r0 + u + t < m) ws[r0 + u + t] = tau * q4[t];
        }
      }
    }
    finish_w(m, tau, vs, ws);
  }
  // ---- steps 1 .. n-2: (v, w) of step k - 1 in vs / ws, indexed from row / column k
  for (int k = 1; k + 1 < n; ++k) {
    const int mo = n - k;      // size of the block (v, w) act on: rows / columns k .. n-1
    const int m = mo - 1;      // length of the new reflector: rows k+1 .. n-1
    T* Ao = A + (int64_t)k * lda + k;  // that block
    // row k (local row 0) after the update = column k of the updated matrix: diagonal d[k], x = its columns k+1 ..
    const T v0 = vs[0], w0 = ws[0];
    if (tid == 0) dd[k] = Ao[0] - T(2) * v0 * w0;
    const T x = tid < m ? Ao[1 + tid] - (v0 * ws[1 + tid] + w0 * vs[1 + tid]) : T(0);
    const T tau = build(k, m, x, vn);
    __syncthreads();
    // fused pass over local rows / columns 1 .. mo-1: a <- a - (v_r w_c + w_r v_c), p_r += a vn[c - 1]
    T* A22 = Ao + lda + 1;
    for (int r0 = wave * UR; r0 < m; r0 += kTdWaves * UR) {
      T acc[UR], vr[UR], wr[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        acc[u] = T(0);
        const int r = r0 + u < m ? r0 + u : m - 1;
        vr[u] = vs[1 + r]; wr[u] = ws[1 + r];
      }
      for (int c = lane; c < m; c += kWave) {
        const T vc = vs[1 + c], wc = ws[1 + c], nc = vn[c];
        T a[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) a[u] = (r0 + u < m) ? A22[(int64_t)(r0 + u) * lda + c] : T(0);
#pragma unroll
        for (int u = 0; u < UR; ++u) {
          a[u] -= vr[u] * wc + wr[u] * vc;
          if (r0 + u < m) A22[(int64_t)(r0 + u) * lda + c] = a[u];
          acc[u] += a[u] * nc;
        }
      }
#pragma unroll
      for (int u = 0; u < UR; u += 4) {
        T q4[4] = {acc[u], acc[u + 1], acc[u + 2], acc[u + 3]};
        wave_sum4(q4);
        if (lane == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (r0 + u + t < m) wn[r0 + u + t] = tau * q4[t];
        }
      }
    }
    finish_w(m, tau, vn, wn);  // (its barriers also drain the global stores of the pass)
    T* t0 = vs; vs = vn; vn = t0;
    t0 = ws; ws = wn; wn = t0;
  }
  // the last diagonal entry: block of size 1 under (v, w) of step n - 2 (v = [1], w = [w0])
  if (tid == 0) {
    dd[n - 1] = A[(int64_t)(n - 1) * lda + (n - 1)] - T(2) * vs[0] * ws[0];
    ee[n - 1] = T(0);
    tt[n - 1] = T(0);
  }
}

// ---------------------------------------------------------------- the same reduction spread over the chip (few big matrices)
// One workgroup per matrix leaves 255 of the 256 CUs idle when there is ONE matrix (config C1: three n = 1024 bonds, ~30 ms each
// on a single CU).  The fused pass is row- and column-parallel, so here it is a launch of its own over (8-row group) x (256-column
// chunk) workgroups of one wave each, and what happens between two passes -- w of the previous step from the chunks' partial
// products, the next reflector from the updated row -- is a one-workgroup launch: 2 n - 1 launches per matrix, no device-side grid
// synchronisation.  State between launches: v / w of two consecutive steps and the partial products, 8 n elements per matrix.
template <typename T>
struct TdmArgs {
  int n, k;
  T* A;
  int64_t lda, strideA;
  T* d;
  T* e;
  T* tau;
  T* ws;  // [batch][8][n]: V[2][n], W[2][n] (by step parity), PP[4][n]
};
constexpr int kTdmCols = 256;  // columns per pass workgroup

// between two passes (one workgroup per matrix).  k = 0: first reflector from the raw row 0.  1 <= k <= n-2: w_{k-1} from the
// partial products of pass k - 1, then reflector k from the updated row k.  k = n-1: w_{n-2}, last diagonal entry.
template <typename T>
__global__ __launch_bounds__(kTdThreads) void tdm_mid_kernel(TdmArgs<T> p) {
  __shared__ T red[kTdWaves + 1];
  const int n = p.n, k = p.k, tid = threadIdx.x;
  const int64_t lda = p.lda;
  T* __restrict__ A = p.A + (int64_t)blockIdx.x * p.strideA;
  T* __restrict__ dd = p.d + (int64_t)blockIdx.x * n;
  T* __restrict__ ee = p.e + (int64_t)blockIdx.x * n;
  T* __restrict__ tt = p.tau + (int64_t)blockIdx.x * n;
  T* __restrict__ W0 = p.ws + (int64_t)blockIdx.x * 8 * n;
  T* Vp = W0 + ((k - 1) & 1) * n;          // v_{k-1}, index 0 <-> row k
  T* Wp = W0 + (2 + ((k - 1) & 1)) * n;    // w_{k-1}
  T* Vk = W0 + (k & 1) * n;                // v_k, index 0 <-> row k + 1
  const T* PP = W0 + 4 * n;
  T v0 = T(0), w0 = T(0), vmine = T(0), wmine = T(0);  // (v, w)_{k-1} at index 0 and at index 1 + tid
  if (k >= 1) {
    const int mo = n - k;  // support of v_{k-1}
    const int ncc = (mo + kTdmCols - 1) / kTdmCols;
    T pv = T(0);
    if (tid < mo)
      for (int cc = 0; cc < ncc; ++cc) pv += PP[cc * n + tid];
    const T taup = tt[k - 1];
    const T vt = tid < mo ? Vp[tid] : T(0);
    const T dot = td_block_sum(pv * vt, red);
    const T wt = pv - T(0.5) * taup * dot * vt;
    if (tid < mo) Wp[tid] = wt;
    if (tid == 0) { red[kTdWaves] = wt; }
    __syncthreads();
    w0 = red[kTdWaves];
    v0 = T(1);  // (the leading component of every reflector)
    __syncthreads();
    // (v, w)_{k-1} at index 1 + tid: the neighbour's values, through global memory (written above by this workgroup)
    if (tid + 1 < mo) { vmine = Vp[tid + 1]; wmine = Wp[tid + 1]; }
  }
  if (k == n - 1) {
    if (tid == 0) { dd[k] = A[(int64_t)k * lda + k] - T(2) * v0 * w0; ee[k] = T(0); tt[k] = T(0); }
    return;
  }
  const int m = n - k - 1;
  if (tid == 0) dd[k] = A[(int64_t)k * lda + k] - T(2) * v0 * w0;
  const T x = tid < m ? A[(int64_t)k * lda + (k + 1) + tid] - (v0 * wmine + w0 * vmine) : T(0);
  if (tid == 0) red[kTdWaves] = x;
  const T ss = td_block_sum((tid >= 1 && tid < m) ? x * x : T(0), red);
  const T alpha = red[kTdWaves];
  T beta = alpha, tau = T(0), scale = T(0);
  if (ss != T(0)) {
    beta = -copysign(sqrt(alpha * alpha + ss), alpha);
    tau = (beta - alpha) / beta;
    scale = T(1) / (alpha - beta);
  }
  const T v = tid == 0 ? T(1) : x * scale;
  if (tid < m) {
    Vk[tid] = v;
    A[(int64_t)k * lda + (k + 1) + tid] = v;
  }
  if (tid == 0) { ee[k] = beta; tt[k] = tau; }
}

// pass k over the trailing block (rows / columns k+1 ..): a <- a - (v_r w_c + w_r v_c) with (v, w)_{k-1} (k >= 1), and the chunk's
// share of tau_k A v_k.  One wave per workgroup: UR rows x kTdmCols columns.
template <typename T>
__global__ __launch_bounds__(kWave) void tdm_pass_kernel(TdmArgs<T> p) {
  const int n = p.n, k = p.k, lane = threadIdx.x;
  const int m = n - k - 1;
  const int ncc = (m + kTdmCols - 1) / kTdmCols;
  const int rg = blockIdx.x / ncc, cc = blockIdx.x - rg * ncc;
  const int r0 = rg * UR, c0 = cc * kTdmCols;
  const int64_t lda = p.lda;
  T* __restrict__ A22 = p.A + (int64_t)blockIdx.y * p.strideA + (int64_t)(k + 1) * lda + (k + 1);
  T* __restrict__ W0 = p.ws + (int64_t)blockIdx.y * 8 * n;
  const T* __restrict__ Vp = W0 + ((k - 1) & 1) * n;
  const T* __restrict__ Wp = W0 + (2 + ((k - 1) & 1)) * n;
  const T* __restrict__ Vk = W0 + (k & 1) * n;
  T* __restrict__ PP = W0 + 4 * n;
  const T tau = p.tau[(int64_t)blockIdx.y * n + k];
  const bool upd = k >= 1;
  T acc[UR], vr[UR], wr[UR];
#pragma unroll
  for (int u = 0; u < UR; ++u) {
    acc[u] = T(0);
    const int r = r0 + u < m ? r0 + u : m - 1;
    vr[u] = upd ? Vp[1 + r] : T(0);
    wr[u] = upd ? Wp[1 + r] : T(0);
  }
  for (int c = c0 + lane; c < m && c < c0 + kTdmCols; c += kWave) {
    const T vc = upd ? Vp[1 + c] : T(0), wc = upd ? Wp[1 + c] : T(0), nc = Vk[c];
    T a[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) a[u] = (r0 + u < m) ? A22[(int64_t)(r0 + u) * lda + c] : T(0);
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (upd) {
        a[u] -= vr[u] * wc + wr[u] * vc;
        if (r0 + u < m) A22[(int64_t)(r0 + u) * lda + c] = a[u];
      }
      acc[u] += a[u] * nc;
    }
  }
#pragma unroll
  for (int u = 0; u < UR; u += 4) {
    T q4[4] = {acc[u], acc[u + 1], acc[u + 2], acc[u + 3]};
    wave_sum4(q4);
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (r0 + u + t < m) PP[cc * n + r0 + u + t] = tau * q4[t];
    }
  }
}

// ---------------------------------------------------------------- k largest eigenpairs of the tridiagonal matrix
template <typename T>
struct EigSelArgs {
  int n, k;
  const T* d;
  const T* e;
  T* lam;      // [batch][k] descending
  T* Z;        // [batch][n][k]
  T* scratch;  // [batch][3][n][64]: D+, L, U of the twisted factorisations, lane-interleaved
};

template <typename T>
__global__ __launch_bounds__(kWave) void tri_eigsel_kernel(EigSelArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = p.n, k = p.k, lane = threadIdx.x;
  T* ds = reinterpret_cast<T*>(smem_raw);  // [n]
  T* es = ds + n;                          // [n]
  T* e2 = es + n;                          // [n]
  const T* __restrict__ dg = p.d + (int64_t)blockIdx.x * n;
  const T* __restrict__ eg = p.e + (int64_t)blockIdx.x * n;
  T gl = Num<T>::big_theta(), gu = -Num<T>::big_theta(), emax2 = T(0);
  for (int i = lane; i < n; i += kWave) {
    const T di = dg[i], ei = i + 1 < n ? eg[i] : T(0), em = i > 0 ? eg[i - 1] : T(0);
    ds[i] = di; es[i] = ei; e2[i] = ei * ei;
    const T rad = fabs(ei) + fabs(em);
    gl = fmin(gl, di - rad); gu = fmax(gu, di + rad); emax2 = fmax(emax2, ei * ei);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    gl = fmin(gl, __shfl_xor(gl, off, 64)); gu = fmax(gu, __shfl_xor(gu, off, 64)); emax2 = fmax(emax2, __shfl_xor(emax2, off, 64));
  }
  __syncthreads();
  const T eps = Num<T>::eps();
  const T tnorm = fmax(fabs(gl), fabs(gu));
  const T pivmin = Num<T>::tiny() * fmax(T(1), emax2) / eps;  // (LAPACK stebz: safemin * max(1, max e^2); / eps: the fast reciprocal flushes denormals)
  gl -= T(2.1) * tnorm * eps * n + T(2.1) * pivmin;
  gu += T(2.1) * tnorm * eps * n + T(2.1) * pivmin;
  // number of eigenvalues < sigma (Sturm count of the LDL^T pivots of T - sigma I)
  auto count_below = [&](T sigma) {
    T q = ds[0] - sigma;
    if (fabs(q) < pivmin) q = -pivmin;
    int cnt = q < T(0) ? 1 : 0;
    for (int i = 1; i < n; ++i) {
      q = ds[i] - sigma - e2[i - 1] / q;
      if (fabs(q) < pivmin) q = -pivmin;
      cnt += q < T(0) ? 1 : 0;
    }
    return cnt;
  };
  // ---- multisection: G lanes per eigenvalue (k rounded up to a power of two), every round cuts the bracket by G + 1
  int kp = 1;
  while (kp < k) kp <<= 1;
  const int G = kWave / kp;          // >= 1 (k <= 64)
  const int j = lane / G, s = lane % G;
  const bool live = j < k;
  const int idx = n - 1 - (live ? j : 0);  // ascending index of the j-th largest eigenvalue
  T lo = gl, hi = gu;
  const int max_rounds = sizeof(T) == 4 ? 48 : 110;
  for (int it = 0; it < max_rounds; ++it) {
    const T wdt = hi - lo;
    const bool done = !(wdt > T(2) * eps * fmax(fabs(lo), fabs(hi)) + T(2) * pivmin);
    if (__ballot(live && !done) == 0ull) break;
    const T sigma = lo + wdt * (T)(s + 1) / (T)(G + 1);
    const bool above = count_below(sigma) >= idx + 1;  // sigma lies above the eigenvalue
    const unsigned long long mask = __ballot(above);
    const unsigned long long grp = (mask >> (j * G)) & (G == 64 ? ~0ull : ((1ull << G) - 1ull));
    const int f = grp ? __ffsll((long long)grp) - 1 : G;  // first shift of the group that is above (G: none)
    const T nlo = f == 0 ? lo : lo + wdt * (T)f / (T)(G + 1);
    const T nhi = f == G ? hi : lo + wdt * (T)(f + 1) / (T)(G + 1);
    if (!done) { lo = nlo; hi = nhi; }
  }
  // eigenvalue j sits in the lanes of group j; vector work: one lane per eigenvalue (lane jv < k takes eigenvalue jv)
  const T lam_grp = T(0.5) * (lo + hi);
  const int jv = lane;
  const T lam = __shfl(lam_grp, (jv < k ? jv : 0) * G, 64);
  const bool vlive = jv < k;
  if (vlive) p.lam[(int64_t)blockIdx.x * k + jv] = lam;
  // ---- twisted factorisation of T - lam I: D+ / L from the top, D- / U from the bottom, twist where |gamma| is smallest
  T* __restrict__ sc = p.scratch + (int64_t)blockIdx.x * 3 * n * kWave;
  T* Dp = sc + lane;                      // [i * 64]
  T* Lw = sc + (int64_t)n * kWave + lane;
  T* Uw = sc + (int64_t)2 * n * kWave + lane;
  T q = ds[0] - lam;
  for (int i = 0; i + 1 < n; ++i) {
    if (fabs(q) < pivmin) q = -pivmin;
    const T l = es[i] / q;
    if (vlive) { Dp[(int64_t)i * kWave] = q; Lw[(int64_t)i * kWave] = l; }
    q = ds[i + 1] - lam - l * es[i];
  }
  if (vlive) Dp[(int64_t)(n - 1) * kWave] = q;
  T best = fabs(q);  // gamma_{n-1} = D+_{n-1}
  int r = n - 1;
  q = ds[n - 1] - lam;
  for (int i = n - 2; i >= 0; --i) {
    if (fabs(q) < pivmin) q = -pivmin;
    const T u = es[i] / q;
    if (vlive) Uw[(int64_t)i * kWave] = u;
    q = ds[i] - lam - u * es[i];  // D-_i
    const T gam = (vlive ? Dp[(int64_t)i * kWave] : T(0)) + q - (ds[i] - lam);
    if (fabs(gam) < best) { best = fabs(gam); r = i; }
  }
  // x_r = 1;  x_i = -L_i x_{i+1} (i < r);  x_{i+1} = -U_i x_i (i >= r)
  T* __restrict__ Zo = p.Z + (int64_t)blockIdx.x * n * k + jv;
  T nrm2 = T(1), xc = T(0);
  for (int i = 0; i + 1 < n; ++i) {  // downwards from the twist
    if (i == r) xc = T(1);
    if (vlive && i >= r) {
      if (i == r) Zo[(int64_t)i * k] = T(1);
      xc = -Uw[(int64_t)i * kWave] * xc;
      Zo[(int64_t)(i + 1) * k] = xc;
      nrm2 += xc * xc;
    }
  }
  if (vlive && r == n - 1) Zo[(int64_t)(n - 1) * k] = T(1);
  xc = T(0);
  for (int i = n - 2; i >= 0; --i) {  // upwards from the twist
    if (i + 1 == r) xc = T(1);
    if (vlive && i < r) {
      xc = -Lw[(int64_t)i * kWave] * xc;
      Zo[(int64_t)i * k] = xc;
      nrm2 += xc * xc;
    }
  }
  const T inv = T(1) / sqrt(nrm2);
  if (vlive)
    for (int i = 0; i < n; ++i) {
      T z = Zo[(int64_t)i * k] * inv;
      if (!(fabs(z) < Num<T>::big_theta())) z = T(0);  // (overflowed recurrences: the caller's collapse check sees the damage)
      Zo[(int64_t)i * k] = z;
    }
}

// ---------------------------------------------------------------- X = Q Z, Q = H_0 ... H_{n-2}
template <typename T>
struct BackArgs {
  int n, k;
  const T* A;  // reflectors as left by tridiag_big_kernel
  int64_t lda, strideA;
  const T* tau;  // [batch][n]
  T* Z;          // [batch][n][k] in place
};

template <typename T>
__global__ __launch_bounds__(kTdThreads) void tridiag_back_kernel(BackArgs<T> p) {
  constexpr int RPL = 1024 / kWave;  // rows per lane (n <= 1024)
  const int n = p.n, k = p.k, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* __restrict__ A = p.A + (int64_t)blockIdx.x * p.strideA;
  const T* __restrict__ tt = p.tau + (int64_t)blockIdx.x * n;
  T* __restrict__ Z = p.Z + (int64_t)blockIdx.x * n * k;
  for (int col = wave; col < k; col += kTdWaves) {  // a wave per column, the column in registers: row lane + 64 q
    T z[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int row = lane + kWave * q;
      z[q] = row < n ? Z[(int64_t)row * k + col] : T(0);
    }
    for (int kk = n - 2; kk >= 0; --kk) {
      const T tau = tt[kk];
      if (tau == T(0)) continue;  // (wave-uniform)
      const T* __restrict__ rowk = A + (int64_t)kk * p.lda;  // u[row] = rowk[row] for row > kk (u[kk + 1] = 1 stored)
      T u[RPL], dot = T(0);
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const int row = lane + kWave * q;
        u[q] = (row > kk && row < n) ? rowk[row] : T(0);
        dot += u[q] * z[q];
      }
      dot = wave_sum_dpp(dot) * tau;
#pragma unroll
      for (int q = 0; q < RPL; ++q) z[q] -= dot * u[q];
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int row = lane + kWave * q;
      if (row < n) Z[(int64_t)row * k + col] = z[q];
    }
  }
}

// ---------------------------------------------------------------- dispatch
int64_t eigsel_scratch_bytes(int dtype, int64_t n, int64_t batch) { return batch * 3 * n * kWave * (dtype == TTR_F64 ? 8 : 4); }
int eigsel_max_n() { return 1024; }

int64_t tridiag_workspace_bytes(int dtype, int64_t n, int64_t batch) { return batch * 8 * n * (dtype == TTR_F64 ? 8 : 4); }

// few big matrices: the multi-launch variant (one matrix per workgroup keeps at most `batch` CUs busy)
static bool tridiag_spread(int64_t n, int64_t batch) { return n >= 512 && batch <= 4; }

template <typename T>
static int tridiag_typed(int64_t n, int64_t batch, T* A, int64_t lda, int64_t strideA, T* d, T* e, T* tau, T* ws, hipStream_t stream) {
  ProfScope prof(TTR_PROF_EIGH, stream);
  if (ws && tridiag_spread(n, batch)) {
    TdmArgs<T> q;
    q.n = (int)n; q.A = A; q.lda = lda; q.strideA = strideA; q.d = d; q.e = e; q.tau = tau; q.ws = ws;
    for (int k = 0; k < (int)n; ++k) {
      q.k = k;
      hipLaunchKernelGGL(tdm_mid_kernel<T>, dim3((unsigned)batch), dim3(kTdThreads), 0, stream, q);
      if (k + 1 < (int)n) {
        const int m = (int)n - k - 1;
        const unsigned gx = (unsigned)(((m + UR - 1) / UR) * ((m + kTdmCols - 1) / kTdmCols));
        hipLaunchKernelGGL(tdm_pass_kernel<T>, dim3(gx, (unsigned)batch), dim3(kWave), 0, stream, q);
      }
    }
    TTR_HIP_CHECK(hipGetLastError());
    return TTR_OK;
  }
  TridiagArgs<T> p;
  p.n = (int)n; p.A = A; p.lda = lda; p.strideA = strideA; p.d = d; p.e = e; p.tau = tau;
  const size_t lds = ((4 * (size_t)n + kTdWaves + 1) * sizeof(T) + 15) & ~size_t(15);
  hipLaunchKernelGGL(tridiag_big_kernel<T>, dim3((unsigned)batch), dim3(kTdThreads), lds, stream, p);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}
int tridiag_dispatch(int dtype, int64_t n, int64_t batch, void* A, int64_t lda, int64_t strideA, void* d, void* e, void* tau,
                     void* ws, hipStream_t stream) {
  if (dtype == TTR_F32) return tridiag_typed<float>(n, batch, (float*)A, lda, strideA, (float*)d, (float*)e, (float*)tau, (float*)ws, stream);
  return tridiag_typed<double>(n, batch, (double*)A, lda, strideA, (double*)d, (double*)e, (double*)tau, (double*)ws, stream);
}

template <typename T>
static int eigsel_typed(int64_t n, int64_t batch, int64_t k, const T* d, const T* e, T* lam, T* Z, T* scratch, hipStream_t stream) {
  EigSelArgs<T> p;
  p.n = (int)n; p.k = (int)k; p.d = d; p.e = e; p.lam = lam; p.Z = Z; p.scratch = scratch;
  const size_t lds = (3 * (size_t)n * sizeof(T) + 15) & ~size_t(15);
  ProfScope prof(TTR_PROF_EIGH, stream);
  hipLaunchKernelGGL(tri_eigsel_kernel<T>, dim3((unsigned)batch), dim3(kWave), lds, stream, p);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}
int eigsel_dispatch(int dtype, int64_t n, int64_t batch, int64_t k, const void* d, const void* e, void* lam, void* Z, void* scratch,
                    hipStream_t stream) {
  if (dtype == TTR_F32) return eigsel_typed<float>(n, batch, k, (const float*)d, (const float*)e, (float*)lam, (float*)Z, (float*)scratch, stream);
  return eigsel_typed<double>(n, batch, k, (const double*)d, (const double*)e, (double*)lam, (double*)Z, (double*)scratch, stream);
}

template <typename T>
static int back_typed(int64_t n, int64_t batch, int64_t k, const T* A, int64_t lda, int64_t strideA, const T* tau, T* Z, hipStream_t stream) {
  BackArgs<T> p;
  p.n = (int)n; p.k = (int)k; p.A = A; p.lda = lda; p.strideA = strideA; p.tau = tau; p.Z = Z;
  ProfScope prof(TTR_PROF_EIGH, stream);
  hipLaunchKernelGGL(tridiag_back_kernel<T>, dim3((unsigned)batch), dim3(kTdThreads), 0, stream, p);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}
int tridiag_back_dispatch(int dtype, int64_t n, int64_t batch, int64_t k, const void* A, int64_t lda, int64_t strideA, const void* tau,
                          void* Z, hipStream_t stream) {
  if (dtype == TTR_F32) return back_typed<float>(n, batch, k, (const float*)A, lda, strideA, (const float*)tau, (float*)Z, stream);
  return back_typed<double>(n, batch, k, (const double*)A, lda, strideA, (const double*)tau, (double*)Z, stream);
}

}  // namespace ttr
