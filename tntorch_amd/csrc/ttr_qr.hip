// Batched tall-skinny Householder QR for the left-to-right TT sweep (gfx950).
//
//   A[b] (m x n) = Q[b] (m x k) R[b] (k x n),  k = min(m, n),  n <= 64
//
// The left unfolding of a TT core is (R*I) x R -- thousands of rows, a few dozen
// columns -- and, in the main use of rounding (a+b, a*b), exactly rank deficient.
// Gram-based orthogonalisation (CholeskyQR) breaks down on such inputs, so this is a
// true Householder QR (LAPACK geqrf/orgqr conventions), organised as a
// communication-avoiding TSQR:
//
//   factor:  the rows are cut into blocks of <= 256 rows; one workgroup factors one
//            block.  The block lives in REGISTERS: thread (seg, c) owns NC rows of
//            column c, so the reflector dot products v^T a_c and the rank-1 updates
//            are thread-local FMA chains, the column norm is one wave reduction, and
//            LDS only carries the current reflector (256 values) and the per-segment
//            partial dots.  The reflector is streamed to HBM transposed (Vt[j][row],
//            one coalesced 1 KiB store per step); the block's R (n x n) goes to the
//            stacked matrix of the next tree level.
//   tree:    the stacked R factors (n rows per block) are factored again by the same
//            kernel until one block is left; its R is the result.
//   apply:   Q is formed top-down: the top block applies its reflectors to [I; 0],
//            every lower block to [T_b; 0] where T_b is its n x k slice of the level
//            above.  Again thread (seg, c) owns a column segment in registers.
//
// HBM traffic per level-0 block: read 256*n, write 256*n (Vt), read Vt + write Q in
// apply: 4 * 256 * n * s bytes; the tree levels add a geometric 1/3 on top.
#include "ttr_common.h"

namespace ttr {

constexpr int BR = 256;  // rows per block (= threads: thread t also acts as "row t")

template <typename T>
struct QrLevel {
  // input matrix of this level
  const T* X;
  int64_t ldx, strideX;
  int64_t m;   // rows
  int n;       // cols
  int nb;      // row blocks (evenly split)
  T* Vt;       // [batch][nb][n][BR] reflectors, transposed, explicit (1 on the diagonal, 0 above)
  T* tau;      // [batch][nb][n]
  T* Rout;     // where the block's R goes: next level's X (row block b*n) or the user's R
  int64_t ldr, strideR;
  int top;     // 1: single block, writes k x n to the user's R
};

// 16-byte LDS reads of the (wave-uniform) reflector: 4 floats / 2 doubles per ds_read_b128.
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  typedef float type __attribute__((ext_vector_type(4)));
};
template <>
struct Vec16<double> {
  static constexpr int N = 2;
  typedef double type __attribute__((ext_vector_type(2)));
};

// w = sum_r v[r] * a[r] over a thread's NC-row column segment; v is read from LDS in 16-byte pieces,
// four independent accumulators break the FMA dependency chain.
template <typename T, int NC>
__device__ __forceinline__ T seg_dot(const T* __restrict__ v, const T (&a)[NC]) {
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  T acc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
  for (int q = 0; q < NC / VN; ++q) {
    const V x = *reinterpret_cast<const V*>(v + q * VN);
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[(q * VN + e) & 3] += x[e] * a[q * VN + e];
  }
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

template <typename T, int NC>
__device__ __forceinline__ void seg_axpy(const T* __restrict__ v, T f, T (&a)[NC]) {
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
#pragma unroll
  for (int q = 0; q < NC / VN; ++q) {
    const V x = *reinterpret_cast<const V*>(v + q * VN);
#pragma unroll
    for (int e = 0; e < VN; ++e) a[q * VN + e] -= f * x[e];
  }
}

__device__ __forceinline__ void block_rows(int64_t m, int nb, int b, int64_t& row0, int& rows) {
  const int64_t q = m / nb, rem = m % nb;
  row0 = (int64_t)b * q + (b < rem ? b : rem);
  rows = (int)(q + (b < rem ? 1 : 0));
}

// ---------------------------------------------------------------- factor
template <typename T, int NC>
__global__ __launch_bounds__(kThreads) void qr_factor_kernel(QrLevel<T> p) {
  constexpr int SEGS = kThreads / NC;  // row segments; each owns NC rows (SEGS * NC == BR)
  __shared__ __attribute__((aligned(16))) T xbuf[BR];          // current column (as stored)
  __shared__ __attribute__((aligned(16))) T vbuf[BR];          // current reflector (explicit)
  __shared__ __attribute__((aligned(16))) T wpart[SEGS][NC];   // per-segment partial dots
  __shared__ T rdiag[NC];                                      // beta_j = R[j][j]
  const int tid = threadIdx.x;
  const int c = tid % NC, seg = tid / NC;
  const int b = blockIdx.x;
  const int64_t bt = blockIdx.y;
  int64_t row0;
  int rows;
  block_rows(p.m, p.nb, b, row0, rows);
  const int n = p.n;
  const int kb = rows < n ? rows : n;

  const T* __restrict__ X = p.X + bt * p.strideX + row0 * p.ldx;
  T a[NC];
#pragma unroll
  for (int r = 0; r < NC; ++r) {
    const int rg = seg * NC + r;
    a[r] = (rg < rows && c < n) ? X[(int64_t)rg * p.ldx + c] : T(0);
  }
  T* __restrict__ Vt = p.Vt + ((bt * p.nb + b) * (int64_t)n) * BR;
  T* __restrict__ tau = p.tau + (bt * p.nb + b) * (int64_t)n;
  if (tid < NC) rdiag[tid] = T(0);

  for (int j = 0; j < kb; ++j) {
    // (1) the owner of column j publishes it (16-byte LDS stores, one lane per segment)
    if (c == j) {
      using V = typename Vec16<T>::type;
      constexpr int VN = Vec16<T>::N;
#pragma unroll
      for (int q = 0; q < NC / VN; ++q) {
        V x;
#pragma unroll
        for (int e = 0; e < VN; ++e) x[e] = a[q * VN + e];
        *reinterpret_cast<V*>(&xbuf[seg * NC + q * VN]) = x;
      }
    }
    __syncthreads();
    // (2) every wave reduces the whole column redundantly: no cross-wave step
    T ss = 0;
#pragma unroll
    for (int q = 0; q < BR / kWave; ++q) {
      const int rg = (tid & 63) + q * kWave;
      const T x = xbuf[rg];
      ss += (rg > j) ? x * x : T(0);
    }
    ss = wave_sum(ss);
    const T alpha = xbuf[j];
    const T xt = xbuf[tid];
    T beta, tj, scale;
    if (ss == T(0)) {  // LAPACK larfg: H = I
      beta = alpha; tj = T(0); scale = T(0);
    } else {
      beta = -copysign(sqrt(alpha * alpha + ss), alpha);
      tj = (beta - alpha) / beta;
      scale = T(1) / (alpha - beta);
    }
    const T vt = (tid > j) ? xt * scale : (tid == j ? T(1) : T(0));
    vbuf[tid] = vt;
    Vt[(int64_t)j * BR + tid] = vt;  // coalesced, fire and forget
    if (tid == 0) { tau[j] = tj; rdiag[j] = beta; }
    __syncthreads();
    // (3) the columns right of j take the dot product with the reflector.  The owner's registers are
    //     left alone: rows < j of column j are final R entries, the diagonal lives in rdiag, and the
    //     reflector itself already went to Vt.
    const bool active = (c > j && c < n);
    if (active) wpart[seg][c] = seg_dot<T, NC>(&vbuf[seg * NC], a);
    __syncthreads();
    // (4) rank-1 update of the trailing columns
    if (active) {
      T w = 0;
#pragma unroll
      for (int s2 = 0; s2 < SEGS; ++s2) w += wpart[s2][c];
      seg_axpy<T, NC>(&vbuf[seg * NC], tj * w, a);
    }
    // next (1) writes xbuf (last read before the 2nd barrier) -> no barrier needed here;
    // vbuf/wpart are rewritten only after the next iteration's first barrier.
  }
  __syncthreads();

  // R: rows 0..kb-1 live in segment 0 (kb <= n <= NC)
  T* __restrict__ Rout = p.Rout + bt * p.strideR + (p.top ? 0 : (int64_t)b * n * p.ldr);
  const int rrows = p.top ? kb : n;
  if (seg == 0 && c < n) {
    const T dg = rdiag[c];
#pragma unroll
    for (int r = 0; r < NC; ++r) {
      if (r < rrows) {
        T v = T(0);
        if (r < kb) v = (r < c) ? a[r] : (r == c ? dg : T(0));
        Rout[(int64_t)r * p.ldr + c] = v;
      }
    }
  }
}

// ---------------------------------------------------------------- apply (form Q top-down)
template <typename T>
struct QrApply {
  const T* Vt;
  const T* tau;
  int64_t m;   // rows of this level's matrix
  int n;       // reflector count upper bound / cols of the factored matrix
  int nb;
  int kcols;   // columns of Q being formed (k)
  const T* Top;  // level above: (nb * n) x kcols, row block b*n ; nullptr => identity
  int64_t ldtop, strideTop;
  T* Out;
  int64_t ldout, strideOut;
};

template <typename T, int NC>
__global__ __launch_bounds__(kThreads) void qr_apply_kernel(QrApply<T> p) {
  constexpr int SEGS = kThreads / NC;
  __shared__ __attribute__((aligned(16))) T vbuf[3][BR];
  __shared__ __attribute__((aligned(16))) T wpart[2][SEGS][NC];
  __shared__ T taus[NC];
  const int tid = threadIdx.x;
  const int c = tid % NC, seg = tid / NC;
  const int b = blockIdx.x;
  const int64_t bt = blockIdx.y;
  int64_t row0;
  int rows;
  block_rows(p.m, p.nb, b, row0, rows);
  const int n = p.n;
  const int kb = rows < n ? rows : n;
  const int kc = p.kcols;

  const T* __restrict__ Vt = p.Vt + ((bt * p.nb + b) * (int64_t)n) * BR;
  const T* __restrict__ tau = p.tau + (bt * p.nb + b) * (int64_t)n;
  if (tid < n) taus[tid] = tid < kb ? tau[tid] : T(0);

  T a[NC];
  if (p.Top) {
    const T* __restrict__ Top = p.Top + bt * p.strideTop + (int64_t)b * n * p.ldtop;
#pragma unroll
    for (int r = 0; r < NC; ++r) {
      const int rg = seg * NC + r;
      a[r] = (rg < n && rg < rows && c < kc) ? Top[(int64_t)rg * p.ldtop + c] : T(0);
    }
  } else {
#pragma unroll
    for (int r = 0; r < NC; ++r) a[r] = (seg * NC + r == c && c < kc) ? T(1) : T(0);
  }

  // reflectors are applied last-to-first: Q [T;0] = H_0 H_1 ... H_{kb-1} [T;0]
  T vnext = kb > 0 ? Vt[(int64_t)(kb - 1) * BR + tid] : T(0);
  if (kb > 0) vbuf[(kb - 1) % 3][tid] = vnext;
  __syncthreads();
  for (int j = kb - 1; j >= 0; --j) {
    const int cur = j % 3;
    const int wb = j & 1;
    if (j > 0) vnext = Vt[(int64_t)(j - 1) * BR + tid];  // prefetch, lands under the FMAs
    wpart[wb][seg][c] = seg_dot<T, NC>(&vbuf[cur][seg * NC], a);
    if (j > 0) vbuf[(j - 1) % 3][tid] = vnext;
    __syncthreads();
    T ws = 0;
#pragma unroll
    for (int s = 0; s < SEGS; ++s) ws += wpart[wb][s][c];
    seg_axpy<T, NC>(&vbuf[cur][seg * NC], taus[j] * ws, a);
  }

  T* __restrict__ Out = p.Out + bt * p.strideOut + row0 * p.ldout;
  if (c < kc) {
#pragma unroll
    for (int r = 0; r < NC; ++r) {
      const int rg = seg * NC + r;
      if (rg < rows) Out[(int64_t)rg * p.ldout + c] = a[r];
    }
  }
}

// ---------------------------------------------------------------- host-side tree
struct QrPlan {
  int levels;
  int64_t m[16];
  int nb[16];
  // workspace offsets in elements
  int64_t off_vt[16], off_tau[16], off_x[16], off_out[16];
  int64_t total;  // elements
};

static QrPlan make_plan(int64_t m, int64_t n, int64_t batch) {
  QrPlan pl{};
  int64_t cur = m;
  int L = 0;
  for (;;) {
    pl.m[L] = cur;
    pl.nb[L] = (int)ceil_div(cur, BR);
    ++L;
    if (pl.nb[L - 1] <= 1) break;
    cur = (int64_t)pl.nb[L - 1] * n;
  }
  pl.levels = L;
  int64_t off = 0;
  for (int l = 0; l < L; ++l) {
    pl.off_vt[l] = off; off += batch * pl.nb[l] * n * BR;
    pl.off_tau[l] = off; off += align_up(batch * pl.nb[l] * n, 64);
    if (l > 0) {
      pl.off_x[l] = off; off += batch * pl.m[l] * n;    // stacked R factors (input of level l)
      pl.off_out[l] = off; off += batch * pl.m[l] * n;  // Q of level l (m_l x k)
    }
  }
  pl.total = off;
  return pl;
}

int64_t qr_workspace_bytes(int dtype, int64_t m, int64_t n, int64_t batch) {
  if (m <= 0 || n <= 0 || batch <= 0) return 0;
  return make_plan(m, n, batch).total * (dtype == TTR_F64 ? 8 : 4);
}

template <typename T, int NC>
static int qr_run(int64_t m, int n, int64_t batch, const T* A, int64_t lda, int64_t strideA, T* Q, int64_t ldq,
                  int64_t strideQ, T* R, int64_t ldr, int64_t strideR, T* ws, const QrPlan& pl, hipStream_t stream) {
  const int L = pl.levels;
  const int k = (int)(m < n ? m : n);
  // ---- factor, bottom-up
  for (int l = 0; l < L; ++l) {
    QrLevel<T> p;
    p.X = l == 0 ? A : ws + pl.off_x[l];
    p.ldx = l == 0 ? lda : n;
    p.strideX = l == 0 ? strideA : pl.m[l] * n;
    p.m = pl.m[l]; p.n = n; p.nb = pl.nb[l];
    p.Vt = ws + pl.off_vt[l];
    p.tau = ws + pl.off_tau[l];
    p.top = (l == L - 1);
    if (p.top) { p.Rout = R; p.ldr = ldr; p.strideR = strideR; }
    else { p.Rout = ws + pl.off_x[l + 1]; p.ldr = n; p.strideR = pl.m[l + 1] * n; }
    ProfScope prof(TTR_PROF_QR_FACTOR, stream);
    hipLaunchKernelGGL((qr_factor_kernel<T, NC>), dim3((unsigned)pl.nb[l], (unsigned)batch), dim3(kThreads), 0, stream, p);
  }
  // ---- form Q, top-down
  for (int l = L - 1; l >= 0; --l) {
    QrApply<T> p;
    p.Vt = ws + pl.off_vt[l];
    p.tau = ws + pl.off_tau[l];
    p.m = pl.m[l]; p.n = n; p.nb = pl.nb[l]; p.kcols = k;
    if (l == L - 1) { p.Top = nullptr; p.ldtop = 0; p.strideTop = 0; }
    else { p.Top = ws + pl.off_out[l + 1]; p.ldtop = k; p.strideTop = pl.m[l + 1] * n; }
    if (l == 0) { p.Out = Q; p.ldout = ldq; p.strideOut = strideQ; }
    else { p.Out = ws + pl.off_out[l]; p.ldout = k; p.strideOut = pl.m[l] * n; }
    ProfScope prof(TTR_PROF_QR_APPLY, stream);
    hipLaunchKernelGGL((qr_apply_kernel<T, NC>), dim3((unsigned)pl.nb[l], (unsigned)batch), dim3(kThreads), 0, stream, p);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

template <typename T>
static int qr_typed(int64_t m, int64_t n, int64_t batch, const void* A, int64_t lda, int64_t strideA, void* Q,
                    int64_t ldq, int64_t strideQ, void* R, int64_t ldr, int64_t strideR, void* ws, int64_t ws_bytes,
                    hipStream_t stream) {
  const QrPlan pl = make_plan(m, n, batch);
  TTR_REQUIRE(ws_bytes >= pl.total * (int64_t)sizeof(T), TTR_E_WORKSPACE, "ttr_qr: workspace %lld < %lld bytes",
              (long long)ws_bytes, (long long)(pl.total * (int64_t)sizeof(T)));
  TTR_REQUIRE(batch <= 65535, TTR_E_UNSUPPORTED, "ttr_qr: batch %lld > 65535", (long long)batch);
#define TTR_QR_CASE(NCV)                                                                                          \
  return qr_run<T, NCV>(m, (int)n, batch, (const T*)A, lda, strideA, (T*)Q, ldq, strideQ, (T*)R, ldr, strideR, \
                        (T*)ws, pl, stream)
  if (n <= 16) TTR_QR_CASE(16);
  if (n <= 32) TTR_QR_CASE(32);
  TTR_QR_CASE(64);
#undef TTR_QR_CASE
}

int qr_max_cols(int) { return 64; }

int qr_dispatch(int dtype, int64_t m, int64_t n, int64_t batch, const void* A, int64_t lda, int64_t strideA, void* Q,
                int64_t ldq, int64_t strideQ, void* R, int64_t ldr, int64_t strideR, void* ws, int64_t ws_bytes,
                hipStream_t stream) {
  TTR_REQUIRE(n <= qr_max_cols(dtype), TTR_E_UNSUPPORTED, "ttr_qr: n = %lld exceeds the %d-column panel kernel",
              (long long)n, qr_max_cols(dtype));
  if (dtype == TTR_F32)
    return qr_typed<float>(m, n, batch, A, lda, strideA, Q, ldq, strideQ, R, ldr, strideR, ws, ws_bytes, stream);
  return qr_typed<double>(m, n, batch, A, lda, strideA, Q, ldq, strideQ, R, ldr, strideR, ws, ws_bytes, stream);
}

}  // namespace ttr
