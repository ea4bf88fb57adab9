// Device-resident block-Jacobi driver for symmetric eigenproblems above the single-workgroup limit (gfx950).
//
// Replaces torch.linalg.eigh / svd (round.py:96, 115) for the bonds of a dense TT-SVD whose Gram matrix has n = I r > 112
// rows (BASELINE configs C1: n = 1024, C3: n = 256).  G (n x n) is cut into nbk = n / b blocks of b <= 32 columns; a
// round pairs the blocks (round-robin tournament), every pair's 2b x 2b diagonal problem is diagonalised by the LDS
// Jacobi kernel (ttr_eigh.hip, `pair_tab` mode: the four blocks are gathered where they lie) and the pair rotations are
// applied by the kernel below:
//
//     G[P, Q] <- W_p^T G[P, Q] W_q      for every pair of pairs (p, q)     (two 64 x 64 x 64 MFMA products in LDS)
//     V[:, Q] <- V[:, Q] W_q            for every chunk of 2b rows
//
// in place (each (p, q) block is owned by one workgroup and reads nothing but its own block and the two W's).  Blocks are
// never moved: the round's pairing is a small device table.  Convergence is decided ON THE DEVICE (bj_control_kernel, one
// launch per sweep): "a whole sweep rotated nothing" (relative mode) or "off-diagonal mass <= tol ||G||" / stagnation
// (absolute mode); once the control word is set every later launch of the driver returns at its first instruction, so
// the host enqueues the maximum number of sweeps without ever reading anything back.
//
// Round 2 ran this loop on the host (index_select x 3, stack, 3 GEMM launches + permute copies per round, one .item() per
// sweep: ~320 launches per n = 256 bond, 2 % of C3's roofline).
#include "ttr_common.h"

namespace ttr {

template <typename T>
struct BjApply {
  int b, npairs;               // block width, pairs per round (n = 2 b npairs)
  T* G;
  int64_t ldg, strideG;
  T* V;
  int64_t ldv, strideV;
  const int32_t* pair_tab;     // [npairs][2] block indices of this round
  const T* W;                  // [items * npairs][w][w]
  const int32_t* skip_flag;
  double* offsq;               // optional [items]: += squared off-diagonal entries of the updated G (absolute stop test)
};

constexpr int BJ_LD = 65;  // 64 x 64 operand images: odd leading dimension, row and column walks conflict-free

template <typename T>
__global__ __launch_bounds__(kThreads) void bj_apply_kernel(BjApply<T> p) {
  using M = Mfma<T>;
  using Acc = typename M::Acc;
  if (*p.skip_flag != 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char bj_smem[];  // 3 x 64 x 65 elements (fp64: 98 KB, above the static limit)
  T* const Xs = reinterpret_cast<T*>(bj_smem);
  T* const Wq = Xs + 64 * BJ_LD;
  T* const Wp = Wq + 64 * BJ_LD;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cl = lane & 15, g = lane >> 4;
  const int q = blockIdx.x, py = blockIdx.y;
  const int64_t item = blockIdx.z;
  const int b = p.b, w = 2 * b, np = p.npairs;
  const bool gblock = py < np;
  const int pr = gblock ? py : py - np;  // pair index of the block rows (G) / index of the 2b-row chunk (V)
  const int64_t qi = (int64_t)p.pair_tab[2 * q] * b, qj = (int64_t)p.pair_tab[2 * q + 1] * b;
  const int64_t pi = gblock ? (int64_t)p.pair_tab[2 * pr] * b : (int64_t)pr * w;
  const int64_t pj = gblock ? (int64_t)p.pair_tab[2 * pr + 1] * b : (int64_t)pr * w + b;
  T* __restrict__ X = gblock ? p.G + item * p.strideG : p.V + item * p.strideV;
  const int64_t ldx = gblock ? p.ldg : p.ldv;
  auto grow = [&](int i) { return i < b ? pi + i : pj + (i - b); };
  auto gcol = [&](int j) { return j < b ? qi + j : qj + (j - b); };
  const T* __restrict__ Wqg = p.W + (item * np + q) * (int64_t)w * w;
  const T* __restrict__ Wpg = p.W + (item * np + pr) * (int64_t)w * w;
  for (int idx = tid; idx < 64 * 64; idx += kThreads) {
    const int i = idx >> 6, j = idx & 63;
    const bool in = i < w && j < w;
    Xs[i * BJ_LD + j] = in ? X[grow(i) * ldx + gcol(j)] : T(0);
    Wq[i * BJ_LD + j] = in ? Wqg[i * w + j] : T(0);
    if (gblock) Wp[i * BJ_LD + j] = in ? Wpg[i * w + j] : T(0);
  }
  __syncthreads();
  // wave wv owns the 32 x 32 quadrant (wr, wc) of the 64 x 64 product: 2 x 2 tiles of 16 x 16
  const int wr = wv >> 1, wc = wv & 1;
  Acc acc[2][2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = M::zero();
  for (int ks = 0; ks < 16; ++ks) {  // T = X Wq
    const int k = 4 * ks + g;
    T a[2], bb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = Xs[(32 * wr + 16 * t + cl) * BJ_LD + k];
      bb[t] = Wq[k * BJ_LD + 32 * wc + 16 * t + cl];
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = M::mma(a[tm], bb[tn], acc[tm][tn]);
  }
  if (gblock) {
    __syncthreads();  // every wave has read its rows of X
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[(32 * wr + 16 * tm + M::row(lane, r)) * BJ_LD + 32 * wc + 16 * tn + cl] = acc[tm][tn][r];
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = M::zero();
    for (int ks = 0; ks < 16; ++ks) {  // out = Wp^T T
      const int k = 4 * ks + g;
      T a[2], bb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = Wp[k * BJ_LD + 32 * wr + 16 * t + cl];
        bb[t] = Xs[k * BJ_LD + 32 * wc + 16 * t + cl];
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = M::mma(a[tm], bb[tn], acc[tm][tn]);
    }
  }
  double off = 0.0;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 32 * wr + 16 * tm + M::row(lane, r), j = 32 * wc + 16 * tn + cl;
        if (i < w && j < w) {
          const int64_t gi = grow(i), gj = gcol(j);
          const T v = acc[tm][tn][r];
          X[gi * ldx + gj] = v;
          if (gi != gj) off += (double)v * (double)v;
        }
      }
  if (gblock && p.offsq) {
    off = wave_sum(off);
    if (lane == 0) atomicAdd(&p.offsq[item], off);
  }
}

// One launch per sweep, after its last round: sets ctrl[0] (done) from what the sweep left behind and resets the
// accumulators.  ctrl: [0] done, [1] problems that rotated in this sweep, [2] sweeps performed.  state: offsq[items],
// then one double = the previous sweep's worst off-diagonal ratio.
template <typename T>
__global__ void bj_control_kernel(int32_t* ctrl, double* state, const T* gnorm, int items, int relative, double tol) {
  if (ctrl[0] != 0) return;
  const int lane = threadIdx.x;
  int done = 0;
  if (relative) {
    done = ctrl[1] == 0;
  } else {
    double worst = 0.0;
    for (int i = lane; i < items; i += kWave) {
      const double gn = (double)gnorm[i];
      const double r = sqrt(state[i]) / (gn > 1e-300 ? gn : 1e-300);
      worst = r > worst ? r : worst;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double other = __shfl_xor(worst, o, 64);
      worst = other > worst ? other : worst;
    }
    const double prev = state[items];
    const int sweep = ctrl[2];
    // converged, or stalled AT the rounding floor of the in-place updates (a few dozen tol): the off-diagonal mass of a
    // large flat-spectrum matrix falls by less than half per sweep for the first sweeps (n = 1024: 0.08 after four) --
    // "no longer halving" alone (the round-2 rule) stopped there
    done = worst <= tol || (sweep >= 3 && prev >= 0.0 && worst > 0.7 * prev && worst <= 64.0 * tol);
    if (lane == 0) state[items] = worst;
  }
  for (int i = lane; i < items; i += kWave) state[i] = 0.0;
  if (lane == 0) {
    ctrl[1] = 0;
    ctrl[2] = ctrl[2] + 1;
    if (done) ctrl[0] = 1;
  }
}

template <typename T>
static int bj_apply_typed(int64_t b, int64_t npairs, int64_t items, T* G, int64_t ldg, int64_t strideG, T* V, int64_t ldv,
                          int64_t strideV, const int32_t* pair_tab, const T* W, const int32_t* ctrl, double* offsq,
                          hipStream_t stream) {
  BjApply<T> p;
  p.b = (int)b; p.npairs = (int)npairs;
  p.G = G; p.ldg = ldg; p.strideG = strideG;
  p.V = V; p.ldv = ldv; p.strideV = strideV;
  p.pair_tab = pair_tab; p.W = W; p.skip_flag = ctrl; p.offsq = offsq;
  const size_t lds = 3 * 64 * BJ_LD * sizeof(T);
  if (lds > 64 * 1024)
    TTR_HIP_CHECK(hipFuncSetAttribute((const void*)bj_apply_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope prof(TTR_PROF_GEMM, stream);
  for (int64_t i0 = 0; i0 < items; i0 += 65535) {  // grid.z limit
    const int64_t ni = items - i0 < 65535 ? items - i0 : 65535;
    BjApply<T> ps = p;
    ps.G = G + i0 * strideG; ps.V = V + i0 * strideV; ps.W = W + i0 * npairs * 4 * b * b;
    if (offsq) ps.offsq = offsq + i0;
    hipLaunchKernelGGL(bj_apply_kernel<T>, dim3((unsigned)npairs, (unsigned)(2 * npairs), (unsigned)ni), dim3(kThreads), lds,
                       stream, ps);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int bj_apply_dispatch(int dtype, int64_t b, int64_t npairs, int64_t items, void* G, int64_t ldg, int64_t strideG, void* V,
                      int64_t ldv, int64_t strideV, const int32_t* pair_tab, const void* W, const int32_t* ctrl, double* offsq,
                      hipStream_t stream) {
  if (dtype == TTR_F32)
    return bj_apply_typed<float>(b, npairs, items, (float*)G, ldg, strideG, (float*)V, ldv, strideV, pair_tab, (const float*)W,
                                 ctrl, offsq, stream);
  return bj_apply_typed<double>(b, npairs, items, (double*)G, ldg, strideG, (double*)V, ldv, strideV, pair_tab, (const double*)W,
                                ctrl, offsq, stream);
}

int bj_control_dispatch(int dtype, int64_t items, int32_t* ctrl, double* state, const void* gnorm, int relative, double tol,
                        hipStream_t stream) {
  if (dtype == TTR_F32)
    hipLaunchKernelGGL(bj_control_kernel<float>, dim3(1), dim3(kWave), 0, stream, ctrl, state, (const float*)gnorm, (int)items,
                       relative, tol);
  else
    hipLaunchKernelGGL(bj_control_kernel<double>, dim3(1), dim3(kWave), 0, stream, ctrl, state, (const double*)gnorm, (int)items,
                       relative, tol);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

}  // namespace ttr
