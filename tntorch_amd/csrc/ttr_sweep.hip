// Fused kernels of the right-to-left truncation sweep (gfx950): everything that touches the 64 x (I r) right
// unfolding M of a core between the push-left and the new core.
//
//   ttr_rowgram   G  = M M^T                        (pass 1 of the two-pass 'svd' truncation, and 'eig')   round.py:104-109
//   ttr_rotgram   G2 = (V1^T M)(V1^T M)^T           (pass 2: Gram matrix of the ROTATED rows; the rotated matrix itself is
//                                                    never written -- it only exists 16 columns at a time in registers)
//   ttr_project   right = diag(1/sigma) (V1 V2r)^T M,  left = (V1 V2r) diag(sigma)                          round.py:163-172
//
// All three stream M once, 16 (projection: 32) columns per wave and step, and keep every intermediate in MFMA accumulator registers:
//   * the rotated 16-column slab is computed TRANSPOSED (slab^T = M_c^T V1: A operand = 16 x 4 pieces of M read straight
//     from global memory, B operand = V1 from LDS), which leaves it in the accumulator layout
//     lane (g, cl), register r  <->  slab[row 16 t + cl][column 4 g + r];
//   * in that layout the accumulator registers ARE both MFMA operands of the Gram update
//     G[16 ti + i][16 tj + j] += sum_c slab[16 ti + i][c] slab[16 tj + j][c]  (A[i][k]: lane i + 16 k, B[k][j]: lane j + 16 k,
//     k <-> column 4 k + r): no LDS round trip, no transposition, 10 of the 16 tiles (symmetry).
// Arithmetic per 16 columns and wave: 64 + 40 MFMA 16x16x4 for the rotation + Gram (AI = 64 flop/B: bound by the fp32
// MFMA rate, which equals the vector rate on gfx950), 32 for the projection to rank 32 (AI = 11 flop/B: HBM-bound).
#include <type_traits>

#include "ttr_common.h"

namespace ttr {

constexpr int KLD = 80;  // leading dimension of the LDS images [k][i]: fragment reads (16 i x 2 k per 32 lanes) hit 32 banks

template <typename T>
struct SweepArgs {
  int R;            // rows of M (<= 64)
  int64_t n;        // columns of M
  const T* M;
  int64_t ldm, strideM;
  const T* V1;      // R x R or nullptr
  int64_t ldv1, strideV1;
  // rowgram / rotgram
  T* G;             // [batch][nsplit][R][R]
  int nsplit;
  const int32_t* skip;  // optional [batch]: != 0 -> this item's workgroups return at once (its G is not written)
  const int32_t* rows32;  // optional [batch] (rowgram, project): != 0 -> rows 32.. of this item's M are exactly zero and are not loaded
  // project
  const T* V2;      // R x R (columns = directions), first ro used
  int64_t ldv2, strideV2;
  const T* sigma;   // [ro] per item (nullable: no scaling)
  int64_t stride_sigma;
  int ro;           // output rank (<= 64)
  int scale_right;  // 1: right rows divided by sigma (round.py:168), left columns multiplied by sigma
  T* right;         // ro x n
  int64_t ldr, strideR;
  T* left;          // R x ro (nullable)
  int64_t ldl, strideL;
  int stagger;      // TTR_KNOB_SWEEP_STAGGER: 0 = columns in order, 1 = project staggered per item, 2 = the Gram kernels as well
};

__host__ __device__ inline void split_range(int64_t chunks, int nsplit, int split, int64_t& c0, int64_t& c1) {
  const int64_t per = (chunks + nsplit - 1) / nsplit;
  c0 = (int64_t)split * per;
  c1 = c0 + per < chunks ? c0 + per : chunks;
}

// N consecutive elements with one (or two) wide loads when the caller established the alignment, element-wise otherwise
template <typename T, int N>
struct Pack {
  T v[N];
};
template <typename T, int N>
__device__ __forceinline__ Pack<T, N> load_pack(const T* __restrict__ p, bool aligned, int64_t valid) {
  Pack<T, N> out;
  if (aligned && valid >= N) {
    typedef T VT __attribute__((ext_vector_type(N)));
    const VT x = *reinterpret_cast<const VT*>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) out.v[i] = x[i];
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) out.v[i] = (i < valid) ? p[i] : T(0);
  }
  return out;
}

// ---------------------------------------------------------------- Gram matrix of the (rotated) rows
template <typename T, bool IDENT>
__global__ __launch_bounds__(kThreads) void rotgram_kernel(SweepArgs<T> p) {
  using M = Mfma<T>;
  using Acc = typename M::Acc;
  constexpr int RED = 4 * 10 * 256;  // per-wave partial tiles for the final reduction
  constexpr int VSZ = 64 * KLD;
  __shared__ __attribute__((aligned(16))) T smem[RED > VSZ ? RED : VSZ];
  T* Vl = smem;  // V1 as [k][row], zero padded to 64 x 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 15, g = lane >> 4;
  const int64_t b = blockIdx.y;
  if (p.skip && p.skip[b] != 0) return;  // (block-uniform) pass 2 is not needed for this item: see ttr_spectrum_flat
  const int split = blockIdx.x;
  const T* __restrict__ Mp = p.M + b * p.strideM;
  const int R = p.R;
  if constexpr (!IDENT) {
    const T* __restrict__ V1 = p.V1 + b * p.strideV1;
    for (int idx = tid; idx < 64 * 64; idx += kThreads) {
      const int k = idx >> 6, j = idx & 63;
      Vl[k * KLD + j] = (k < R && j < R) ? V1[(int64_t)k * p.ldv1 + j] : T(0);
    }
    __syncthreads();
  }
  Acc G[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) G[i] = M::zero();
  // (row tiles of an item whose rows 32.. are exactly zero -- rows32 -- are skipped: 3 of the 10 tile products remain)
  const int ntl = (IDENT && p.rows32 && p.rows32[b] != 0 && R > 32) ? 2 : 4;
  auto gram_update = [&](const Acc (&mw)[4]) {
    int idx = 0;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = ti; tj < 4; ++tj) {
        if (tj < ntl) {   // block-uniform
#pragma unroll
          for (int r = 0; r < 4; ++r) G[idx] = M::mma(mw[ti][r], mw[tj][r], G[idx]);
        }
        ++idx;
      }
  };
  if constexpr (IDENT) {
    // 16 columns per step: lane (g, cl) register r <-> M[16 t + cl][c0 + 4 g + r] (any assignment of the 16 columns to
    // (g, r) serves the Gram sum): four consecutive elements, one 16-byte load per row tile
    const bool al = ((p.ldm & 3) == 0) && ((reinterpret_cast<uintptr_t>(Mp) & (4 * sizeof(T) - 1)) == 0);
    int64_t cb, ce;
    split_range((p.n + 15) / 16, p.nsplit, split, cb, ce);
    // (the next step's loads are issued before the current step's MFMAs: with 4 waves per SIMD the 40 MFMAs of a step do not
    // cover an HBM round trip)
    // (rows32: the carry of a bond whose QR packed its rows -- ttr_qr_factor_pushed on an R factor of numerical rank <= 32 -- has
    // exactly zero rows 32..: they are not loaded, their Gram tiles come out as the zeros they are)
    const int Rl = (p.rows32 && p.rows32[b] != 0 && R > 32) ? 32 : R;
    auto load_slab = [&](int64_t c, Acc (&mw)[4]) {
      const int64_t col = c * 16 + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = 16 * t + cl;
        const Pack<T, 4> x = load_pack<T, 4>(Mp + (int64_t)row * p.ldm + col, al, row < Rl ? p.n - col : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) mw[t][r] = x.v[r];
      }
    };
    // (step order: see project_kernel; the Gram sums are accumulated in an item-dependent order when staggered, knob value 2)
    const int64_t nsteps = ce > cb + wave ? (ce - cb - wave + 3) / 4 : 0;
    const int64_t rot = (p.stagger >= 2 && nsteps > 1) ? (int64_t)(b % nsteps) : 0;
    auto step_c = [&](int64_t sidx) { int64_t sq = sidx + rot; if (sq >= nsteps) sq -= nsteps; return cb + wave + 4 * sq; };
    Acc cur[4], nxt[4];
    if (nsteps > 0) load_slab(step_c(0), cur);
    for (int64_t sidx = 0; sidx < nsteps; ++sidx) {
      const bool more = sidx + 1 < nsteps;
      if (more) load_slab(step_c(sidx + 1), nxt);
      gram_update(cur);
      if (more) {
#pragma unroll
        for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
      }
    }
  } else {
    // 16 columns per step; A operand = 16 x 4 pieces of M straight from global memory (four 64-byte row segments per
    // load).  The kernel runs at ~80 % of the fp32 MFMA rate like this (measured): wider loads only cost registers.
    int64_t cb, ce;
    split_range((p.n + 15) / 16, p.nsplit, split, cb, ce);
    const int Rl = (p.rows32 && p.rows32[b] != 0 && R > 32) ? 32 : R;   // (zero rows 32..: not loaded -- they may not even be written)
    auto load_cols = [&](int64_t c, T (&a)[16]) {
      const int64_t c0 = c * 16;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int k = 4 * ks + g;
        a[ks] = (k < Rl && c0 + cl < p.n) ? Mp[(int64_t)k * p.ldm + c0 + cl] : T(0);
      }
    };
    const int64_t nsteps = ce > cb + wave ? (ce - cb - wave + 3) / 4 : 0;
    const int64_t rot = (p.stagger >= 2 && nsteps > 1) ? (int64_t)(b % nsteps) : 0;
    auto step_c = [&](int64_t sidx) { int64_t sq = sidx + rot; if (sq >= nsteps) sq -= nsteps; return cb + wave + 4 * sq; };
    T a[16], an[16];
    if (nsteps > 0) load_cols(step_c(0), a);
    for (int64_t sidx = 0; sidx < nsteps; ++sidx) {
      const bool more = sidx + 1 < nsteps;
      if (more) load_cols(step_c(sidx + 1), an);  // in flight under this step's 104 MFMAs (two waves per SIMD do not cover an HBM round trip)
      Acc mw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        mw[t] = M::zero();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) mw[t] = M::mma(a[ks], Vl[(4 * ks + g) * KLD + 16 * t + cl], mw[t]);
      }
      gram_update(mw);
      if (more) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) a[ks] = an[ks];
      }
    }
  }
  // reduce the four waves' partial tiles and write the block's partial Gram matrix (both triangles)
  __syncthreads();  // Vl is dead: smem becomes the reduction buffer
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) smem[(wave * 10 + i) * 256 + lane * 4 + r] = G[i][r];
  __syncthreads();
  T* __restrict__ Gout = p.G + ((int64_t)b * p.nsplit + split) * R * R;
  for (int idx = tid; idx < 10 * 256; idx += kThreads) {
    const int tile = idx >> 8, e = idx & 255, ln = e >> 2, r = e & 3;
    const T v = (smem[idx] + smem[2560 + idx]) + (smem[5120 + idx] + smem[7680 + idx]);
    int ti = 0, rem = tile;  // tile -> (ti, tj), ti <= tj, row-major over the upper triangle of a 4 x 4 tile grid
    while (rem >= 4 - ti) { rem -= 4 - ti; ++ti; }
    const int tj = ti + rem;
    const int row = 16 * ti + M::row(ln, r), col = 16 * tj + (ln & 15);
    if (row < R && col < R) {
      Gout[row * R + col] = v;
      if (ti != tj) Gout[col * R + row] = v;
    }
  }
}

// ---------------------------------------------------------------- projection onto the kept directions
// LDS image of U = V1 V2[:, :ro] shared by the projection kernels: rows k in blocks of 16 (one block per wave of the
// prologue), block stride UBLK, row stride KLD -- so that a wave can write ITS 16 rows over the V1 rows only it reads.
constexpr int V1LD = 82;            // V1 as [k][m]: A-operand reads (16 k x 2 m per 32 lanes: words 82 cl + g) hit 32 banks
constexpr int UBLK = 16 * V1LD;     // 1312 >= 16 * KLD
__device__ __forceinline__ int urow(int k) { return (k >> 4) * UBLK + (k & 15) * KLD; }

template <typename T>
__global__ __launch_bounds__(kThreads, (sizeof(T) == 4 ? 4 : 1)) void project_kernel(SweepArgs<T> p) {
  using M = Mfma<T>;
  using Acc = typename M::Acc;
  // V2[:, :ro] as [m][i], zero padded (prologue only: leading dimension 72 instead of the conflict-free 80 -- with it the three
  // arrays take 39.7 KB, FOUR workgroups per CU instead of three; the main loop has no software prefetch (stores sit between
  // its loads, and a prefetched step's loads then wait in the same in-order counter: measured 416 -> 700 us), so the HBM
  // round trip is covered by the number of resident waves alone)
  constexpr int V2LD = 72;
  __shared__ __attribute__((aligned(16))) T V2l[64 * V2LD];
  __shared__ __attribute__((aligned(16))) T V1l[64 * V1LD];  // V1 as [k][m]; wave w then overwrites ITS rows 16 w .. 16 w + 15
  T* Ul = V1l;                                               // with U rows (urow()): no other wave reads them before
  __shared__ T isg[64];  // 1 / sigma of the output rows (TTR_SCALE_DIV semantics: 0 below the smallest normal), or 1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 15, g = lane >> 4;
  const int64_t b = blockIdx.y;
  const int split = blockIdx.x;
  const T* __restrict__ Mp = p.M + b * p.strideM;
  const int R = p.R, ro = p.ro;
  const int nt = (ro + 15) / 16;
  const T* __restrict__ V2 = p.V2 + b * p.strideV2;
  const T* __restrict__ sg = p.sigma ? p.sigma + b * p.stride_sigma : nullptr;
  T* __restrict__ Lo = (p.left && split == 0) ? p.left + b * p.strideL : nullptr;
  // ---- prologue: U = V1 V2[:, :ro] on the matrix cores (or U = V2[:, :ro]); split 0 also emits left = U diag(sigma)
  for (int idx = tid; idx < 64 * 64; idx += kThreads) {
    const int m = idx >> 6, i = idx & 63;
    V2l[m * V2LD + i] = (m < R && i < ro) ? V2[(int64_t)m * p.ldv2 + i] : T(0);
  }
  if (tid < 64) {
    T sc = T(1);
    if (p.scale_right && sg && tid < ro) {
      const T x = sg[tid];
      sc = (fabs((double)x) < (double)Num<T>::tiny()) ? T(0) : T(1) / x;
    }
    isg[tid] = sc;
  }
  if (p.V1) {
    const T* __restrict__ V1 = p.V1 + b * p.strideV1;
    for (int idx = tid; idx < 64 * 64; idx += kThreads) {
      const int k = idx >> 6, m = idx & 63;
      V1l[k * V1LD + m] = (k < R && m < R) ? V1[(int64_t)k * p.ldv1 + m] : T(0);
    }
    __syncthreads();
    // wave w: rows 16 w .. 16 w + 15 of U.  All four tiles are computed before the first one is stored (the stores
    // overwrite the wave's own V1 rows), and go straight from the accumulators to LDS.
    Acc u0 = M::zero(), u1 = M::zero(), u2 = M::zero(), u3 = M::zero();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const T a = V1l[(16 * wave + cl) * V1LD + 4 * ks + g];
      u0 = M::mma(a, V2l[(4 * ks + g) * V2LD + cl], u0);
      u1 = M::mma(a, V2l[(4 * ks + g) * V2LD + 16 + cl], u1);
      u2 = M::mma(a, V2l[(4 * ks + g) * V2LD + 32 + cl], u2);
      u3 = M::mma(a, V2l[(4 * ks + g) * V2LD + 48 + cl], u3);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T* dst = Ul + urow(16 * wave + M::row(lane, r)) + cl;
      dst[0] = u0[r];
      dst[16] = u1[r];
      dst[32] = u2[r];
      dst[48] = u3[r];
    }
  } else {
    __syncthreads();
    for (int idx = tid; idx < 64 * 64; idx += kThreads) {
      const int k = idx >> 6, i = idx & 63;
      Ul[urow(k) + i] = V2l[k * V2LD + i];
    }
  }
  __syncthreads();
  if (Lo) {  // left = U diag(sigma), coalesced, from the LDS image
    for (int idx = tid; idx < R * ro; idx += kThreads) {
      const int k = idx / ro, i = idx - k * ro;
      const T u = Ul[urow(k) + i];
      Lo[(int64_t)k * p.ldl + i] = (p.scale_right && sg) ? u * sg[i] : u;
    }
  }
  // ---- main loop: 16 SL columns per step as SL interleaved 16-column slabs (columns c0 + SL i + u): one 4 SL-byte load of M per
  // row and lane (whole 128-byte lines per 16 lanes) and 4 SL-byte stores of the result; SL = 2 (see the end of the kernel).
  T* __restrict__ Ro = p.right + b * p.strideR;
  const bool al2 = ((p.ldm & 1) == 0) && ((p.ldr & 1) == 0) && ((reinterpret_cast<uintptr_t>(Mp) & (2 * sizeof(T) - 1)) == 0) &&
                   ((reinterpret_cast<uintptr_t>(Ro) & (2 * sizeof(T) - 1)) == 0);
  const bool al4 = ((p.ldm & 3) == 0) && ((p.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(Mp) & (4 * sizeof(T) - 1)) == 0) &&
                   ((reinterpret_cast<uintptr_t>(Ro) & (4 * sizeof(T) - 1)) == 0);
  const int Rl = (p.rows32 && p.rows32[b] != 0 && R > 32) ? 32 : R;   // (see rotgram_kernel: zero rows are not loaded)
  // Step order: every item's rows lie ldm elements apart (8 KB at the metric's bonds) and every item is aligned alike, so
  // workgroups that walk their columns in step would keep hitting the same HBM channels (address bits 8 .. 12 are the column's;
  // measured on ttr_orth_fixup, profiles/r05_orth_stamps.txt).  Item b starts at step b mod nsteps and wraps around; the columns
  // are independent of each other, so the result is bit-identical in any order.  (TTR_KNOB_SWEEP_STAGGER = 0: in order.)
  // NK = K steps that carry data (round 6): a `rows32` item (rows 32.. exactly zero: not loaded) used to multiply its 8 zero K steps
  // all the same -- half of the kernel's MFMAs (counters: the matrix pipe 43 % busy on the metric's input); skipped products add
  // a * 0 to their accumulators: bit-identical
  auto main_loop = [&](auto NKC, auto SLC) {
    constexpr int NK = decltype(NKC)::value, SL = decltype(SLC)::value;
    const bool al = SL == 4 ? al4 : al2;
    // the split (nsplit: pick_split) is defined on 32-column chunks; a wave's steps walk its share SL * 16 columns at a time
    int64_t cb, ce;
    split_range((p.n + 31) / 32, p.nsplit, split, cb, ce);
    const int64_t col_b = cb * 32, col_e = ce * 32 < p.n ? ce * 32 : p.n;
    const int64_t nch = (col_e - col_b + 16 * SL - 1) / (16 * SL);
    const int64_t nsteps = nch > wave ? (nch - wave + 3) / 4 : 0;
    const int64_t rot = (p.stagger && nsteps > 1) ? (int64_t)(b % nsteps) : 0;
    for (int64_t sidx = 0; sidx < nsteps; ++sidx) {
      int64_t sq = sidx + rot;
      if (sq >= nsteps) sq -= nsteps;
      const int64_t col = col_b + (wave + 4 * sq) * (16 * SL) + SL * cl;
      const int64_t lim = col_e - col;     // columns of this lane's group inside the split's share
      T bm[SL][NK];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const int k = 4 * ks + g;
        const Pack<T, SL> x = load_pack<T, SL>(Mp + (int64_t)k * p.ldm + col, al, (k < Rl && lim > 0) ? lim : 0);
#pragma unroll
        for (int u = 0; u < SL; ++u) bm[u][ks] = x.v[u];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < nt) {
          Acc acc[SL];
#pragma unroll
          for (int u = 0; u < SL; ++u) acc[u] = M::zero();
#pragma unroll
          for (int ks = 0; ks < NK; ++ks) {
            const T af = Ul[urow(4 * ks + g) + 16 * t + cl];
#pragma unroll
            for (int u = 0; u < SL; ++u) acc[u] = M::mma(af, bm[u][ks], acc[u]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + M::row(lane, r);
            if (row < ro && lim > 0) {
              const T sc = isg[row];
              T* dst = Ro + (int64_t)row * p.ldr + col;
              if (al && lim >= SL) {
                typedef T VT __attribute__((ext_vector_type(SL)));
                VT o;
#pragma unroll
                for (int u = 0; u < SL; ++u) o[u] = acc[u][r] * sc;
                *reinterpret_cast<VT*>(dst) = o;
              } else {
#pragma unroll
                for (int u = 0; u < SL; ++u)
                  if (u < lim) dst[u] = acc[u][r] * sc;
              }
            }
          }
        }
      }
    }
  };
  using I2 = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>;
  using I8 = std::integral_constant<int, 8>;
  using I16 = std::integral_constant<int, 16>;
  // SL = 4 (16-byte accesses) is compiled only with -DTTR_PROJECT_WIDE: measured SLOWER on the kind (3.14 against 2.96 - 3.01 ms per
  // 4096-train step, profiles/r06_project_ab.txt; review item 4's wide accesses), and 16 K steps x 4 slabs spill under the
  // 128-register cap of four workgroups per CU
  if (Rl <= 32) {
#ifdef TTR_PROJECT_WIDE
    if (al4) main_loop(I8{}, I4{});
    else
#endif
      main_loop(I8{}, I2{});
  } else {
    main_loop(I16{}, I2{});
  }
  (void)al4;
}

// ================================================================ tall matrices (dense TT-SVD steps): M is rows x n, n <= 64
// The first (largest) step of a dense right-to-left TT-SVD truncates a (prod I_1..I_{N-1}) x I_N unfolding: millions of
// rows, at most 64 columns, and the whole input tensor in bytes.  Same register technique with the roles of rows and
// columns exchanged -- the contraction runs over the ROWS, 16 per wave and step:
//   ttr_colgram    G = M^T M                (ROT = false: the 16 x 16 pieces of M are loaded straight into the accumulator
//                                            layout, 64-byte row segments) or G = (M V1)^T (M V1) (ROT = true: the rotated
//                                            16-row slab is an MFMA product whose A operand is read with 16-byte loads)
//   ttr_colproject left = M U [diag(1/sigma)]  (rows x ro),  right = [diag(sigma)] U^T (ro x n),  U = V1 V2[:, :ro]
// 'svd' therefore reads the tensor three times and writes only the carry (ro / n of its size) -- no rotated copy.
constexpr int CLD = 68;  // LDS images [k][j] read with k = 16 kk + 4 g + u: 4 * 68 = 16 mod 32 -> the two lane rows of a
                         // 32-lane group hit disjoint banks

template <typename T>
struct ColArgs {
  int64_t rows;
  int n;
  const T* M;
  int64_t ldm, strideM;
  const T* V1;
  int64_t ldv1, strideV1;
  T* G;        // [batch][nsplit][n][n]
  int nsplit;
  const int32_t* skip;  // optional [batch]: != 0 -> the item's workgroups return at once (colgram; see ttr_spectrum_flat)
  const T* V2;
  int64_t ldv2, strideV2;
  const T* sigma;
  int64_t stride_sigma;
  int ro;
  int left_ortho;  // 1: left columns divided by sigma, right rows multiplied (round.py:173-178); 0: plain (round.py:179-182)
  T* left;
  int64_t ldl, strideL;
  T* right;
  int64_t ldr, strideR;
};

template <typename T, bool ROT>
__global__ __launch_bounds__(kThreads, (sizeof(T) == 4 ? (ROT ? 3 : 4) : 1)) void colgram_kernel(ColArgs<T> p) {
  using M = Mfma<T>;
  using Acc = typename M::Acc;
  constexpr int RED = 4 * 10 * 256;
  constexpr int VSZ = 64 * CLD;
  __shared__ __attribute__((aligned(16))) T smem[RED > VSZ ? RED : VSZ];
  T* Vl = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 15, g = lane >> 4;
  const int64_t b = blockIdx.y;
  if (p.skip && p.skip[b] != 0) return;
  const int split = blockIdx.x;
  const T* __restrict__ Mp = p.M + b * p.strideM;
  const int n = p.n;
  if constexpr (ROT) {
    const T* __restrict__ V1 = p.V1 + b * p.strideV1;
    for (int idx = tid; idx < 64 * 64; idx += kThreads) {
      const int k = idx >> 6, j = idx & 63;
      Vl[k * CLD + j] = (k < n && j < n) ? V1[(int64_t)k * p.ldv1 + j] : T(0);
    }
    __syncthreads();
  }
  Acc G[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) G[i] = M::zero();
  int64_t cb, ce;
  split_range((p.rows + 15) / 16, p.nsplit, split, cb, ce);
  const bool al = ((p.ldm & 3) == 0) && ((reinterpret_cast<uintptr_t>(Mp) & (4 * sizeof(T) - 1)) == 0);
  const int ntl = (n + 15) >> 4;  // column tiles that exist: the Gram tiles beyond them are zero (n = 32: 3 of the 10 tile pairs)
  auto gram_update = [&](const Acc (&mw)[4]) {
    int idx = 0;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = ti; tj < 4; ++tj) {
        if (tj < ntl) {  // (wave-uniform)
#pragma unroll
          for (int r = 0; r < 4; ++r) G[idx] = M::mma(mw[ti][r], mw[tj][r], G[idx]);
        }
        ++idx;
      }
  };
  if constexpr (!ROT) {
    // lane (g, cl), register r <-> M[r0 + 4 g + r][16 t + cl]; the next slab's loads are issued before this slab's MFMAs (the
    // loop has no stores: the loads of two slabs per wave are in flight -- four workgroups per CU do not cover an HBM round trip)
    auto load_slab = [&](int64_t c, Acc (&mw)[4]) {
      const int64_t r0 = c * 16;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = r0 + 4 * g + r;
          const int col = 16 * t + cl;
          mw[t][r] = (row < p.rows && col < n) ? Mp[row * p.ldm + col] : T(0);
        }
    };
    if (ntl <= 2) {  // (measured: at n = 32 the prefetch gives 2.6 -> 4.4 TB/s together with the tile skip; at n = 64 it costs 7 %)
      Acc cur[4], nxt[4];
      int64_t c = cb + wave;
      if (c < ce) load_slab(c, cur);
      for (; c < ce; c += 4) {
        const bool more = c + 4 < ce;
        if (more) load_slab(c + 4, nxt);
        gram_update(cur);
        if (more) {
#pragma unroll
          for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
        }
      }
    } else {
      for (int64_t c = cb + wave; c < ce; c += 4) {
        Acc mw[4];
        load_slab(c, mw);
        gram_update(mw);
      }
    }
  } else {
    for (int64_t c = cb + wave; c < ce; c += 4) {
      const int64_t r0 = c * 16;
      Acc mw[4];
      // A operand: lane (i = cl, g) holds M[r0 + cl][16 kk + 4 g + u]; K step (kk, u) <-> k = 16 kk + 4 g + u
      T a[4][4];
      const int64_t row = r0 + cl;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int col = 16 * kk + 4 * g;
        const Pack<T, 4> x = load_pack<T, 4>(Mp + row * p.ldm + col, al, row < p.rows ? n - col : 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) a[kk][u] = x.v[u];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        mw[t] = M::zero();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int u = 0; u < 4; ++u) mw[t] = M::mma(a[kk][u], Vl[(16 * kk + 4 * g + u) * CLD + 16 * t + cl], mw[t]);
      }
      gram_update(mw);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) smem[(wave * 10 + i) * 256 + lane * 4 + r] = G[i][r];
  __syncthreads();
  T* __restrict__ Gout = p.G + ((int64_t)b * p.nsplit + split) * n * n;
  for (int idx = tid; idx < 10 * 256; idx += kThreads) {
    const int tile = idx >> 8, e = idx & 255, ln = e >> 2, r = e & 3;
    const T v = (smem[idx] + smem[2560 + idx]) + (smem[5120 + idx] + smem[7680 + idx]);
    int ti = 0, rem = tile;
    while (rem >= 4 - ti) { rem -= 4 - ti; ++ti; }
    const int tj = ti + rem;
    const int row = 16 * ti + M::row(ln, r), col = 16 * tj + (ln & 15);
    if (row < n && col < n) {
      Gout[row * n + col] = v;
      if (ti != tj) Gout[col * n + row] = v;
    }
  }
}

// out[b][e] = sum_p in[b][p][e]  (fixed order: deterministic)
template <typename T>
__global__ __launch_bounds__(kThreads) void sum_parts_kernel(const T* __restrict__ in, T* __restrict__ out, int parts, int64_t count) {
  const int64_t b = blockIdx.y;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < count; e += (int64_t)gridDim.x * kThreads) {
    T s = T(0);
    for (int q = 0; q < parts; ++q) s += in[((int64_t)b * parts + q) * count + e];
    out[b * count + e] = s;
  }
}

// HASV1 = false (U = V2[:, :ro] given: no prologue product): the kernel keeps 17 KB of LDS instead of 55 KB -- six workgroups per
// CU instead of two.  Its main loop cannot prefetch (the step's stores sit between the loads and their use in one in-order
// counter, see project_kernel), so the HBM round trip is covered by resident waves alone; the host shim forms U with one small
// GEMM for tall inputs.
template <typename T, bool HASV1>
__global__ __launch_bounds__(kThreads) void colproject_kernel(ColArgs<T> p) {
  using M = Mfma<T>;
  using Acc = typename M::Acc;
  constexpr int V1LD = 66;
  __shared__ __attribute__((aligned(16))) T V2l[HASV1 ? 64 * KLD : 4];   // V2[:, :ro] as [m][i] (prologue only)
  __shared__ __attribute__((aligned(16))) T V1l[HASV1 ? 64 * V1LD : 4];  // V1 as [k][m] (prologue only)
  __shared__ __attribute__((aligned(16))) T Ul[64 * CLD];    // U = V1 V2[:, :ro] as [k][i], zero padded
  __shared__ T csc[64];                                      // column scale of left (1 / sigma or 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 15, g = lane >> 4;
  const int64_t b = blockIdx.y;
  const int split = blockIdx.x;
  const T* __restrict__ Mp = p.M + b * p.strideM;
  const int n = p.n, ro = p.ro;
  const int nt = (ro + 15) / 16;
  const T* __restrict__ V2 = p.V2 + b * p.strideV2;
  const T* __restrict__ sg = p.sigma ? p.sigma + b * p.stride_sigma : nullptr;
  T* __restrict__ Rt = (p.right && split == 0) ? p.right + b * p.strideR : nullptr;
  if constexpr (HASV1) {
    for (int idx = tid; idx < 64 * 64; idx += kThreads) {
      const int m = idx >> 6, i = idx & 63;
      V2l[m * KLD + i] = (m < n && i < ro) ? V2[(int64_t)m * p.ldv2 + i] : T(0);
    }
    const T* __restrict__ V1 = p.V1 + b * p.strideV1;
    for (int idx = tid; idx < 64 * 64; idx += kThreads) {
      const int k = idx >> 6, m = idx & 63;
      V1l[k * V1LD + m] = (k < n && m < n) ? V1[(int64_t)k * p.ldv1 + m] : T(0);
    }
  } else {
    for (int idx = tid; idx < 64 * 64; idx += kThreads) {
      const int k = idx >> 6, i = idx & 63;
      Ul[k * CLD + i] = (k < n && i < ro) ? V2[(int64_t)k * p.ldv2 + i] : T(0);
    }
  }
  if (tid < 64) {
    T sc = T(1);
    if (p.left_ortho && sg && tid < ro) {
      const T x = sg[tid];
      sc = (fabs((double)x) < (double)Num<T>::tiny()) ? T(0) : T(1) / x;
    }
    csc[tid] = sc;
  }
  __syncthreads();
  if constexpr (HASV1) {  // wave w: rows 16 w .. 16 w + 15 of U, straight from the accumulators to LDS
    Acc u0 = M::zero(), u1 = M::zero(), u2 = M::zero(), u3 = M::zero();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const T a = V1l[(16 * wave + cl) * V1LD + 4 * ks + g];
      u0 = M::mma(a, V2l[(4 * ks + g) * KLD + cl], u0);
      u1 = M::mma(a, V2l[(4 * ks + g) * KLD + 16 + cl], u1);
      u2 = M::mma(a, V2l[(4 * ks + g) * KLD + 32 + cl], u2);
      u3 = M::mma(a, V2l[(4 * ks + g) * KLD + 48 + cl], u3);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T* dst = Ul + (16 * wave + M::row(lane, r)) * CLD + cl;
      dst[0] = u0[r];
      dst[16] = u1[r];
      dst[32] = u2[r];
      dst[48] = u3[r];
    }
    __syncthreads();
  }
  if (Rt) {  // right = [diag(sigma)] U^T, coalesced, from the LDS image
    for (int idx = tid; idx < ro * n; idx += kThreads) {
      const int i = idx / n, k = idx - i * n;
      const T u = Ul[k * CLD + i];
      Rt[(int64_t)i * p.ldr + k] = (p.left_ortho && sg) ? u * sg[i] : u;
    }
  }
  T* __restrict__ Lo = p.left + b * p.strideL;
  const bool al = ((p.ldm & 3) == 0) && ((reinterpret_cast<uintptr_t>(Mp) & (4 * sizeof(T) - 1)) == 0);
  int64_t cb, ce;
  split_range((p.rows + 15) / 16, p.nsplit, split, cb, ce);
  for (int64_t c = cb + wave; c < ce; c += 4) {
    const int64_t r0 = c * 16;
    T a[4][4];
    const int64_t row = r0 + cl;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int col = 16 * kk + 4 * g;
      const Pack<T, 4> x = load_pack<T, 4>(Mp + row * p.ldm + col, al, row < p.rows ? n - col : 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) a[kk][u] = x.v[u];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < nt) {
        Acc acc = M::zero();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = M::mma(a[kk][u], Ul[(16 * kk + 4 * g + u) * CLD + 16 * t + cl], acc);
        const int col = 16 * t + cl;
        const T sc = csc[col < 64 ? col : 0];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t orow = r0 + M::row(lane, r);
          if (orow < p.rows && col < ro) Lo[orow * p.ldl + col] = acc[r] * sc;
        }
      }
    }
  }
}

static int col_split(int64_t rows, int64_t batch) {
  const int64_t chunks = (rows + 15) / 16;
  int64_t want = (4096 + batch - 1) / batch;
  int64_t maxs = chunks / 16;  // >= 4 steps per wave
  if (maxs < 1) maxs = 1;
  int64_t s = want < maxs ? want : maxs;
  if (s > 4096) s = 4096;
  return (int)(s < 1 ? 1 : s);
}

int64_t colgram_workspace_bytes(int dtype, int64_t rows, int64_t n, int64_t batch) {
  const int sp = col_split(rows, batch);
  return sp > 1 ? (int64_t)sp * batch * n * n * (dtype == TTR_F64 ? 8 : 4) : 0;
}

template <typename T>
static int colgram_typed(int64_t rows, int64_t n, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM,
                         const void* V1, int64_t ldv1, int64_t strideV1, void* G, void* ws, int64_t ws_bytes,
                         hipStream_t stream, const int32_t* skip) {
  const int sp = col_split(rows, batch);
  TTR_REQUIRE(sp == 1 || (ws && ws_bytes >= (int64_t)sp * batch * n * n * (int64_t)sizeof(T)), TTR_E_WORKSPACE,
              "ttr_colgram: workspace too small");
  ColArgs<T> p{};
  p.rows = rows; p.n = (int)n; p.M = (const T*)Mx; p.ldm = ldm; p.strideM = strideM;
  p.V1 = (const T*)V1; p.ldv1 = ldv1; p.strideV1 = strideV1;
  p.G = sp == 1 ? (T*)G : (T*)ws; p.nsplit = sp;
  ProfScope prof(V1 ? TTR_PROF_ROTGRAM : TTR_PROF_ROWGRAM, stream);
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
    ColArgs<T> q = p;
    q.M = p.M + b0 * strideM;
    if (V1) q.V1 = p.V1 + b0 * strideV1;
    q.G = p.G + b0 * sp * n * n;
    q.skip = skip ? skip + b0 : nullptr;
    const dim3 grid((unsigned)sp, (unsigned)nb);
    if (V1) hipLaunchKernelGGL((colgram_kernel<T, true>), grid, dim3(kThreads), 0, stream, q);
    else hipLaunchKernelGGL((colgram_kernel<T, false>), grid, dim3(kThreads), 0, stream, q);
    if (sp > 1) {
      int64_t gx = ceil_div(n * n, kThreads);
      hipLaunchKernelGGL(sum_parts_kernel<T>, dim3((unsigned)gx, (unsigned)nb), dim3(kThreads), 0, stream,
                         (const T*)q.G, (T*)G + b0 * n * n, sp, n * n);
    }
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int colgram_dispatch(int dtype, int64_t rows, int64_t n, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM,
                     const void* V1, int64_t ldv1, int64_t strideV1, void* G, void* ws, int64_t ws_bytes, hipStream_t stream,
                     const int32_t* skip) {
  TTR_REQUIRE(n >= 1 && n <= 64, TTR_E_UNSUPPORTED, "ttr_colgram: %lld columns (the fused kernels hold <= 64)", (long long)n);
  if (dtype == TTR_F32) return colgram_typed<float>(rows, n, batch, Mx, ldm, strideM, V1, ldv1, strideV1, G, ws, ws_bytes, stream, skip);
  return colgram_typed<double>(rows, n, batch, Mx, ldm, strideM, V1, ldv1, strideV1, G, ws, ws_bytes, stream, skip);
}

template <typename T>
static int colproject_typed(int64_t rows, int64_t n, int64_t ro, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM,
                            const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2, int64_t strideV2,
                            const void* sigma, int64_t stride_sigma, int left_ortho, void* left, int64_t ldl,
                            int64_t strideL, void* right, int64_t ldr, int64_t strideR, hipStream_t stream) {
  ColArgs<T> p{};
  p.rows = rows; p.n = (int)n; p.M = (const T*)Mx; p.ldm = ldm; p.strideM = strideM;
  p.V1 = (const T*)V1; p.ldv1 = ldv1; p.strideV1 = strideV1;
  p.V2 = (const T*)V2; p.ldv2 = ldv2; p.strideV2 = strideV2;
  p.sigma = (const T*)sigma; p.stride_sigma = stride_sigma; p.ro = (int)ro; p.left_ortho = left_ortho;
  p.left = (T*)left; p.ldl = ldl; p.strideL = strideL;
  p.right = (T*)right; p.ldr = ldr; p.strideR = strideR;
  p.nsplit = col_split(rows, batch);
  ProfScope prof(TTR_PROF_PROJECT, stream);
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
    ColArgs<T> q = p;
    q.M = p.M + b0 * strideM;
    if (V1) q.V1 = p.V1 + b0 * strideV1;
    q.V2 = p.V2 + b0 * strideV2;
    if (sigma) q.sigma = p.sigma + b0 * stride_sigma;
    q.left = p.left + b0 * strideL;
    if (right) q.right = p.right + b0 * strideR;
    if (q.V1) hipLaunchKernelGGL((colproject_kernel<T, true>), dim3((unsigned)p.nsplit, (unsigned)nb), dim3(kThreads), 0, stream, q);
    else hipLaunchKernelGGL((colproject_kernel<T, false>), dim3((unsigned)p.nsplit, (unsigned)nb), dim3(kThreads), 0, stream, q);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int colproject_dispatch(int dtype, int64_t rows, int64_t n, int64_t ro, int64_t batch, const void* Mx, int64_t ldm,
                        int64_t strideM, const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2,
                        int64_t strideV2, const void* sigma, int64_t stride_sigma, int left_ortho, void* left, int64_t ldl,
                        int64_t strideL, void* right, int64_t ldr, int64_t strideR, hipStream_t stream) {
  TTR_REQUIRE(n >= 1 && n <= 64 && ro >= 1 && ro <= n, TTR_E_UNSUPPORTED,
              "ttr_colproject: %lld columns / %lld kept (the fused kernel holds <= 64 columns)", (long long)n, (long long)ro);
  if (dtype == TTR_F32)
    return colproject_typed<float>(rows, n, ro, batch, Mx, ldm, strideM, V1, ldv1, strideV1, V2, ldv2, strideV2, sigma,
                                   stride_sigma, left_ortho, left, ldl, strideL, right, ldr, strideR, stream);
  return colproject_typed<double>(rows, n, ro, batch, Mx, ldm, strideM, V1, ldv1, strideV1, V2, ldv2, strideV2, sigma,
                                  stride_sigma, left_ortho, left, ldl, strideL, right, ldr, strideR, stream);
}

int g_sweep_stagger = 1;   // ttr_debug_set_knob(TTR_KNOB_SWEEP_STAGGER)

static int pick_split(int64_t n, int64_t batch) {
  const int64_t chunks = (n + 31) / 32;        // (the projection walks 32 columns per step and wave)
  int64_t want = (2048 + batch - 1) / batch;  // aim at >= 2048 workgroups on the 256 CUs ...
  int64_t maxs = chunks / 8;                  // ... but keep >= 2 steps per wave
  if (maxs < 1) maxs = 1;
  int64_t s = want < maxs ? want : maxs;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

int sweep_gram_parts(int64_t n, int64_t batch) { return pick_split(n, batch); }

template <typename T>
static int gram_typed(int64_t R, int64_t n, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM, const void* V1,
                      int64_t ldv1, int64_t strideV1, void* G, int64_t nsplit, hipStream_t stream, const int32_t* skip,
                      const int32_t* rows32) {
  SweepArgs<T> p{};
  p.skip = skip;
  p.rows32 = rows32;
  p.R = (int)R; p.n = n; p.M = (const T*)Mx; p.ldm = ldm; p.strideM = strideM;
  p.V1 = (const T*)V1; p.ldv1 = ldv1; p.strideV1 = strideV1;
  p.G = (T*)G; p.nsplit = (int)nsplit;
  p.stagger = g_sweep_stagger;
  if (work_census_on()) {
    // executed work per item: rows32 items load / multiply 32 of the R rows; pass-through items (skip) of the rotated pass do
    // nothing.  Gram: 10 of the 16 tiles (symmetry); rotation V1^T M: R x Reff x n.
    const double s = (double)sizeof(T), sym = 10.0 / 16.0;
    auto gr = [&](double Re) { return 2.0 * Re * Re * (double)n * sym + (V1 ? 2.0 * (double)R * Re * (double)n : 0.0); };
    const double fl[4] = {gr((double)R), gr(R > 32 ? 32.0 : (double)R), 0.0, 0.0};
    const double by[4] = {s * R * (double)n, s * (R > 32 ? 32.0 : (double)R) * (double)n, 0.0, 0.0};
    work_items(V1 ? TTR_PROF_ROTGRAM : TTR_PROF_ROWGRAM, rows32, skip, batch, fl, by, stream);
  }
  ProfScope prof(V1 ? TTR_PROF_ROTGRAM : TTR_PROF_ROWGRAM, stream);
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
    SweepArgs<T> q = p;
    q.M = p.M + b0 * strideM;
    if (V1) q.V1 = p.V1 + b0 * strideV1;
    q.G = p.G + b0 * nsplit * R * R;
    if (skip) q.skip = skip + b0;
    if (rows32) q.rows32 = rows32 + b0;
    const dim3 grid((unsigned)nsplit, (unsigned)nb);
    if (V1) hipLaunchKernelGGL((rotgram_kernel<T, false>), grid, dim3(kThreads), 0, stream, q);
    else hipLaunchKernelGGL((rotgram_kernel<T, true>), grid, dim3(kThreads), 0, stream, q);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int sweep_gram_dispatch(int dtype, int64_t R, int64_t n, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM,
                        const void* V1, int64_t ldv1, int64_t strideV1, void* G, int64_t nsplit, hipStream_t stream,
                        const int32_t* skip, const int32_t* rows32) {
  TTR_REQUIRE(R >= 1 && R <= 64, TTR_E_UNSUPPORTED, "ttr_rowgram / ttr_rotgram: %lld rows (the fused kernels hold <= 64)",
              (long long)R);
  TTR_REQUIRE(nsplit >= 1 && nsplit <= 65535, TTR_E_INVALID, "ttr_rowgram / ttr_rotgram: bad split %lld", (long long)nsplit);
  if (dtype == TTR_F32) return gram_typed<float>(R, n, batch, Mx, ldm, strideM, V1, ldv1, strideV1, G, nsplit, stream, skip, rows32);
  return gram_typed<double>(R, n, batch, Mx, ldm, strideM, V1, ldv1, strideV1, G, nsplit, stream, skip, rows32);
}

template <typename T>
static int project_typed(int64_t R, int64_t n, int64_t ro, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM,
                         const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2, int64_t strideV2,
                         const void* sigma, int64_t stride_sigma, int scale_right, void* right, int64_t ldr,
                         int64_t strideR, void* left, int64_t ldl, int64_t strideL, hipStream_t stream, const int32_t* rows32) {
  SweepArgs<T> p{};
  p.rows32 = rows32;
  p.R = (int)R; p.n = n; p.M = (const T*)Mx; p.ldm = ldm; p.strideM = strideM;
  p.V1 = (const T*)V1; p.ldv1 = ldv1; p.strideV1 = strideV1;
  p.V2 = (const T*)V2; p.ldv2 = ldv2; p.strideV2 = strideV2;
  p.sigma = (const T*)sigma; p.stride_sigma = stride_sigma; p.ro = (int)ro; p.scale_right = scale_right;
  p.right = (T*)right; p.ldr = ldr; p.strideR = strideR;
  p.left = (T*)left; p.ldl = ldl; p.strideL = strideL;
  p.nsplit = pick_split(n, batch);
  p.stagger = g_sweep_stagger;
  if (work_census_on()) {   // right = U^T M: ro x Reff x n; reads Reff rows, writes ro rows
    const double s = (double)sizeof(T), Re = R > 32 ? 32.0 : (double)R;
    const double fl[4] = {2.0 * R * (double)ro * (double)n, 2.0 * Re * (double)ro * (double)n, 0.0, 0.0};
    const double by[4] = {s * (R + ro) * (double)n, s * (Re + ro) * (double)n, 0.0, 0.0};
    work_items(TTR_PROF_PROJECT, rows32, nullptr, batch, fl, by, stream);
  }
  ProfScope prof(TTR_PROF_PROJECT, stream);
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
    SweepArgs<T> q = p;
    q.M = p.M + b0 * strideM;
    if (V1) q.V1 = p.V1 + b0 * strideV1;
    q.V2 = p.V2 + b0 * strideV2;
    if (sigma) q.sigma = p.sigma + b0 * stride_sigma;
    q.right = p.right + b0 * strideR;
    if (left) q.left = p.left + b0 * strideL;
    if (rows32) q.rows32 = rows32 + b0;
    hipLaunchKernelGGL(project_kernel<T>, dim3((unsigned)p.nsplit, (unsigned)nb), dim3(kThreads), 0, stream, q);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int sweep_project_dispatch(int dtype, int64_t R, int64_t n, int64_t ro, int64_t batch, const void* Mx, int64_t ldm,
                           int64_t strideM, const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2,
                           int64_t strideV2, const void* sigma, int64_t stride_sigma, int scale_right, void* right,
                           int64_t ldr, int64_t strideR, void* left, int64_t ldl, int64_t strideL, hipStream_t stream,
                           const int32_t* rows32) {
  TTR_REQUIRE(R >= 1 && R <= 64 && ro >= 1 && ro <= R, TTR_E_UNSUPPORTED,
              "ttr_project: %lld rows / %lld kept (the fused kernel holds <= 64 rows)", (long long)R, (long long)ro);
  if (dtype == TTR_F32)
    return project_typed<float>(R, n, ro, batch, Mx, ldm, strideM, V1, ldv1, strideV1, V2, ldv2, strideV2, sigma,
                                stride_sigma, scale_right, right, ldr, strideR, left, ldl, strideL, stream, rows32);
  return project_typed<double>(R, n, ro, batch, Mx, ldm, strideM, V1, ldv1, strideV1, V2, ldv2, strideV2, sigma,
                               stride_sigma, scale_right, right, ldr, strideR, left, ldl, strideL, stream, rows32);
}

}  // namespace ttr
