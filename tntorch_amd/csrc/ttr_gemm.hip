// Batched small/tall-skinny GEMM on MFMA for the TT sweeps (gfx950).
//
//   C[b] = scale( op(A[b]) (M x K) * op(B[b]) (K x N) )
//
// Every GEMM of the rounding path has one small dimension (a TT rank, <= ~128) and
// one long one (I * R), so the kernels are HBM/L2-bound: the design goal is coalesced
// streaming of the long operand, not peak MFMA rate.  One workgroup (4 wave64) owns a
// 64 x 64 tile of C; K is walked in steps of 16 through LDS; each wave owns a 32 x 32
// sub-tile = 2 x 2 MFMA 16x16x4 accumulators (exact f32 / f64 arithmetic).
//
// Operands are addressed through (row stride, col stride) pairs so that both
// transposes come for free; the LDS image of a tile is chosen per operand so that the
// global read is coalesced along the unit-stride axis AND the MFMA fragment reads are
// bank-conflict free:
//   k-contiguous operand  -> LDS [mn][k], leading dim BK+1 (fragment read: (BK+1)*i + k)
//   mn-contiguous operand -> LDS [k][mn], leading dim 80   (fragment read: 80*k + i)
// Optional split-K (blockIdx.y) writes partial tiles to a workspace, reduced (and
// scaled) by a second kernel -- used for the Gram matrices of very tall unfoldings.
#include "ttr_common.h"

namespace ttr {

constexpr int BK = 16;  // K step; 32 measured slower on every shape of the path (metric GEMMs 12.8 -> 17.1 ms/step); two LDS
                        // stages with one barrier per step: 12.6 -> 13.2 ms/step; a K <= 64 variant staging the whole K extent at once
                        // (one barrier, all loads in flight): 12.7 -> 17.5 ms/step -- occupancy matters more (all measured, not kept)
constexpr int BKL = 4;  // log2(BK)
// Tile shapes: every wave always owns a 32 x 32 block of C (2 x 2 MFMA tiles); the four waves are laid out
// 2 x 2 (64 x 64 tile), 4 x 1 (128 x 32: products with <= 32 columns, e.g. X[I^3, I] @ A[I, R] of the MTTKRP or
// the rank-32 projections) or 1 x 4 (32 x 128: <= 32 rows).  With the square tile such products waste half of
// the MFMA issue slots -- and fp32 MFMA runs at the vector rate, so that half decides whether the kernel is
// HBM- or issue-bound.
constexpr int lds_tile(int bm) { return bm * (BK + 1) > BK * (bm + 16) ? bm * (BK + 1) : BK * (bm + 16); }

template <typename T>
struct GemmArgs {
  int64_t M, N, K;
  const T* A;
  int64_t a_rs, a_cs, strideA;  // op(A)(i,k) = A[i * a_rs + k * a_cs]
  const T* B;
  int64_t b_rs, b_cs, strideB;  // op(B)(k,j) = B[k * b_rs + j * b_cs]
  T* C;
  int64_t ldc, strideC;
  const T* rs;
  int64_t stride_rs;
  int rs_mode;
  const T* cs;
  int64_t stride_cs;
  int cs_mode;
  int tilesN;
  int nsplit;
  int64_t k_chunk;  // K range per split (multiple of BK)
  T* part;          // [nsplit][batch][M][N] when nsplit > 1
  int64_t batch;
  int axpby;        // C <- beta * C + alpha * op(A) op(B)   (ttr_gemm_axpby)
  T alpha, beta;
};

template <typename T>
__device__ __forceinline__ T apply_scale(T v, const T* s, int64_t idx, int mode) {
  if (mode == TTR_SCALE_NONE) return v;
  T x = s[idx];
  if (mode == TTR_SCALE_MUL) return v * x;
  return (fabs((double)x) < (double)Num<T>::tiny()) ? T(0) : v / x;
}

template <typename T, int BM, int BN, int WN>
__global__ __launch_bounds__(kThreads) void gemm_kernel(GemmArgs<T> p) {
  constexpr int EA = BM * BK / kThreads, EB = BN * BK / kThreads;  // staged elements per thread and K step
  __shared__ T As[lds_tile(BM)];
  __shared__ T Bs[lds_tile(BN)];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t b = blockIdx.z;
  const int tm_idx = blockIdx.x / p.tilesN;
  const int tn_idx = blockIdx.x % p.tilesN;
  const int64_t m0 = (int64_t)tm_idx * BM;
  const int64_t n0 = (int64_t)tn_idx * BN;
  const int split = blockIdx.y;
  const int64_t k_begin = (int64_t)split * p.k_chunk;
  const int64_t k_end = (k_begin + p.k_chunk < p.K) ? (k_begin + p.k_chunk) : p.K;

  const T* __restrict__ A = p.A + b * p.strideA;
  const T* __restrict__ B = p.B + b * p.strideB;

  const bool a_kc = (p.a_cs == 1);   // op(A) contiguous along k
  const bool b_kc = (p.b_rs == 1);   // op(B) contiguous along k
  const int sa_i = a_kc ? BK + 1 : 1, sa_k = a_kc ? 1 : BM + 16;
  const int sb_j = b_kc ? BK + 1 : 1, sb_k = b_kc ? 1 : BN + 16;

  // per-thread staging coordinates (EA / EB elements of the operands per K step)
  int ai[EA], ak[EA], bj[EB], bk[EB];
#pragma unroll
  for (int e = 0; e < EA; ++e) {
    const int idx = tid + kThreads * e;
    if (a_kc) { ak[e] = idx & (BK - 1); ai[e] = idx >> BKL; } else { ai[e] = idx % BM; ak[e] = idx / BM; }
  }
#pragma unroll
  for (int e = 0; e < EB; ++e) {
    const int idx = tid + kThreads * e;
    if (b_kc) { bk[e] = idx & (BK - 1); bj[e] = idx >> BKL; } else { bj[e] = idx % BN; bk[e] = idx / BN; }
  }

  typename Mfma<T>::Acc acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = Mfma<T>::zero();

  T ra[EA], rb[EB];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      const int64_t gi = m0 + ai[e], gk = k0 + ak[e];
      ra[e] = (gi < p.M && gk < k_end) ? A[gi * p.a_rs + gk * p.a_cs] : T(0);
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const int64_t gj = n0 + bj[e], gk2 = k0 + bk[e];
      rb[e] = (gj < p.N && gk2 < k_end) ? B[gk2 * p.b_rs + gj * p.b_cs] : T(0);
    }
  };

  if (k_begin < k_end) fetch(k_begin);
  for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
    for (int e = 0; e < EA; ++e) As[ai[e] * sa_i + ak[e] * sa_k] = ra[e];
#pragma unroll
    for (int e = 0; e < EB; ++e) Bs[bj[e] * sb_j + bk[e] * sb_k] = rb[e];
    __syncthreads();
    if (k0 + BK < k_end) fetch(k0 + BK);  // overlaps the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int kf = kk * 4 + (lane >> 4);
      T a[2], bb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[(wm * 32 + t * 16 + (lane & 15)) * sa_i + kf * sa_k];
        bb[t] = Bs[(wn * 32 + t * 16 + (lane & 15)) * sb_j + kf * sb_k];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mfma<T>::mma(a[i], bb[j], acc[i][j]);
    }
    __syncthreads();
  }

  const bool direct = (p.nsplit == 1);
  T* __restrict__ C = direct ? (p.C + b * p.strideC) : (p.part + ((int64_t)split * p.batch + b) * p.M * p.N);
  const int64_t ldc = direct ? p.ldc : p.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * 32 + i * 16 + Mfma<T>::row(lane, r);
        const int64_t col = n0 + wn * 32 + j * 16 + (lane & 15);
        if (row < p.M && col < p.N) {
          T v = acc[i][j][r];
          if (direct) {
            v = apply_scale(v, p.rs ? p.rs + b * p.stride_rs : nullptr, row, p.rs ? p.rs_mode : TTR_SCALE_NONE);
            v = apply_scale(v, p.cs ? p.cs + b * p.stride_cs : nullptr, col, p.cs ? p.cs_mode : TTR_SCALE_NONE);
            if (p.axpby) v = p.alpha * v + (p.beta != T(0) ? p.beta * C[row * ldc + col] : T(0));
          }
          C[row * ldc + col] = v;
        }
      }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void splitk_reduce_kernel(GemmArgs<T> p) {
  const int64_t b = blockIdx.y;
  const int64_t mn = p.M * p.N;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < mn; idx += (int64_t)gridDim.x * kThreads) {
    T s = 0;
    for (int sp = 0; sp < p.nsplit; ++sp) s += p.part[((int64_t)sp * p.batch + b) * mn + idx];
    const int64_t row = idx / p.N, col = idx % p.N;
    s = apply_scale(s, p.rs ? p.rs + b * p.stride_rs : nullptr, row, p.rs ? p.rs_mode : TTR_SCALE_NONE);
    s = apply_scale(s, p.cs ? p.cs + b * p.stride_cs : nullptr, col, p.cs ? p.cs_mode : TTR_SCALE_NONE);
    T* dst = p.C + b * p.strideC + row * p.ldc + col;
    if (p.axpby) s = p.alpha * s + (p.beta != T(0) ? p.beta * *dst : T(0));
    *dst = s;
  }
}

// ---------------------------------------------------------------------------------------------- compute-shaped products
// fp32 products with BOTH output dimensions >= 128 -- the n = I r = 1024 Gram matrix, rotation and rotated Gram matrix of
// the second step of a dense 64^k TT-SVD (BASELINE config C1: 3.5e13 flop, fp32-MFMA-bound), the n = 256 bonds
// of config C3 -- are not HBM-bound, and the 64 x 64 tile above spends as long on its two barriers, sixteen dword loads
// and LDS traffic per K step as on its 16 MFMAs per wave (measured ~60 TFLOP/s).  This kernel owns a 128 x 128 tile per
// workgroup: every wave a 64 x 64 quadrant = 4 x 4 accumulators, 64 MFMAs (2048 cycles of matrix pipe) per K step of 16
// against 4 sixteen-byte global loads per thread, staged through one LDS image per operand:
//   k-contiguous operand  -> [mn][20]: 16-byte writes, and ONE 16-byte read hands a lane the four K steps of its fragment
//                            (MFMA step s of lane group g consumes k = 4 g + s -- any bijection of the 16 k's onto
//                            (step, group) is a valid K order as long as both operands use the same one)
//   mn-contiguous operand -> [k][144]: 16-byte writes (512 contiguous bytes per 32 lanes from HBM), b32 fragment reads
// Symmetric products (A^T A, A A^T: `sym`) only run the tiles on and above the diagonal and mirror them on output.
// Workgroup -> (tile, split) map: the dispatcher places consecutive workgroups on consecutive XCDs (observed, used for
// speed only), so the workgroups that read the same operand rows at the same time -- all tiles of one K split, or the
// column tiles of one row panel -- are given ids that are congruent modulo 8 and adjacent: their common operand is
// fetched from HBM once per XCD and served from that XCD's L2 to the others.
constexpr int GBT = 128;   // tile edge
constexpr int GLDK = 20;   // [mn][k] image: 16 k + 4 pad
constexpr int GLDM = 144;  // [k][mn] image: 128 mn + 16 pad

struct BigMap {
  int inner, outer;  // ids: j = L / 8, in = j % inner, o = (j / inner) * 8 + L % 8 (workgroups with o >= outer exit)
  int split_mode;    // 1: in = tile, o = K split;  0: in = column tile, o = row tile
  int sym, tiles1d;  // sym: tile list = upper triangle of a tiles1d x tiles1d grid
};

template <bool AKC, bool BKC>
__global__ __launch_bounds__(kThreads, 2) void gemm_big_kernel(GemmArgs<float> p, BigMap mp) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float As[AKC ? GBT * GLDK : BK * GLDM];
  __shared__ __attribute__((aligned(16))) float Bs[BKC ? GBT * GLDK : BK * GLDM];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cl = lane & 15, g = lane >> 4;
  const int wm = wv >> 1, wn = wv & 1;
  const int64_t b = blockIdx.z;
  const int L = blockIdx.x, xcd = L & 7, jj = L >> 3;
  const int in = jj % mp.inner, o = (jj / mp.inner) * 8 + xcd;
  if (o >= mp.outer) return;
  int tile, split;
  if (mp.split_mode) { tile = in; split = o; } else { tile = o * mp.inner + in; split = 0; }
  int tmi, tni;
  if (mp.sym) {
    int t = mp.split_mode ? tile : in + o * mp.inner;
    tmi = 0;
    while (t >= mp.tiles1d - tmi) { t -= mp.tiles1d - tmi; ++tmi; }
    tni = tmi + t;
  } else if (mp.split_mode) {
    tmi = tile / p.tilesN; tni = tile % p.tilesN;
  } else {
    tmi = o; tni = in;
  }
  const int64_t m0 = (int64_t)tmi * GBT, n0 = (int64_t)tni * GBT;
  const int64_t k_begin = (int64_t)split * p.k_chunk;
  const int64_t k_end = (k_begin + p.k_chunk < p.K) ? (k_begin + p.k_chunk) : p.K;
  const float* __restrict__ A = p.A + b * p.strideA;
  const float* __restrict__ B = p.B + b * p.strideB;

  // staging: two 16-byte pieces per operand and thread (all extents / leading dimensions are multiples of 4: a piece is
  // either entirely inside the matrix or entirely outside)
  int64_t aoff[2], boff[2];
  bool aok[2], bok[2];
  int akq[2], bkq[2], alds[2], blds[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int f = tid + kThreads * e;
    if (AKC) { const int i = f >> 2, kq = f & 3; aoff[e] = (m0 + i) * p.a_rs + 4 * kq; aok[e] = m0 + i < p.M; akq[e] = 4 * kq; alds[e] = i * GLDK + 4 * kq; }
    else { const int iq = f & 31, kk = f >> 5; aoff[e] = (int64_t)kk * p.a_cs + m0 + 4 * iq; aok[e] = m0 + 4 * iq < p.M; akq[e] = kk; alds[e] = kk * GLDM + 4 * iq; }
    if (BKC) { const int i = f >> 2, kq = f & 3; boff[e] = (n0 + i) * p.b_cs + 4 * kq; bok[e] = n0 + i < p.N; bkq[e] = 4 * kq; blds[e] = i * GLDK + 4 * kq; }
    else { const int iq = f & 31, kk = f >> 5; boff[e] = (int64_t)kk * p.b_rs + n0 + 4 * iq; bok[e] = n0 + 4 * iq < p.N; bkq[e] = kk; blds[e] = kk * GLDM + 4 * iq; }
  }
  const int64_t astep = AKC ? 1 : p.a_cs, bstep = BKC ? 1 : p.b_rs;  // element stride of one k
  f4 ra[2], rb[2];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      ra[e] = (aok[e] && k0 + akq[e] < k_end) ? *reinterpret_cast<const f4*>(A + aoff[e] + k0 * astep) : f4{0.f, 0.f, 0.f, 0.f};
      rb[e] = (bok[e] && k0 + bkq[e] < k_end) ? *reinterpret_cast<const f4*>(B + boff[e] + k0 * bstep) : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (k_begin < k_end) fetch(k_begin);
  for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      *reinterpret_cast<f4*>(&As[alds[e]]) = ra[e];
      *reinterpret_cast<f4*>(&Bs[blds[e]]) = rb[e];
    }
    __syncthreads();
    if (k0 + BK < k_end) fetch(k0 + BK);  // in flight under the 64 MFMAs below
    f4 af[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = 64 * wm + 16 * t + cl;
      if (AKC) af[t] = *reinterpret_cast<const f4*>(&As[row * GLDK + 4 * g]);
      else af[t] = f4{As[(4 * g + 0) * GLDM + row], As[(4 * g + 1) * GLDM + row], As[(4 * g + 2) * GLDM + row], As[(4 * g + 3) * GLDM + row]};
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int col = 64 * wn + 16 * tn + cl;
      f4 bf;
      if (BKC) bf = *reinterpret_cast<const f4*>(&Bs[col * GLDK + 4 * g]);
      else bf = f4{Bs[(4 * g + 0) * GLDM + col], Bs[(4 * g + 1) * GLDM + col], Bs[(4 * g + 2) * GLDM + col], Bs[(4 * g + 3) * GLDM + col]};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tm][s4], bf[s4], acc[tm][tn], 0, 0, 0);
    }
    __syncthreads();
  }

  const bool direct = (p.nsplit == 1);
  float* __restrict__ C = direct ? (p.C + b * p.strideC) : (p.part + ((int64_t)split * p.batch + b) * p.M * p.N);
  const int64_t ldc = direct ? p.ldc : p.N;
  const bool mirror = mp.sym && tmi != tni;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + 64 * wm + 16 * i + 4 * g + r;
        const int64_t col = n0 + 64 * wn + 16 * j + cl;
        if (row < p.M && col < p.N) {
          float v = acc[i][j][r], vt = v;
          if (direct) {
            v = apply_scale(v, p.rs ? p.rs + b * p.stride_rs : nullptr, row, p.rs ? p.rs_mode : TTR_SCALE_NONE);
            v = apply_scale(v, p.cs ? p.cs + b * p.stride_cs : nullptr, col, p.cs ? p.cs_mode : TTR_SCALE_NONE);
            if (p.axpby) v = p.alpha * v + (p.beta != 0.f ? p.beta * C[row * ldc + col] : 0.f);
          }
          C[row * ldc + col] = v;
          if (mirror) {  // (symmetric products are only dispatched here without scales / axpby)
            C[col * ldc + row] = vt;
          }
        }
      }
}

// Plan of the big-tile path for a product shape; nsplit is what gemm_workspace_bytes and the dispatch must agree on.
struct BigPlan {
  bool use;
  int64_t tilesM, tilesN;
  int nsplit;
};
int g_gemm_big = 1;  // ttr_debug_set_knob(TTR_KNOB_GEMM_BIG)

static BigPlan big_plan(int dtype, int64_t M, int64_t N, int64_t K, int64_t batch, bool sym) {
  BigPlan pl{false, 0, 0, 1};
  if (!g_gemm_big || dtype != TTR_F32 || M < GBT || N < GBT || K < 64) return pl;
  if ((M | N | K) & 3) return pl;
  pl.use = true;
  pl.tilesM = ceil_div(M, GBT); pl.tilesN = ceil_div(N, GBT);
  const int64_t tiles = sym ? pl.tilesM * (pl.tilesM + 1) / 2 : pl.tilesM * pl.tilesN;
  const int64_t wgs = tiles * batch;
  if (wgs < 768 && K >= 4096) {
    // too few tiles for 256 CUs x 3 workgroups: split K, in multiples of 8 (one split per XCD id).  Among the candidates
    // that leave >= 2048 rows per split, take the one that fills whole rounds of the chip's 768 workgroup slots best
    // (36 symmetric tiles x 32 splits = 1.5 rounds ran at 75 % of the rate of x 64 = 3.0 rounds, measured)
    const int64_t maxs = K / 2048 < 256 ? K / 2048 : 256;
    int64_t best = 1;
    double best_eff = 0.0;
    for (int64_t s = 8; s <= maxs; s += 8) {
      const int64_t w = wgs * s, rounds = ceil_div(w, 768);
      const double eff = (double)w / (double)(rounds * 768);
      if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
    }
    pl.nsplit = (int)best;
  }
  return pl;
}

// Split only when the plain launch would leave most of the 256 CUs idle and K is long.
static int pick_nsplit(int64_t tiles, int64_t batch, int64_t K) {
  const int64_t wgs = tiles * batch;
  if (wgs >= 512 || K < 2048) return 1;
  int64_t want = ceil_div(1024, wgs);
  int64_t maxs = K / 512;
  if (maxs < 1) maxs = 1;
  int64_t s = want < maxs ? want : maxs;
  if (s > 256) s = 256;
  return (int)(s < 1 ? 1 : s);
}

// 0: 64 x 64 (2 x 2 waves), 1: 128 x 32 (4 x 1), 2: 32 x 128 (1 x 4)
static int pick_shape(int64_t M, int64_t N) {
  if (N <= 32 && M >= 128) return 1;
  if (M <= 32 && N >= 128) return 2;
  return 0;
}
static void tile_dims(int shape, int64_t& bm, int64_t& bn) {
  bm = shape == 1 ? 128 : (shape == 2 ? 32 : 64);
  bn = shape == 1 ? 32 : (shape == 2 ? 128 : 64);
}

template <typename T>
static int gemm_impl(int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                     int64_t strideA, const void* B, int64_t ldb, int64_t strideB, void* C, int64_t ldc,
                     int64_t strideC, const void* rs, int64_t stride_rs, int rs_mode, const void* cs,
                     int64_t stride_cs, int cs_mode, int64_t batch, void* ws, int64_t ws_bytes, hipStream_t stream,
                     int axpby, double alpha, double beta) {
  GemmArgs<T> p;
  p.axpby = axpby; p.alpha = (T)alpha; p.beta = (T)beta;
  p.M = M; p.N = N; p.K = K;
  p.A = (const T*)A; p.strideA = strideA;
  if (!transA) { p.a_rs = lda; p.a_cs = 1; } else { p.a_rs = 1; p.a_cs = lda; }
  p.B = (const T*)B; p.strideB = strideB;
  if (!transB) { p.b_rs = ldb; p.b_cs = 1; } else { p.b_rs = 1; p.b_cs = ldb; }
  p.C = (T*)C; p.ldc = ldc; p.strideC = strideC;
  p.rs = (const T*)rs; p.stride_rs = stride_rs; p.rs_mode = rs_mode;
  p.cs = (const T*)cs; p.stride_cs = stride_cs; p.cs_mode = cs_mode;
  p.batch = batch;
  if (work_census_on()) {   // dense product: every tile is computed (the symmetric big-tile path: the upper tile triangle)
    const double sfac = (sizeof(T) == 4 && A == B && lda == ldb && strideA == strideB && transA != transB && M == N && !rs && !cs && !axpby &&
                         big_plan(TTR_F32, M, N, K, batch, true).use) ? 0.5 * (1.0 + 128.0 / (double)(M > 128 ? M : 128)) : 1.0;
    const double fl[4] = {2.0 * (double)M * (double)N * (double)K * sfac, 0, 0, 0};
    const double by[4] = {(double)sizeof(T) * ((double)M * K * (A == B ? 1.0 : 1.0) + (A == B ? 0.0 : (double)K * N) + (double)M * N), 0, 0, 0};
    work_items(TTR_PROF_GEMM, nullptr, nullptr, batch, fl, by, stream);
  }
  if constexpr (sizeof(T) == 4) {
    // symmetric product: the same matrix on both sides, transposed on exactly one (A^T A or A A^T), no epilogue
    const bool sym = A == B && lda == ldb && strideA == strideB && transA != transB && M == N && !rs && !cs && !axpby;
    const BigPlan pl = big_plan(TTR_F32, M, N, K, batch, sym);
    const bool aligned = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && lda % 4 == 0 && ldb % 4 == 0 &&
                         strideA % 4 == 0 && strideB % 4 == 0;
    if (pl.use && aligned && batch <= 65535) {
      int nsplit = pl.nsplit;
      if (nsplit > 1 && (!ws || ws_bytes < (int64_t)nsplit * batch * M * N * 4)) nsplit = 1;
      p.nsplit = nsplit;
      p.k_chunk = nsplit == 1 ? align_up(K, BK) : align_up(ceil_div(K, nsplit), BK);
      p.part = (T*)ws;
      p.tilesN = (int)pl.tilesN;
      BigMap mp;
      mp.sym = sym ? 1 : 0; mp.tiles1d = (int)pl.tilesM;
      const int64_t tiles = sym ? pl.tilesM * (pl.tilesM + 1) / 2 : pl.tilesM * pl.tilesN;
      mp.split_mode = (nsplit > 1 || sym) ? 1 : 0;
      if (mp.split_mode) { mp.inner = (int)tiles; mp.outer = nsplit; }
      else { mp.inner = (int)pl.tilesN; mp.outer = (int)pl.tilesM; }
      const int64_t blocks = (int64_t)mp.inner * align_up(mp.outer, 8);
      TTR_REQUIRE(blocks <= 2147483647LL, TTR_E_UNSUPPORTED, "ttr_gemm: too many tiles");
      const dim3 grid((unsigned)blocks, 1, (unsigned)batch);
      const bool akc = p.a_cs == 1, bkc = p.b_rs == 1;
      ProfScope prof(TTR_PROF_GEMM, stream);
      const GemmArgs<float>& pf = reinterpret_cast<const GemmArgs<float>&>(p);
      if (akc && bkc) hipLaunchKernelGGL((gemm_big_kernel<true, true>), grid, dim3(kThreads), 0, stream, pf, mp);
      else if (akc) hipLaunchKernelGGL((gemm_big_kernel<true, false>), grid, dim3(kThreads), 0, stream, pf, mp);
      else if (bkc) hipLaunchKernelGGL((gemm_big_kernel<false, true>), grid, dim3(kThreads), 0, stream, pf, mp);
      else hipLaunchKernelGGL((gemm_big_kernel<false, false>), grid, dim3(kThreads), 0, stream, pf, mp);
      if (nsplit > 1) {
        int64_t gx = ceil_div(M * N, kThreads);
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)gx, (unsigned)batch), dim3(kThreads), 0, stream, p);
      }
      TTR_HIP_CHECK(hipGetLastError());
      return TTR_OK;
    }
  }
  const int shape = pick_shape(M, N);
  int64_t BM, BN;
  tile_dims(shape, BM, BN);
  const int64_t tilesM = ceil_div(M, BM), tilesN = ceil_div(N, BN);
  p.tilesN = (int)tilesN;
  int nsplit = pick_nsplit(tilesM * tilesN, batch, K);
  if (nsplit > 1) {
    const int64_t need = (int64_t)nsplit * batch * M * N * (int64_t)sizeof(T);
    if (!ws || ws_bytes < need) nsplit = 1;  // no workspace: plain launch
  }
  p.nsplit = nsplit;
  p.k_chunk = nsplit == 1 ? align_up(K > 0 ? K : 1, BK) : align_up(ceil_div(K, nsplit), BK);
  p.part = (T*)ws;
  TTR_REQUIRE(batch <= 65535, TTR_E_UNSUPPORTED, "ttr_gemm: batch %lld > 65535", (long long)batch);
  TTR_REQUIRE(tilesM * tilesN <= 2147483647LL, TTR_E_UNSUPPORTED, "ttr_gemm: too many tiles");
  dim3 grid((unsigned)(tilesM * tilesN), (unsigned)nsplit, (unsigned)batch);
  {
    ProfScope prof(TTR_PROF_GEMM, stream);
    if (shape == 1)
      hipLaunchKernelGGL((gemm_kernel<T, 128, 32, 1>), grid, dim3(kThreads), 0, stream, p);
    else if (shape == 2)
      hipLaunchKernelGGL((gemm_kernel<T, 32, 128, 4>), grid, dim3(kThreads), 0, stream, p);
    else
      hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2>), grid, dim3(kThreads), 0, stream, p);
    if (nsplit > 1) {
      int64_t gx = ceil_div(M * N, kThreads);
      if (gx > 1024) gx = 1024;
      hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)gx, (unsigned)batch), dim3(kThreads), 0, stream, p);
    }
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int64_t gemm_workspace_bytes(int dtype, int64_t M, int64_t N, int64_t K, int64_t batch) {
  int64_t BM, BN;
  tile_dims(pick_shape(M, N), BM, BN);
  const int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
  int ns = pick_nsplit(tiles, batch, K);
  // the big-tile path decides its own split count; symmetric products (fewer tiles) may split further than general ones
  const int nb = big_plan(dtype, M, N, K, batch, false).nsplit, nbs = M == N ? big_plan(dtype, M, N, K, batch, true).nsplit : 1;
  if (nb > ns) ns = nb;
  if (nbs > ns) ns = nbs;
  if (ns == 1) return 0;
  return (int64_t)ns * batch * M * N * (dtype == TTR_F64 ? 8 : 4);
}

int gemm_dispatch(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                  int64_t strideA, const void* B, int64_t ldb, int64_t strideB, void* C, int64_t ldc, int64_t strideC,
                  const void* rs, int64_t stride_rs, int rs_mode, const void* cs, int64_t stride_cs, int cs_mode,
                  int64_t batch, void* ws, int64_t ws_bytes, hipStream_t stream, int axpby, double alpha, double beta) {
  if (batch > 65535) {  // the batch is a grid dimension: larger batches run in slices (the workspace of a slice fits the full one)
    const int64_t elem = dtype == TTR_F64 ? 8 : 4;
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
      const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
      const int rc = gemm_dispatch(dtype, transA, transB, M, N, K, (const char*)A + b0 * strideA * elem, lda, strideA,
                                   (const char*)B + b0 * strideB * elem, ldb, strideB, (char*)C + b0 * strideC * elem, ldc, strideC,
                                   rs ? (const char*)rs + b0 * stride_rs * elem : nullptr, stride_rs, rs_mode,
                                   cs ? (const char*)cs + b0 * stride_cs * elem : nullptr, stride_cs, cs_mode, nb, ws, ws_bytes,
                                   stream, axpby, alpha, beta);
      if (rc != TTR_OK) return rc;
    }
    return TTR_OK;
  }
  if (dtype == TTR_F32)
    return gemm_impl<float>(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, rs, stride_rs,
                            rs_mode, cs, stride_cs, cs_mode, batch, ws, ws_bytes, stream, axpby, alpha, beta);
  return gemm_impl<double>(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, rs, stride_rs,
                           rs_mode, cs, stride_cs, cs_mode, batch, ws, ws_bytes, stream, axpby, alpha, beta);
}

}  // namespace ttr
