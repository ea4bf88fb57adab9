// Fused kernels of the right-to-left truncation sweep (gfx950): everything that touches the 64 x (I r) right
// unfolding M of a core between the push-left and the new core.
//
//   ttr_rowgram   G  = M M^T                        (pass 1 of the two-pass 'svd' truncation, and 'eig')   round.py:104-109
//   ttr_rotgram   G2 = (V1^T M)(V1^T M)^T           (pass 2: Gram matrix of the ROTATED rows; the rotated matrix itself is
//                                                    never written -- it only exists 16 columns at a time in registers)
//   ttr_project   right = diag(1/sigma) (V1 V2r)^T M,  left = (V1 V2r) diag(sigma)                          round.py:163-172
//
// All three stream M once, 16 columns per wave and step, and keep every intermediate in MFMA accumulator registers:
//   * the rotated 16-column slab is computed TRANSPOSED (slab^T = M_c^T V1: A operand = 16 x 4 pieces of M read straight
//     from global memory, B operand = V1 from LDS), which leaves it in the accumulator layout
//     lane (g, cl), register r  <->  slab[row 16 t + cl][column 4 g + r];
//   * in that layout the accumulator registers ARE both MFMA operands of the Gram update
//     G[16 ti + i][16 tj + j] += sum_c slab[16 ti + i][c] slab[16 tj + j][c]  (A[i][k]: lane i + 16 k, B[k][j]: lane j + 16 k,
//     k <-> column 4 k + r): no LDS round trip, no transposition, 10 of the 16 tiles (symmetry).
// Arithmetic per 16 columns and wave: 64 + 40 MFMA 16x16x4 for the rotation + Gram (AI = 64 flop/B: bound by the fp32
// MFMA rate, which equals the vector rate on gfx950), 32 for the projection to rank 32 (AI = 11 flop/B: HBM-bound).
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
};

__host__ __device__ inline void split_range(int64_t chunks, int nsplit, int split, int64_t& c0, int64_t& c1) {
  const int64_t per = (chunks + nsplit - 1) / nsplit;
  c0 = (int64_t)split * per;
  c1 = c0 + per < chunks ? c0 + per : chunks;
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
  int64_t cb, ce;
  split_range((p.n + 15) / 16, p.nsplit, split, cb, ce);
  for (int64_t c = cb + wave; c < ce; c += 4) {
    const int64_t c0 = c * 16;
    Acc mw[4];
    if constexpr (IDENT) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * t + cl;
          const int64_t col = c0 + M::row(lane, r);
          mw[t][r] = (row < R && col < p.n) ? Mp[(int64_t)row * p.ldm + col] : T(0);
        }
    } else {
      T a[16];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int k = 4 * ks + g;
        a[ks] = (k < R && c0 + cl < p.n) ? Mp[(int64_t)k * p.ldm + c0 + cl] : T(0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        mw[t] = M::zero();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) mw[t] = M::mma(a[ks], Vl[(4 * ks + g) * KLD + 16 * t + cl], mw[t]);
      }
    }
    int idx = 0;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = ti; tj < 4; ++tj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) G[idx] = M::mma(mw[ti][r], mw[tj][r], G[idx]);
        ++idx;
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
template <typename T>
__global__ __launch_bounds__(kThreads) void project_kernel(SweepArgs<T> p) {
  using M = Mfma<T>;
  using Acc = typename M::Acc;
  __shared__ __attribute__((aligned(16))) T Ul[64 * KLD];  // U = V1 V2[:, :ro] as [k][i] (i = output row), zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 15, g = lane >> 4;
  const int64_t b = blockIdx.y;
  const int split = blockIdx.x;
  const T* __restrict__ Mp = p.M + b * p.strideM;
  const int R = p.R, ro = p.ro;
  const T* __restrict__ V2 = p.V2 + b * p.strideV2;
  const T* __restrict__ sg = p.sigma ? p.sigma + b * p.stride_sigma : nullptr;
  {
    const T* __restrict__ V1 = p.V1 ? p.V1 + b * p.strideV1 : nullptr;
    T* __restrict__ Lo = (p.left && split == 0) ? p.left + b * p.strideL : nullptr;
    for (int idx = tid; idx < 64 * 64; idx += kThreads) {
      const int k = idx >> 6, i = idx & 63;
      T u = T(0);
      if (k < R && i < ro) {
        if (V1) {
          for (int m = 0; m < R; ++m) u += V1[(int64_t)k * p.ldv1 + m] * V2[(int64_t)m * p.ldv2 + i];
        } else {
          u = V2[(int64_t)k * p.ldv2 + i];
        }
        if (Lo) Lo[(int64_t)k * p.ldl + i] = (p.scale_right && sg) ? u * sg[i] : u;
      }
      Ul[k * KLD + i] = u;
    }
    __syncthreads();
  }
  const int nt = (ro + 15) / 16;
  int64_t cb, ce;
  split_range((p.n + 15) / 16, p.nsplit, split, cb, ce);
  T* __restrict__ Ro = p.right + b * p.strideR;
  T rs[4][4];  // 1 / sigma of this lane's output rows
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * t + M::row(lane, r);
      T s = T(1);
      if (p.scale_right && sg && row < ro) {
        const T x = sg[row];
        s = (fabs((double)x) < (double)Num<T>::tiny()) ? T(0) : T(1) / x;  // TTR_SCALE_DIV semantics
      }
      rs[t][r] = s;
    }
  for (int64_t c = cb + wave; c < ce; c += 4) {
    const int64_t c0 = c * 16;
    T bm[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = 4 * ks + g;
      bm[ks] = (k < R && c0 + cl < p.n) ? Mp[(int64_t)k * p.ldm + c0 + cl] : T(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < nt) {
        Acc acc = M::zero();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = M::mma(Ul[(4 * ks + g) * KLD + 16 * t + cl], bm[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * t + M::row(lane, r);
          if (row < ro && c0 + cl < p.n) Ro[(int64_t)row * p.ldr + c0 + cl] = acc[r] * rs[t][r];
        }
      }
    }
  }
}

static int pick_split(int64_t n, int64_t batch) {
  const int64_t chunks = (n + 15) / 16;
  int64_t want = (2048 + batch - 1) / batch;  // aim at >= 2048 workgroups on the 256 CUs ...
  int64_t maxs = chunks / 16;                 // ... but keep >= 4 steps per wave
  if (maxs < 1) maxs = 1;
  int64_t s = want < maxs ? want : maxs;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

int sweep_gram_parts(int64_t n, int64_t batch) { return pick_split(n, batch); }

template <typename T>
static int gram_typed(int64_t R, int64_t n, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM, const void* V1,
                      int64_t ldv1, int64_t strideV1, void* G, int64_t nsplit, hipStream_t stream) {
  SweepArgs<T> p{};
  p.R = (int)R; p.n = n; p.M = (const T*)Mx; p.ldm = ldm; p.strideM = strideM;
  p.V1 = (const T*)V1; p.ldv1 = ldv1; p.strideV1 = strideV1;
  p.G = (T*)G; p.nsplit = (int)nsplit;
  ProfScope prof(V1 ? TTR_PROF_ROTGRAM : TTR_PROF_GEMM, stream);
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
    SweepArgs<T> q = p;
    q.M = p.M + b0 * strideM;
    if (V1) q.V1 = p.V1 + b0 * strideV1;
    q.G = p.G + b0 * nsplit * R * R;
    const dim3 grid((unsigned)nsplit, (unsigned)nb);
    if (V1) hipLaunchKernelGGL((rotgram_kernel<T, false>), grid, dim3(kThreads), 0, stream, q);
    else hipLaunchKernelGGL((rotgram_kernel<T, true>), grid, dim3(kThreads), 0, stream, q);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int sweep_gram_dispatch(int dtype, int64_t R, int64_t n, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM,
                        const void* V1, int64_t ldv1, int64_t strideV1, void* G, int64_t nsplit, hipStream_t stream) {
  TTR_REQUIRE(R >= 1 && R <= 64, TTR_E_UNSUPPORTED, "ttr_rowgram / ttr_rotgram: %lld rows (the fused kernels hold <= 64)",
              (long long)R);
  TTR_REQUIRE(nsplit >= 1 && nsplit <= 65535, TTR_E_INVALID, "ttr_rowgram / ttr_rotgram: bad split %lld", (long long)nsplit);
  if (dtype == TTR_F32) return gram_typed<float>(R, n, batch, Mx, ldm, strideM, V1, ldv1, strideV1, G, nsplit, stream);
  return gram_typed<double>(R, n, batch, Mx, ldm, strideM, V1, ldv1, strideV1, G, nsplit, stream);
}

template <typename T>
static int project_typed(int64_t R, int64_t n, int64_t ro, int64_t batch, const void* Mx, int64_t ldm, int64_t strideM,
                         const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2, int64_t strideV2,
                         const void* sigma, int64_t stride_sigma, int scale_right, void* right, int64_t ldr,
                         int64_t strideR, void* left, int64_t ldl, int64_t strideL, hipStream_t stream) {
  SweepArgs<T> p{};
  p.R = (int)R; p.n = n; p.M = (const T*)Mx; p.ldm = ldm; p.strideM = strideM;
  p.V1 = (const T*)V1; p.ldv1 = ldv1; p.strideV1 = strideV1;
  p.V2 = (const T*)V2; p.ldv2 = ldv2; p.strideV2 = strideV2;
  p.sigma = (const T*)sigma; p.stride_sigma = stride_sigma; p.ro = (int)ro; p.scale_right = scale_right;
  p.right = (T*)right; p.ldr = ldr; p.strideR = strideR;
  p.left = (T*)left; p.ldl = ldl; p.strideL = strideL;
  p.nsplit = pick_split(n, batch);
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
    hipLaunchKernelGGL(project_kernel<T>, dim3((unsigned)p.nsplit, (unsigned)nb), dim3(kThreads), 0, stream, q);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int sweep_project_dispatch(int dtype, int64_t R, int64_t n, int64_t ro, int64_t batch, const void* Mx, int64_t ldm,
                           int64_t strideM, const void* V1, int64_t ldv1, int64_t strideV1, const void* V2, int64_t ldv2,
                           int64_t strideV2, const void* sigma, int64_t stride_sigma, int scale_right, void* right,
                           int64_t ldr, int64_t strideR, void* left, int64_t ldl, int64_t strideL, hipStream_t stream) {
  TTR_REQUIRE(R >= 1 && R <= 64 && ro >= 1 && ro <= R, TTR_E_UNSUPPORTED,
              "ttr_project: %lld rows / %lld kept (the fused kernel holds <= 64 rows)", (long long)R, (long long)ro);
  if (dtype == TTR_F32)
    return project_typed<float>(R, n, ro, batch, Mx, ldm, strideM, V1, ldv1, strideV1, V2, ldv2, strideV2, sigma,
                                stride_sigma, scale_right, right, ldr, strideR, left, ldl, strideL, stream);
  return project_typed<double>(R, n, ro, batch, Mx, ldm, strideM, V1, ldv1, strideV1, V2, ldv2, strideV2, sigma,
                               stride_sigma, scale_right, right, ldr, strideR, left, ldl, strideL, stream);
}

}  // namespace ttr
