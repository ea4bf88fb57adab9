// Batched tall-skinny Householder QR for the left-to-right TT sweep (gfx950).
//
//   A[b] (m x n) = Q[b] (m x k) R[b] (k x n),  k = min(m, n),  n <= 64
//
// The left unfolding of a TT core is (R*I) x R -- thousands of rows, a few dozen columns -- and, in
// the main use of rounding (a+b, a*b), exactly rank deficient.  Gram-based orthogonalisation
// (CholeskyQR) breaks down on such inputs, so this is a true Householder QR (LAPACK geqrf/orgqr
// conventions), organised as a communication-avoiding TSQR whose blocks are factored with the
// blocked compact-WY algorithm on the matrix cores:
//
//   block    64*NW rows x (16*NT) columns, one workgroup of NW wave64 (NW = 8 above 256 rows, else 4).  The
//            block lives in REGISTERS as MFMA accumulator tiles: wave w owns rows 64w..64w+63 = 4 row
//            tiles x NT column tiles of 16x16 (v_mfma_*_16x16x4 C/D layout), 64 VGPRs for fp32.
//   panel    16 columns at a time, factored in a COLUMN-OWNING layout: the panel is transposed through
//            LDS so that wave w holds 16/NW whole columns (lane = row); the owner factors its columns
//            locally (norm and dot products are wave-local DPP reductions, one round per step),
//            publishes the reflectors, one LDS-only barrier, the waves to its right apply them.
//   block    after a panel: S = V^T V (MFMA), the 16x16 triangular factor T (larft recurrence), then
//   update   the trailing tiles get  A2 <- (I - V T^T V^T) A2  as three MFMA GEMMs
//            W = V^T A2 (the accumulator registers ARE the B operand: K-step s of a row tile is
//            register s of every lane group), W2 = -T^T W, A2 += V W2 -- no LDS traffic for A2.
//   tree     the R factors (n rows per block) are stacked and factored again by the same kernel
//            until one block is left; its R is the result.
//   apply    Q (or Q [C;0], the fused "push-left" of round_tt) is formed top-down, one block per
//            workgroup, entirely on MFMA: per panel (last to first) W = V^T C, W2 = -T W,
//            C += V W2.  No sequential Householder steps at all.
//
// Reflectors are stored explicitly (unit diagonal, zeros above) and transposed, Vt[j][row], so a
// Householder step emits one coalesced 1 KiB store; T factors (16x16 per panel) sit next to them.
#include <type_traits>

#include "ttr_common.h"

namespace ttr {

// rows per block = threads per block = 64 * NW (thread t also acts as "row t"); NW = 4 or 8 waves, chosen per
// tree level by the host: the sequential Householder steps of a block cost about the same for 256 and for 512
// rows (they are latency / issue bound, lane = row), so the larger block halves the number of blocks AND of
// tree nodes -- the smaller one is kept for matrices of <= 256 rows.
constexpr int BR4 = 256;
constexpr int PW = 16;    // panel width = MFMA tile edge
constexpr int VLD = 17;   // leading dimension of 16-column LDS panels (conflict-free column walks)
template <int I>
using IC = std::integral_constant<int, I>;

template <typename T>
struct QrLevel {
  const T* X;          // input matrix of this level: element (row, col) at X[row * ldx + col * xcs]
  int64_t ldx, strideX, xcs;  // xcs = 1 except for a TRANSPOSED level-0 input (ttr_qr_t: ldx = 1, xcs = leading dimension)
  int64_t m;           // rows
  int n;               // cols (<= 16*NT)
  int nb;              // row blocks (evenly split)
  T* Vt;               // [batch][nb][16*NT][BR] explicit reflectors, transposed
  T* tau;              // [batch][nb][16*NT]
  T* Tg;               // [batch][nb][NT][16][16] compact-WY triangular factors
  T* Rout;             // the block's R: next level's X (row block b*n) or the user's R
  int64_t ldr, strideR;
  int top;             // 1: single block, writes k x n to the user's R
  int rank_skip_c;     // rank-revealing early exit of PAIR blocks: a panel below (c eps)^2 of the block's squared norm is H = I; 0 = off
  // PUSHED level 0, row packing (round 4): when rows 32 .. 63 of Rm are below rank_skip_c eps of ||Rm||_F (the R factor of a
  // rank-inflated train: numerical rank <= 32 of 64) the pushed rows (kk >= 32, i) are dropped as zeros and every wave holds TWO
  // mode indices -- rows 0..31 = (kk, i = NW b + w), rows 32..63 = (kk, i = NW partner + w), partner = b +- nb / 2: one half of the blocks
  // (which half alternates with the item: XCD balance) absorbs the other, whose blocks write a zero R and exit.  Decided per item by every block from Rm itself; block 0 records it for the
  // apply kernel in pack_flag[item] (always written when non-null).  pack_ok = 0: never pack.
  int32_t* pack_flag;
  int pack_ok;
  // (round 6) pack_pre != 0: pack_flag[] was filled BEFORE this launch (pack_flags_kernel, same criterion): the blocks read the
  // decision instead of deriving it from Rm -- an absorbed block returns at once (it used to stage Rm, request its first K group
  // and write a zero R first: 9 k cycles, 16384 of them at the tail of every level-0 launch of the metric), its partner writes
  // the zero R block and the zero taus for it.
  int pack_pre;
  int grid_swap;
  // PUSHED level 0: the factored matrix is the left unfolding of  P[kk,i,c] = sum_r0 Rm[kk,r0] C[r0,i,c]
  // (tensor.py:1826-1832 fused into the next QR): block b owns rows {(kk, i): i = 4b + wave}
  const T* Rm;         // [pk x pRin], leading dimension ldrm
  int64_t ldrm, strideRm;
  const T* Cn;         // next core [pRin][pI][n] contiguous
  int64_t strideCn;
  int pk, pRin, pI;
  // block-diagonal next core (the middle core of a TT sum a + b, tensor.py:445-668, never materialised): rows < sumRa /
  // columns < sumCa come from Cn = a's core [sumRa][pI][sumCa], the rest from Cn2 = b's core [pRin - sumRa][pI][n - sumCa]
  const T* Cn2;
  int64_t strideCn2;
  int sumRa, sumCa;
  long long* dbg;      // optional: cycle stamps of block (dbg_bx, dbg_by) at phase boundaries (diagnostics)
  int dbg_bx, dbg_by;
  // TOP level, fp32 (optional): the user's R is left at the exponent the block was factored at -- R 2^-e with e the exponent of
  // the top block's largest entry, always taken -- and e is ADDED to expo_acc[item]: the per-core power-of-two normalisation of a
  // rounding sweep (ttr_pow2_normalize after every factorisation: 8 launches per 64^8 train) without a launch of its own.
  int32_t* expo_acc;
  // PUSHED level 0 (round 6, TTR_KNOB_QR_STAGGER, kilo-cycles; 0 = off): workgroups 256 .. 511 of the launch -- the second resident
  // block of every CU under round-robin dispatch -- start that much later, so that one block's HBM phase (the push) runs under
  // the other's panel chain instead of beside its push.
  int stagger_kc;
};

// beta = -sign(alpha) sqrt(alpha^2 + ss), tau = (beta - alpha)/beta, scale = 1/(alpha - beta)  (LAPACK larfg).
// fp32 uses the 1-ulp hardware sqrt / reciprocal: a Householder step sits on a serial latency chain and the
// IEEE division / sqrt expansions are ~40 dependent instructions; H = I - tau v v^T stays orthogonal to O(eps).
__device__ __forceinline__ void larfg_scalars(float alpha, float ss, float& beta, float& tj, float& scale) {
  beta = -copysignf(__builtin_amdgcn_sqrtf(alpha * alpha + ss), alpha);
  tj = (beta - alpha) * __builtin_amdgcn_rcpf(beta);
  scale = __builtin_amdgcn_rcpf(alpha - beta);
}
__device__ __forceinline__ void larfg_scalars(double alpha, double ss, double& beta, double& tj, double& scale) {
  beta = -copysign(sqrt(alpha * alpha + ss), alpha);
  tj = (beta - alpha) / beta;
  scale = 1.0 / (alpha - beta);
}

__device__ __forceinline__ void block_rows(int64_t m, int nb, int b, int64_t& row0, int& rows) {
  const int64_t q = m / nb, rem = m % nb;
  row0 = (int64_t)b * q + (b < rem ? b : rem);
  rows = (int)(q + (b < rem ? 1 : 0));
}

// ---------------------------------------------------------------- factor
// PAIR (NW = 8 only: two panel columns per wave): the owner factors its two columns with two 2-value reductions and
// publishes the pair (v0, v1, tau0, tau1, v0^T v1); the waves to its right apply BOTH reflectors with ONE 4-value
// reduction,  c <- c - a0 v0 - a1 v1,  a0 = tau0 v0^T c,  a1 = tau1 (v1^T c - (v0^T v1) a0);  the reflectors go to the
// workspace after the panel (all threads, coalesced) instead of from the owner's sequential chain.

// (item, block) of a workgroup.  grid_swap 0: grid (nb, batch).  1: grid (batch, nb), block-major -- all items' block 0, then all
// items' block 1, ...: the working blocks of packed items (b < nb / 2) are dispatched before the absorbed ones.  2 (round 5): the
// same two halves, but INSIDE a half the blocks of an item follow each other (item 0's blocks 0 .. nb/2-1, item 1's, ...).  Why:
// what a level-0 block of the fused push reads (core[:, i, :], i = 8 b .. 8 b + 7: 256-byte pieces 16 KB apart) and what the
// apply writes (rows kk I + i: 128-byte pieces 8 KB apart) has its address bits 11 .. 13 fixed by b and every item is aligned
// alike -- with mode 1 all resident workgroups share b and camp on an eighth of the HBM channels (the same effect measured on
// ttr_orth_fixup, profiles/r05_orth_stamps.txt).
__device__ __forceinline__ void block_of(int grid_swap, int nb, int& b, int64_t& bt) {
  if (grid_swap == 2) {
    const int half = nb >> 1;
    const int64_t id = (int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.y;   // launch order (x fastest); gridDim.x = batch
    const int64_t nwork = (int64_t)gridDim.x * half;
    const int64_t r = id < nwork ? id : id - nwork;
    bt = r / half;
    b = (int)(r - bt * half) + (id < nwork ? 0 : half);
  } else if (grid_swap) {
    b = blockIdx.y; bt = blockIdx.x;
  } else {
    b = blockIdx.x; bt = blockIdx.y;
  }
}

template <typename T, int NT, bool PUSHED, int NW, bool PAIR>
__global__ __launch_bounds__(64 * NW, (sizeof(T) == 4 ? 4 : 1)) void qr_factor_kernel(QrLevel<T> p) {
  using M = Mfma<T>;
  using Acc = typename M::Acc;
  typedef T V2 __attribute__((ext_vector_type(2)));  // fp32: v_pk_mul / v_pk_fma operands (two rows per instruction)
  typedef T V4 __attribute__((ext_vector_type(4)));
  constexpr int NP = PW * NT;  // padded column count
  constexpr int BR = 64 * NW;  // rows = threads of this block
  constexpr int NTH = 64 * NW;
  constexpr int CPW = PW / NW; // panel columns owned by a wave in the column-owning layout (4 or 2)
  __shared__ __attribute__((aligned(16))) T Vs[BR * VLD];       // current panel's reflectors [row][j]
  __shared__ T taus[NP];
  __shared__ T Ts[PW * VLD], Ss[PW * VLD];
  constexpr int WPC = NP > 2 * PW ? NP - PW : PW;               // W only exists for the trailing column tiles (tn >= 1)
  // per-wave partial W (also S partials) Wp[NW][PW][WPC]; during the Householder phases of a PAIR panel the same storage is
  // the reflector exchange buffer Xp[2][64][XLD] (double-buffered by phase parity): lane l's 16 values (v0[0..7], v1[0..7])
  // as four 16-byte accesses at a 20-word lane stride -- conflict-free for ds_write_b128 and ds_read_b128
  constexpr int XLD = 20;
  constexpr int WP_ELEMS = NW * PW * WPC, XP_ELEMS = PAIR ? 2 * 64 * XLD : 0;
  __shared__ __attribute__((aligned(32))) T WXs[WP_ELEMS > XP_ELEMS ? WP_ELEMS : XP_ELEMS];
  T (*Wp)[PW][WPC] = reinterpret_cast<T (*)[PW][WPC]>(WXs);
  T* const Xp = WXs;
  __shared__ T Xs[PW * VLD];                                    // scratch of the recursive T construction
  __shared__ T W2s[PW][NP + 1];
  __shared__ T pairt[NW];                                       // PAIR: v0^T v1 of every owner's pair

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int cl = lane & 15, g = lane >> 4;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid) >> 6;  // provably wave-uniform
  // grid (nb, batch), or -- grid_swap, experiment -- (batch, nb): block-major launch order
  int b;
  int64_t bt;
  block_of(p.grid_swap, p.nb, b, bt);
  int64_t row0;
  int rows;
  block_rows(p.m, p.nb, b, row0, rows);
  const int n = p.n;
  const int kb = rows < n ? rows : n;
  auto rowl = [&](int tm, int reg) { return wave * 64 + tm * 16 + M::row(lane, reg); };

  int dbgi = 0;
  auto stamp = [&]() { if (p.dbg && b == p.dbg_bx && bt == p.dbg_by && tid == 0) p.dbg[dbgi++] = (long long)clock64(); };
  if constexpr (PUSHED && PAIR) {
    if (p.stagger_kc > 0) {
      const int64_t lid = (int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.y;
      if (lid >= 256 && lid < 512) {
        const long long t0 = clock64();
        while (clock64() - t0 < (long long)p.stagger_kc * 1024) __builtin_amdgcn_s_sleep(16);
      }
    }
  }
  stamp();
  Acc acc[4][NT];
  if constexpr (PUSHED) {
    // acc <- Rm (pk x pRin) * C[:, i, :] (pRin x n) for this wave's mode index i = NW*b + wave: the wave's 64
    // rows are exactly (kk = 0..63, i).  Rm^T is staged once in LDS (aliasing Vs, conflict-free A-operand
    // reads), the core slice is read from global directly in MFMA B-operand layout.
    // Rs[kk][r0], leading dimension 66: the coalesced global read is stored without bank conflicts (consecutive
    // r0) and the A-operand reads below (16 kk x 2 r0 per 32-lane group: words 66 cl + g) touch 32 distinct banks.
    constexpr int RLD = 66;
    T* Rs = Vs;  // 64 x 66 <= 256 x 17
    int pre_flag = 0;   // (a VECTOR register until Rm is staged: the scalar copy would make the block wait for the load right here)
    if (p.pack_pre) {
      const int hb = p.nb >> 1;
      const bool absorbed_early = p.pack_ok == 3 ? (b >= hb) : p.pack_ok == 2 ? ((b & 1) != 0) != ((bt & 1) != 0) : (b >= hb) != ((bt & 1) != 0);
      // (only a block that MAY be absorbed waits for the flag here; a working block's flag travels with its Rm loads and is looked
      // at when Rm is staged -- as one test the flag's round trip stood at the head of every working block)
      if (absorbed_early) {
        if (p.pack_flag[bt] != 0) { stamp(); return; }   // absorbed by its partner block, which also writes this block's R and taus
      } else {
        pre_flag = p.pack_flag[bt + (lane & 0)];   // block-uniform value, per-lane address: stays in a VGPR
      }
    }
    const T* __restrict__ Rm = p.Rm + bt * p.strideRm;
    // Rm is the R factor of the previous core's QR in the rounding sweep, i.e. upper triangular: row tile tm of the product
    // then only needs the K steps r0 >= 16 tm (10 of the 16 tile x K-group combinations).  Detected here, not assumed: the
    // entry point takes any matrix.
    bool lower_nz = false;
    T sq_all = T(0), sq_low = T(0);
    {
      // all loads of the thread first (unconditional, from clamped offsets), then the stores: as one loop every iteration
      // waited for its own global round trip -- eight in a row, ~20 k cycles at the head of every block (ISA, round 4)
      constexpr int NE = 64 * 64 / NTH;
      T rvv[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * NTH, kk = idx >> 6, r0 = idx & 63;
        const bool ok = kk < p.pk && r0 < p.pRin;
        rvv[e] = Rm[ok ? (int64_t)kk * p.ldrm + r0 : 0];
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * NTH, kk = idx >> 6, r0 = idx & 63;
        const T rv = (kk < p.pk && r0 < p.pRin) ? rvv[e] : T(0);
        lower_nz = lower_nz || (kk > r0 && rv != T(0));
        sq_all += rv * rv;
        if (kk >= 32) sq_low += rv * rv;
        Rs[kk * RLD + r0] = rv;
      }
    }
    wave_sum2(sq_all, sq_low);
    if (lane == 0) {
      pairt[wave] = __ballot(lower_nz) != 0ull ? T(1) : T(0);  // (pairt is free until the first panel)
      Ss[wave] = sq_all; Ss[NW + wave] = sq_low;               // (so is Ss)
    }
    const int imode = b * NW + wave;
    const bool ivalid = imode < p.pI;
    const int ksteps = (p.pRin + 3) >> 2;
    bool packed = false;
    // which half of an item's blocks does the work when it packs alternates with the item: consecutive workgroups go to
    // consecutive XCDs, i.e. XCD = b for nb = 8 -- with a fixed half, four of the eight XCDs got all the heavy blocks (measured:
    // 11.7 instead of 10.6 ms/step for the kind, slower than not packing at all)
    const int half_nb = p.nb >> 1;
    const bool absorbed_blk = p.pack_ok == 3 ? (b >= half_nb)
                            : p.pack_ok == 2 ? ((b & 1) != 0) != ((bt & 1) != 0) : (b >= half_nb) != ((bt & 1) != 0);
    const int partner_blk = p.pack_ok == 2 ? (b ^ 1) : (b < half_nb ? b + half_nb : b - half_nb);
    if (p.Cn2 == nullptr) {
      // The wave's core slice C[:, i, :] (pRin rows of n values, 16 KiB for a 64 x 64 x 64 core) is streamed in groups
      // of four K steps with the NEXT group's loads in flight under the current group's MFMAs: with one K step per
      // round trip (16 KiB in flight per CU) the whole phase ran at the memory latency -- 64 k of a block's 189 k cycles
      // under load (cycle stamps) -- two groups keep 64 KiB per block in flight.
      // Addressing: a wave-uniform row pointer (SGPR pair, advanced per K step) plus ONE per-lane 32-bit element offset
      // -- the loads are issued back to back without per-load address arithmetic; partial shapes (ragged mode / rank /
      // column counts) take the general path: unconditional loads from clamped offsets, masked afterwards (no branches).
      const int imode_u = b * NW + wave_id;  // = imode, provably wave-uniform
      const T* __restrict__ Cb = p.Cn + bt * p.strideCn;
      const unsigned cs = (unsigned)p.pI * (unsigned)n;  // r0 stride in elements (host checks pRin * pI * n < 2^31)
      const unsigned loff = (unsigned)g * cs + (unsigned)cl;
      const bool full = ivalid && n == NP && (p.pRin & 15) == 0;  // wave-uniform: every element of every group exists
      constexpr int KG = 4;
      const int ngroups = (ksteps + KG - 1) / KG;  // <= 4 (pRin <= 64)
      auto stream = [&](auto FULL) {
        constexpr bool kFull = decltype(FULL)::value != 0;
        T bvA[KG][NT], bvB[KG][NT];
        int im_cur = imode_u;           // the mode index the loads address (packed: a second pass with the absorbed block's)
        int tm_lo = 0, tm_hi = 4, tm_sub = 0;   // row tiles of this pass; A rows of tile tm = Rm rows 16 (tm - tm_sub) ..
        auto load_group = [&](int grp, T (&bv)[KG][NT]) {
          if constexpr (kFull) {
            const T* __restrict__ rowp = Cb + (size_t)im_cur * n + (size_t)(grp * KG * 4) * cs;
#pragma unroll
            for (int kk = 0; kk < KG; ++kk)
#pragma unroll
              for (int tn = 0; tn < NT; ++tn) bv[kk][tn] = rowp[(size_t)(kk * 4) * cs + loff + tn * PW];
          } else {
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
              const int r0 = (grp * KG + kk) * 4 + g;
#pragma unroll
              for (int tn = 0; tn < NT; ++tn) {
                const int col = tn * PW + cl;
                const bool ok = ivalid && r0 < p.pRin && col < n;
                const unsigned o = ok ? (unsigned)imode_u * (unsigned)n + (unsigned)r0 * cs + (unsigned)col : 0u;
                const T v = Cb[o];
                bv[kk][tn] = ok ? v : T(0);
              }
            }
          }
        };
        bool upper = false;
        auto mma_group = [&](int grp, const T (&bv)[KG][NT]) {
#pragma unroll
          for (int kk = 0; kk < KG; ++kk) {
            const int r0 = (grp * KG + kk) * 4 + g;  // rows >= pRin: the staged Rs column is zero (and bv is zero)
            T av[4];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) av[tm] = Rs[(tm * 16 + cl) * RLD + r0];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
              if (tm < tm_lo || tm >= tm_hi) continue;  // wave-uniform (packed: two row tiles per pass)
              // upper-triangular Rm: rows 16 tm' .. 16 tm' + 15 are zero in the columns of K groups < tm' (a K group = 16 columns)
              if (upper && grp < tm - tm_sub) continue;  // wave-uniform
#pragma unroll
              for (int tn = 0; tn < NT; ++tn) acc[tm][tn] = M::mma(tm_sub == 0 ? av[tm] : av[tm & 1], bv[kk][tn], acc[tm][tn]);
            }
          }
        };
        // group 0 of the pass is already in bvA.  `im_next` >= 0: the first group of the NEXT pass (the absorbed block's mode
        // index) is requested as soon as bvA is free, i.e. under the last group's MFMAs (an even number of groups; otherwise
        // the caller loads it)
        auto run_pass = [&](int im_next) -> bool {
          bool prefetched = false;
          for (int grp = 0; grp < ngroups; grp += 2) {
            if (grp + 1 < ngroups) load_group(grp + 1, bvB);
            mma_group(grp, bvA);
            if (grp + 1 < ngroups) {
              if (grp + 2 < ngroups) load_group(grp + 2, bvA);
              else if (im_next >= 0) { im_cur = im_next; load_group(0, bvA); prefetched = true; }
              mma_group(grp + 1, bvB);
            }
          }
          return prefetched;
        };
        load_group(0, bvA);  // in flight while Rs is being staged (the same mode index whether the block packs or not)
        lds_barrier();
        {
          T any = pairt[0], sa = Ss[0], sl = Ss[NW];
#pragma unroll
          for (int w = 1; w < NW; ++w) { any += pairt[w]; sa += Ss[w]; sl += Ss[NW + w]; }
          upper = any == T(0);
          if constexpr (kFull && NW == 8 && PAIR) {
            const T ce = T(p.rank_skip_c) * Num<T>::eps();
            if (p.pack_pre) packed = __builtin_amdgcn_readfirstlane(pre_flag) != 0;   // (the host only sets pack_pre when the static conditions below hold)
            else
              packed = p.pack_ok && p.rank_skip_c > 0 && p.pk == 64 && (p.nb & 1) == 0 && p.pI == NW * p.nb &&
                       lane_get(sl, 0) <= ce * ce * lane_get(sa, 0);
          }
        }
        if (packed && absorbed_blk) return;   // absorbed by its partner block (block-uniform; handled by the caller)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) acc[tm][tn] = M::zero();
        if (packed) tm_hi = 2;
        const bool pre = run_pass(packed ? NW * partner_blk + wave_id : -1);
        if (packed) {   // second pass: the absorbed block's mode index into row tiles 2, 3 (Rm rows 0 .. 31 again)
          im_cur = NW * partner_blk + wave_id;
          tm_lo = 2; tm_hi = 4; tm_sub = 2;
          if (!pre) load_group(0, bvA);
          (void)run_pass(-1);
        }
      };
#ifdef TTR_QR_PUSH_KSTEP
      // Round 6, measured WITHOUT gain and therefore not compiled by default (tools/build_variant.sh kstep ttr_qr.hip
      // -DTTR_QR_PUSH_KSTEP; profiles/r06_push_variants.txt): the K-STEP pipeline (whole shapes with a multiple of 8 K steps; fp32
      // 8-wave PAIR blocks -- the metric's).  Hypothesis: since row packing a pass performs half the MFMAs per byte it loads, and
      // the group pipeline above -- 16 loads, then a burst of 32 MFMAs during which ONE group is in flight -- would run at one
      // memory round trip per group.  Here the 32 operand registers are EIGHT K-step slots: step t is consumed from slot t mod 8 and
      // the slot is refilled at once with step t + 8 (of the next pass when this one ends), so two groups' worth of loads stay in
      // flight through the MFMAs; same MFMAs in the same order: bit-identical (sha256 of R); a block that may turn out absorbed
      // requests nothing before it knows (the group pipeline fetches and drops 32 KB).  Result: push 41 - 45 k cycles against
      // 38 - 45 k, the step 24.88 against 24.89 ms.  The depth of the prefetch is not what paces the phase (push_membench: 1, 2 or 4
      // groups in flight move the same bytes in the same 12.8 k ticks); without ANY MFMA (-DTTR_QR_PUSH_NOMMA) the phase still takes
      // 24 k cycles alone on a CU and 37 k under load: ~9 k of fixed work (R staged through LDS, the range guard and the block norm:
      // five block-wide barriers) + 256 KB at the ~17 B/clk one CU's load path sustains on this pattern.
      auto stream_ks = [&]() {
        constexpr int NS = 8;
        T bq[NS][NT];
        const T* __restrict__ ldp = Cb + (size_t)imode_u * n + loff;   // the loader's position: lane pointer, advanced per K step
        const size_t kstride = (size_t)4 * cs;
        auto load_slot = [&](T (&bs)[NT]) {
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) bs[tn] = ldp[tn * PW];
          ldp += kstride;
        };
        const bool may_absorb = p.pack_ok && p.rank_skip_c > 0 && p.pk == 64 && (p.nb & 1) == 0 && p.pI == NW * p.nb && absorbed_blk;
        bool upper = false;
        auto decide = [&]() {   // after the barrier that publishes Rs and the waves' partial sums
          T any = pairt[0], sa = Ss[0], sl = Ss[NW];
#pragma unroll
          for (int w = 1; w < NW; ++w) { any += pairt[w]; sa += Ss[w]; sl += Ss[NW + w]; }
          upper = any == T(0);
          const T ce = T(p.rank_skip_c) * Num<T>::eps();
          packed = p.pack_ok && p.rank_skip_c > 0 && p.pk == 64 && (p.nb & 1) == 0 && p.pI == NW * p.nb &&
                   lane_get(sl, 0) <= ce * ce * lane_get(sa, 0);
        };
        // A block that may be absorbed decides first and requests its slots afterwards; every other block requests them first (in
        // flight while Rs is staged).  ONE copy of the 32 requests, in slot order (scheduling barriers): the loop's static vmcnt
        // counts must hold on the entry path too -- with two copies hipcc ordered one of them differently and every first slot of an
        // iteration waited for all but 4 loads.
        if (may_absorb) {
          lds_barrier();
          decide();
          if (packed) return;
        }
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) { load_slot(bq[sl]); __builtin_amdgcn_sched_barrier(0); }
        if (!may_absorb) {
          lds_barrier();
          decide();
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) acc[tm][tn] = M::zero();
        const int total = packed ? 2 * ksteps : ksteps;   // K steps over both passes (multiples of NS)
        int t_ld = NS;                                    // next step to request
        // NS steps: consume slot sl, refill it with step t + NS.  REFILL is a compile-time property of the iteration -- every
        // iteration but the last refills all its slots -- because hipcc's s_waitcnt insertion only keeps the loads counted
        // (vmcnt(28) before a slot's first MFMA: seven slots stay in flight) when every path through the loop body issues the same
        // loads in the same order; with the refill under a branch it waited for vmcnt(0) at every slot.
        auto steps8 = [&](int t0, auto REFILL) {
          constexpr bool kRefill = decltype(REFILL)::value != 0;
          const bool second = t0 >= ksteps;               // packed: the absorbed block's mode index into row tiles 2, 3
          const int tm_lo = second ? 2 : 0, tm_hi = packed ? tm_lo + 2 : 4, tm_sub = tm_lo;
          const int ks0 = second ? t0 - ksteps : t0;
#pragma unroll
          for (int sl = 0; sl < NS; ++sl) {
            const int ks = ks0 + sl, grp = ks >> 2;
            const int r0 = ks * 4 + g;
            T av[4];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) av[tm] = Rs[(tm * 16 + cl) * RLD + r0];
#if defined(TTR_QR_PUSH_NOMMA)
            // (diagnostics build: the push without its MFMAs -- what do the loads and the rest of the phase cost?  wrong results)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) asm volatile("" ::"v"(bq[sl][tn]), "v"(av[tn]));
#elif defined(TTR_QR_PUSH_ZSKIP)
            // (-DTTR_QR_PUSH_ZSKIP, measured SLOWER: push 53 - 57 k instead of 42 - 45 k cycles -- the compare + ballot + branch per tile
            // cost the waves more issue time than the skipped MFMAs free on the pipe.)  Zero operand tiles are not multiplied: the cores of a SUM of trains (tensor.py:445-668: blockdiag(a, b), what rounding
            // is mostly called on) are half zeros, and a 4 x 16 B tile that is zero in every lane adds nothing to its accumulators
            // (x + a * 0 = x: bit-identical for finite R).  One compare + ballot per tile, against two MFMAs (64 cycles of the
            // SIMD's matrix pipe, which BOTH resident blocks' pushes share -- the pipe, not HBM, paces the phase).
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
              if (__ballot(bq[sl][tn] != T(0)) == 0ull) continue;   // wave-uniform
#pragma unroll
              for (int tm = 0; tm < 4; ++tm) {
                if (tm < tm_lo || tm >= tm_hi) continue;       // wave-uniform
                if (upper && grp < tm - tm_sub) continue;      // wave-uniform (upper-triangular Rm: zero K groups)
                acc[tm][tn] = M::mma(tm_sub == 0 ? av[tm] : av[tm & 1], bq[sl][tn], acc[tm][tn]);
              }
            }
#else
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
              if (tm < tm_lo || tm >= tm_hi) continue;       // wave-uniform
              if (upper && grp < tm - tm_sub) continue;      // wave-uniform (upper-triangular Rm: zero K groups)
#pragma unroll
              for (int tn = 0; tn < NT; ++tn) acc[tm][tn] = M::mma(tm_sub == 0 ? av[tm] : av[tm & 1], bq[sl][tn], acc[tm][tn]);
            }
#endif
            if constexpr (kRefill) {
              // (the second pass begins: a select on the pointer, not a branch around the loads)
              const T* __restrict__ p2 = Cb + (size_t)(NW * partner_blk + wave_id) * n + loff;
              ldp = (t_ld == ksteps) ? p2 : ldp;
              load_slot(bq[sl]);
              ++t_ld;
            }
          }
        };
        int t0 = 0;
        for (; t0 + NS < total; t0 += NS) steps8(t0, IC<1>{});
        steps8(t0, IC<0>{});
      };
      if constexpr (NW == 8 && PAIR && sizeof(T) == 4) {
        if (full && (ksteps & 7) == 0) stream_ks();
        else if (full) stream(IC<1>{});
        else stream(IC<0>{});
      } else {
        if (full) stream(IC<1>{});
        else stream(IC<0>{});
      }
#else
      if (full) stream(IC<1>{});
      else stream(IC<0>{});
#endif
      if (packed && absorbed_blk) {
        // an absorbed block: its rows live in its partner block.  The level above reads this block's R: zeros; its reflectors are
        // H = I (zero taus: the apply kernel returns at once for it)
        T* __restrict__ Ro = p.Rout + bt * p.strideR + (int64_t)b * n * p.ldr;
        for (int idx = tid; idx < n * n; idx += NTH) Ro[(int64_t)(idx / n) * p.ldr + idx % n] = T(0);
        if (tid < NP) p.tau[(bt * p.nb + b) * (int64_t)NP + tid] = T(0);
        if (p.pack_flag && b == 0 && tid == 0) p.pack_flag[bt] = p.pack_ok;   // (block 0 records the decision whichever half it is in)
        stamp();
        return;
      }
    } else {
      lds_barrier();
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) acc[tm][tn] = M::zero();
      // block-diagonal core blockdiag(a, b): column tiles that lie inside one diagonal block only walk that block's
      // rows (for equal ranks: half the K steps), tiles that straddle the boundary see the zero blocks as zeros
      const int ra = p.sumRa, ca = p.sumCa, cb = n - ca;
      const T* __restrict__ Ca = p.Cn + bt * p.strideCn + (int64_t)imode * ca;
      const T* __restrict__ Cb = p.Cn2 + bt * p.strideCn2 + (int64_t)imode * cb;
      const int64_t sa = (int64_t)p.pI * ca, sb = (int64_t)p.pI * cb;
      // (round 4: the loads of a K step are issued together, from clamped addresses, and the NEXT step's loads before this
      // step's MFMAs -- as one load-use loop the phase was ksteps x NT serial global round trips, 64 for a 64 x 64 x 64 core)
      auto load_step = [&](int ks, T (&bv)[NT]) {
        const int r0 = ks * 4 + g;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
          const int col = tn * PW + cl;
          const bool in_a = col < ca;
          const bool ok = ivalid && r0 < p.pRin && col < n && (in_a ? r0 < ra : r0 >= ra);
          const T* __restrict__ src = in_a ? Ca + (ok ? (int64_t)r0 * sa + col : 0) : Cb + (ok ? (int64_t)(r0 - ra) * sb + (col - ca) : 0);
          const T v = *src;
          bv[tn] = ok ? v : T(0);
        }
      };
      T bcur[NT], bnxt[NT];
      if (ksteps > 0) load_step(0, bcur);
      for (int ks = 0; ks < ksteps; ++ks) {
        if (ks + 1 < ksteps) load_step(ks + 1, bnxt);
        const int r0 = ks * 4 + g;
        T av[4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) av[tm] = Rs[(tm * 16 + cl) * RLD + r0];
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
          const int c0 = tn * PW;
          const bool all_a = c0 + PW <= ca, all_b = c0 >= ca;  // wave-uniform
          if ((all_a && ks * 4 >= ra) || (all_b && ks * 4 + 3 < ra)) continue;  // this tile's rows of the step are all zero
#pragma unroll
          for (int tm = 0; tm < 4; ++tm) acc[tm][tn] = M::mma(av[tm], bcur[tn], acc[tm][tn]);
        }
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) bcur[tn] = bnxt[tn];
      }
    }
    if (p.pack_pre) {
      if (packed) {   // the absorbed block returned at once: its R block (zeros: the level above reads it) and its taus (H = I) are written here
        T* __restrict__ Rp = p.Rout + bt * p.strideR + (int64_t)partner_blk * n * p.ldr;
        for (int idx = tid; idx < n * n; idx += NTH) Rp[(int64_t)(idx / n) * p.ldr + idx % n] = T(0);
        if (tid < NP) p.tau[(bt * p.nb + partner_blk) * (int64_t)NP + tid] = T(0);
      }
    } else if (p.pack_flag && b == 0 && tid == 0) {
      p.pack_flag[bt] = packed ? p.pack_ok : 0;
    }
    lds_barrier();  // Rs aliases Vs
  } else {
    const T* __restrict__ X = p.X + bt * p.strideX + row0 * p.ldx;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rowl(tm, r), col = tn * PW + cl;
          acc[tm][tn][r] = (row < rows && col < n) ? X[(int64_t)row * p.ldx + (int64_t)col * p.xcs] : T(0);
        }
  }
  // fp32 range guard: the Householder steps square the entries (column norms, alpha^2 + ss); for blocks whose
  // entries are near 1e19 or near 1e-19 -- legitimate fp32 data, e.g. one core of a train scaled by 1e-15 -- those
  // squares overflow or fall into the denormal range (the fast sqrt / rcp then return 0 / inf: NaN).  The block is
  // therefore scaled by the exact power of two of its largest entry for the factorisation and R is scaled back
  // on output; reflectors and T factors are scale invariant, so nothing else changes (bit-identical in range).
  int bexp = 0;
  T rank_thr = T(0);
  if constexpr (sizeof(T) == 4 && PAIR) {
    // (round 6) the range guard's block maximum and the block's squared norm -- the threshold of the rank-revealing early exit,
    // below -- travel through ONE block-wide reduction instead of two (a barrier pair of eight waves costs 1 - 2 k cycles): every
    // wave sums its squares at the exponent of ITS largest entry (0 when that is O(1), like the block's) and the partial sums are
    // brought to the block's exponent afterwards -- exact powers of two: the same bits as scaling first and summing then, unless a
    // wave's entries lie 2^60 below the block's (where the old order lost them in the denormals)
    T mx = T(0);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmax(mx, fabs(acc[tm][tn][r]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    int ew = 0;
    if (mx > T(0) && mx < T(3e38)) {
      (void)frexpf((float)mx, &ew);
      if (ew > -8 && ew < 8) ew = 0;
    }
    ew = __builtin_amdgcn_readfirstlane(ew);
    T sq = T(0);
    if (ew == 0) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r) sq += acc[tm][tn][r] * acc[tm][tn][r];
    } else {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const T x = ldexpf((float)acc[tm][tn][r], -ew); sq += x * x; }
    }
    sq = wave_sum_dpp(sq);
    if (lane == 0) { Ss[wave] = mx; Ss[NW + wave] = sq; Ss[2 * NW + wave] = (T)ew; }
    lds_barrier();
    T bm = Ss[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) bm = fmax(bm, Ss[w]);
    if (bm > T(0) && bm < T(3e38)) {
      (void)frexpf((float)bm, &bexp);
      if (!(!PUSHED && p.top && p.expo_acc) && bexp > -8 && bexp < 8) bexp = 0;  // already O(1): leave the data alone
    }
    T bf2 = ldexpf((float)Ss[NW], 2 * ((int)Ss[2 * NW] - bexp));
#pragma unroll
    for (int w = 1; w < NW; ++w) bf2 += ldexpf((float)Ss[NW + w], 2 * ((int)Ss[2 * NW + w] - bexp));
    lds_barrier();  // Ss is reused by the panels
    if constexpr (!PUSHED) {
      if (p.top && p.expo_acc && tid == 0) p.expo_acc[bt] += bexp;   // (one block per item at the top; R keeps the exponent, see below)
    }
    if (bexp != 0) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[tm][tn][r] = ldexpf((float)acc[tm][tn][r], -bexp);
    }
    // Rank-revealing early exit (PAIR blocks, round 4): see below
    const T ce = T(p.rank_skip_c) * Num<T>::eps();
    rank_thr = lane_get(ce * ce * bf2, 0);   // (wave-uniform: lives in SGPRs)
    if (p.rank_skip_c <= 0) rank_thr = T(-1);
  } else {
    if constexpr (sizeof(T) == 4) {
      T mx = T(0);
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmax(mx, fabs(acc[tm][tn][r]));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
      if (lane == 0) Ss[wave] = mx;
      lds_barrier();
      T bm = Ss[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) bm = fmax(bm, Ss[w]);
      lds_barrier();  // Ss is reused by the panels
      if (bm > T(0) && bm < T(3e38)) {
        (void)frexpf((float)bm, &bexp);
        if (!(!PUSHED && p.top && p.expo_acc) && bexp > -8 && bexp < 8) bexp = 0;  // already O(1): leave the data alone
      }
      // (QrLevel::expo_acc; not in the PUSHED instances -- the metric's level-0 kernel sits at its register cap, and a pushed
      // factorisation that is its own top level is a small one: factor_run normalises its R with a launch of its own)
      if constexpr (!PUSHED) {
        if (p.top && p.expo_acc && tid == 0) p.expo_acc[bt] += bexp;   // (one block per item at the top; R keeps the exponent, see below)
      }
      if (bexp != 0) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tm][tn][r] = ldexpf((float)acc[tm][tn][r], -bexp);
      }
    }
    // Rank-revealing early exit (PAIR blocks, round 4): rounding works on rank-INFLATED trains (sums, products: the metric's
    // t = g + g has unfoldings of rank 32 in 64 columns), so after the first panels the remaining columns of a block are
    // rounding noise.  A panel whose remaining part (rows >= j0 of its 16 columns, after the earlier panels' updates) has a
    // squared Frobenius norm below (8 eps)^2 of the block's is not factored: its reflectors are H = I (tau = 0), what is dropped
    // below the diagonal is a backward error of 8 eps ||block||_F -- the size of the factorisation's own -- and the apply kernel
    // skips trailing identity panels altogether.  bf2: the block's squared Frobenius norm (at the factorisation's exponent).
    if constexpr (PAIR) {
      T sq = T(0);
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r) sq += acc[tm][tn][r] * acc[tm][tn][r];
      sq = wave_sum_dpp(sq);
      if (lane == 0) Ss[wave] = sq;
      lds_barrier();
      T bf2 = Ss[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) bf2 += Ss[w];
      lds_barrier();  // Ss is reused by the panels
      const T ce = T(p.rank_skip_c) * Num<T>::eps();
      rank_thr = lane_get(ce * ce * bf2, 0);   // (wave-uniform: lives in SGPRs)
      if (p.rank_skip_c <= 0) rank_thr = T(-1);
    }
  }
  stamp();
#ifdef TTR_QR_WSTAMPS
  // diagnostics build only: per-wave cycle stamps of the first panel's phases, dbg[64 + 40 * wave + 4 * phase + k]
  auto wstamp = [&](int pn, int k4) {
    if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && pn == 0) p.dbg[64 + 40 * wave_id + k4] = (long long)clock64();
  };
#define TTR_WSTAMP(pn, k4) wstamp(pn, k4)
#else
#define TTR_WSTAMP(pn, k4)
#endif
  const int64_t blk = bt * p.nb + b;
  T* __restrict__ Vt = p.Vt + blk * (int64_t)NP * BR;
  T* __restrict__ tau = p.tau + blk * (int64_t)NP;
  T* __restrict__ Tg = p.Tg + blk * (int64_t)NT * PW * PW;
  if (tid < NP) taus[tid] = T(0);

  // Panels are expanded at compile time as well (acc[tm][pnl] must be a static register index).
  auto panel = [&](auto PN) {
    constexpr int pnl = decltype(PN)::value;
    const int j0 = pnl * PW;
    int nsteps = kb - j0;
    nsteps = nsteps < 0 ? 0 : (nsteps > PW ? PW : nsteps);
    // Panel factorisation in a COLUMN-OWNING layout: the 256 x 16 panel is transposed through LDS (the
    // Vs buffer, whose column j is only ever touched by the owner of column j) so that wave w holds panel
    // columns 4w..4w+3 completely -- lane l has rows l, l+64, l+128, l+192.  The column norm and every
    // reflector dot product are then wave-local DPP reductions; the only cross-wave traffic of a
    // Householder step is the reflector itself: ONE barrier per step.  Rows <= jj only exist in q = 0.
    T pc[CPW][NW];  // [column CPW*w + cc][row lane + 64 q]
    // PAIR keeps its two columns as row PAIRS (pcv[cc][h] = rows lane + 128 h, lane + 128 h + 64): every element-wise
    // operation of a Householder step is then a packed two-row instruction -- the step chain is bound by the number of
    // instructions the one active wave has to issue (~8 cycles each, measured), not by their width
    V2 pcv[2][NW / 2];
    // PAIR: no barrier before this store -- the previous panel's update only reads the wave's OWN rows of Vs (A operand of
    // A2 += V W2), which are the rows it rewrites here; the single-step variant's owners write other waves' rows early
    if constexpr (!PAIR) { if (pnl > 0) lds_barrier(); }  // the previous panel's MFMA update may still be reading Vs
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) Vs[rowl(tm, r) * VLD + cl] = acc[tm][pnl][r];
    if constexpr (PAIR) {  // S and T of this panel are assembled during the phases (below): start from zero
      if (tid < PW * PW) { Ts[(tid >> 4) * VLD + (tid & 15)] = T(0); Ss[(tid >> 4) * VLD + (tid & 15)] = T(0); }
    }
    lds_barrier();
    bool rank_skip = false;
    if constexpr (PAIR) {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int q = 0; q < NW; ++q) pcv[cc][q >> 1][q & 1] = Vs[(lane + 64 * q) * VLD + wave * 2 + cc];
      // what is left of this panel: rows >= j0 of its columns (rows < 64 only exist in the first half of pair 0)
      {
        T e = T(0);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          V2 x0 = pcv[cc][0];
          if (lane < j0) x0.x = T(0);
          V2 a2 = x0 * x0;
#pragma unroll
          for (int h = 1; h < NW / 2; ++h) a2 = pcv[cc][h] * pcv[cc][h] + a2;
          e += a2.x + a2.y;
        }
        e = wave_sum_dpp(e);
        if (lane == 0) pairt[wave] = e;   // (pairt: free until the first owner publishes)
        lds_barrier();
        T pr2 = pairt[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) pr2 += pairt[w];
        rank_skip = lane_get(pr2, 0) <= rank_thr;   // block-uniform, scalar
        lds_barrier();                     // pairt is rewritten by the owners
        if (rank_skip) nsteps = 0;
      }
    } else {
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc)
#pragma unroll
        for (int q = 0; q < NW; ++q) pc[cc][q] = Vs[(lane + 64 * q) * VLD + wave * CPW + cc];
    }
    // (PAIR: the owners store their reflectors to Vs only after their phase barrier, which every wave reaches after this read)
    if constexpr (!PAIR) lds_barrier();  // all columns are in registers before reflectors start overwriting Vs
    stamp();
    if (nsteps < PW) {  // unused reflectors of this panel are H = I: v = 0, tau = 0
      for (int j = nsteps; j < PW; ++j) Vs[tid * VLD + j] = T(0);
    }
    // Steps: a RUNTIME loop over the owner wave; the owner factors its CPW columns locally (build a reflector,
    // apply it to its remaining columns, ... -- all wave-local, no barrier), publishes the reflectors, and after
    // ONE barrier the waves to its right apply them to their own columns.  A panel costs NW barrier phases
    // instead of 16 (the fully unrolled 16-step version also thrashed the instruction cache).
    // columns 2p, 2p + 1 of the panel's T (wave 0, lanes = rows i < 2p; T and S in LDS, zero where not yet / never set):
    //   T[i][c] = -tau_c sum_{l < c} T[i][l] S[l][c]   (larft, forward columnwise; T[i][l] = 0 for l < i)
    auto t_columns = [&](int pp) {
      const int c0 = 2 * pp, c1 = c0 + 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own earlier T columns
      if (lane < c0) {
        // fully unrolled over l = 0 .. 13 (entries at l >= c0 of the T row are still zero: no predicate needed) so that all
        // reads are in flight at once: as a counted loop every iteration waited for its own LDS round trip and wave 0
        // arrived last at the phase barriers
        T tl[PW - 2], s0[PW - 2], s1[PW - 2];
#pragma unroll
        for (int l = 0; l < PW - 2; ++l) {
          tl[l] = Ts[lane * VLD + l];
          s0[l] = Ss[l * VLD + c0];
          s1[l] = Ss[l * VLD + c1];
        }
        T a = T(0), bsum = T(0), a2 = T(0), b2 = T(0);
#pragma unroll
        for (int l = 0; l < PW - 2; l += 2) {
          a += tl[l] * s0[l]; bsum += tl[l] * s1[l];
          a2 += tl[l + 1] * s0[l + 1]; b2 += tl[l + 1] * s1[l + 1];
        }
        a += a2; bsum += b2;
        const T ti0 = -taus[j0 + c0] * a;
        bsum += ti0 * Ss[c0 * VLD + c1];
        Ts[lane * VLD + c0] = ti0;
        Ts[lane * VLD + c1] = -taus[j0 + c1] * bsum;
      }
    };
    if constexpr (PAIR) {
      // Every wave runs its OWN sequence of phases (same number of barriers for all): first the phases of the owners to
      // its left (barrier, apply their pair), then its own (factor, publish, barrier, deferred stores), then only the
      // barriers of the owners to its right.  One loop with all three roles under branches made the register allocator
      // keep the columns in two register sets and copy all 16 values back and forth in every phase.
      static_assert(!PAIR || (CPW == 2 && NW == 8), "PAIR: two columns per wave, eight waves");
      constexpr int NH = NW / 2;
      const int nph = (nsteps + 1) >> 1;            // phases = column pairs that take a step (block-uniform)
      const bool mine = wave_id < nph;
      const int napply = mine ? wave_id : nph;
      const bool live = j0 + wave_id * 2 < n;       // the wave's columns exist
      // Waves that still have their phase ahead are on the critical chain of the panel (owner k -> barrier -> owner k + 1
      // applies, factors -> ...): they issue at raised priority; a wave whose phase is over drops to priority 0 for the
      // S / T assembly below, so that it only takes issue slots the chain leaves free (its SIMD partner may be the
      // current owner; without this the chain slowed from ~1600 to ~1900 cycles per phase, cycle stamps).
      __builtin_amdgcn_s_setprio(3);
      for (int owv = 0; owv < napply; ++owv) {
        const int jj = j0 + owv * 2, jj1 = jj + 1;
        T* const xb = Xp + (owv & 1) * (64 * XLD) + lane * XLD;
        lds_barrier();  // the owner's pair (exchange slot), its taus and v0^T v1 are visible
        TTR_WSTAMP(pnl, 4 * owv + 2);
        if (live) {     // apply H1 H0 to the two columns
            const V4* const xr = reinterpret_cast<const V4*>(xb);
            const V4 r0 = xr[0], r1 = xr[1], r2 = xr[2], r3 = xr[3];
            const V2 w0[NH] = {V2{r0.x, r0.y}, V2{r0.z, r0.w}, V2{r1.x, r1.y}, V2{r1.z, r1.w}};
            const V2 w1[NH] = {V2{r2.x, r2.y}, V2{r2.z, r2.w}, V2{r3.x, r3.y}, V2{r3.z, r3.w}};
            const T ta = taus[jj], tb = taus[jj1], t12 = pairt[owv];
            V2 e0 = w0[0] * pcv[0][0], e1 = w0[0] * pcv[1][0], e2 = w1[0] * pcv[0][0], e3 = w1[0] * pcv[1][0];
#pragma unroll
            for (int h = 1; h < NH; ++h) {
              e0 = w0[h] * pcv[0][h] + e0; e1 = w0[h] * pcv[1][h] + e1;
              e2 = w1[h] * pcv[0][h] + e2; e3 = w1[h] * pcv[1][h] + e3;
            }
            T d4[4] = {e0.x + e0.y, e1.x + e1.y, e2.x + e2.y, e3.x + e3.y};
            wave_sum4(d4);
            const T a0a = ta * d4[0], a0b = ta * d4[1];
            const T a1a = tb * (d4[2] - t12 * a0a), a1b = tb * (d4[3] - t12 * a0b);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
              pcv[0][h] = (pcv[0][h] - a0a * w0[h]) - a1a * w1[h];
              pcv[1][h] = (pcv[1][h] - a0b * w0[h]) - a1b * w1[h];
            }
        }
        TTR_WSTAMP(pnl, 4 * owv + 3);
      }
      if (mine) {
        const int owv = wave_id;
        const int j = owv * 2, jj = j0 + j, jj1 = jj + 1;
        const bool two = j + 1 < nsteps;  // the second column of the pair takes a step as well
        T* const xb = Xp + (owv & 1) * (64 * XLD) + lane * XLD;  // this phase's exchange slot of the lane
        TTR_WSTAMP(pnl, 4 * owv + 0);
        {
            V2 v0[NH], v1[NH];
            T t0 = T(0), t1 = T(0);
            // ---- column 0: x0 = sub-column below the pivot; ||x0||^2 and x0 . c1 in one 2-value reduction
            V2 x0[NH];
            x0[0] = pcv[0][0];
            if (!(lane > jj)) x0[0].x = T(0);
#pragma unroll
            for (int h = 1; h < NH; ++h) x0[h] = pcv[0][h];
            V2 e00 = x0[0] * x0[0], e01 = x0[0] * pcv[1][0];
#pragma unroll
            for (int h = 1; h < NH; ++h) { e00 = x0[h] * x0[h] + e00; e01 = x0[h] * pcv[1][h] + e01; }
            T s00 = e00.x + e00.y, s01 = e01.x + e01.y;
            wave_sum2(s00, s01);
            const T alpha0 = lane_get(pcv[0][0].x, jj);
            T beta0, sc0;
            if (s00 < Num<T>::larfg_floor()) { beta0 = alpha0; t0 = T(0); sc0 = T(0); }  // LAPACK larfg: H = I (x = 0, or below 2^-50 of the block)
            else larfg_scalars(alpha0, s00, beta0, t0, sc0);
#pragma unroll
            for (int h = 0; h < NH; ++h) v0[h] = x0[h] * sc0;
            if (!(lane > jj)) v0[0].x = (lane == jj) ? T(1) : T(0);
            {  // H0 on column 1: v0^T c1 = sc0 * x0^T c1 + c1[jj]
              const T f = t0 * (sc0 * s01 + lane_get(pcv[1][0].x, jj));
#pragma unroll
              for (int h = 0; h < NH; ++h) pcv[1][h] = pcv[1][h] - f * v0[h];
            }
            if (lane == jj) pcv[0][0].x = beta0;  // R[jj][jj]
            // ---- column 1: ||x1||^2 and v0 . x1 (for v0^T v1) in one 2-value reduction
            T t12 = T(0);
            if (two) {
              V2 x1[NH];
              x1[0] = pcv[1][0];
              if (!(lane > jj1)) x1[0].x = T(0);
#pragma unroll
              for (int h = 1; h < NH; ++h) x1[h] = pcv[1][h];
              V2 e11 = x1[0] * x1[0], e0v = v0[0] * x1[0];
#pragma unroll
              for (int h = 1; h < NH; ++h) { e11 = x1[h] * x1[h] + e11; e0v = v0[h] * x1[h] + e0v; }
              T s11 = e11.x + e11.y, s0v = e0v.x + e0v.y;
              wave_sum2(s11, s0v);
              const T alpha1 = lane_get(pcv[1][0].x, jj1);
              T beta1, sc1;
              if (s11 < Num<T>::larfg_floor()) { beta1 = alpha1; t1 = T(0); sc1 = T(0); }
              else larfg_scalars(alpha1, s11, beta1, t1, sc1);
#pragma unroll
              for (int h = 0; h < NH; ++h) v1[h] = x1[h] * sc1;
              if (!(lane > jj1)) v1[0].x = (lane == jj1) ? T(1) : T(0);
              t12 = sc1 * s0v + lane_get(v0[0].x, jj1);  // v1 = sc1 * x1 below row jj1, 1 on it
              if (lane == jj1) pcv[1][0].x = beta1;
            } else {
#pragma unroll
              for (int h = 0; h < NH; ++h) v1[h] = V2{T(0), T(0)};
            }
            // publish the pair: four 16-byte stores per lane
            V4* const xw = reinterpret_cast<V4*>(xb);
            xw[0] = V4{v0[0].x, v0[0].y, v0[1].x, v0[1].y};
            xw[1] = V4{v0[2].x, v0[2].y, v0[3].x, v0[3].y};
            xw[2] = V4{v1[0].x, v1[0].y, v1[1].x, v1[1].y};
            xw[3] = V4{v1[2].x, v1[2].y, v1[3].x, v1[3].y};
            if (lane == 0) { taus[jj] = t0; taus[jj1] = t1; pairt[owv] = t12; }
        }
        TTR_WSTAMP(pnl, 4 * owv + 1);
        lds_barrier();
        TTR_WSTAMP(pnl, 4 * owv + 2);
        {
            // off the critical path (the other waves are applying the pair): the reflectors go to the [row][j] panel
            // image the MFMA phases read, and to the workspace (transposed: 256 contiguous bytes per store)
            // (read back from the exchange slot: keeping the pair in registers across the barrier costs spills)
#pragma unroll
            for (int q = 0; q < NW; ++q) {
              const T a = xb[q], c = xb[NW + q];
              Vs[(lane + 64 * q) * VLD + j] = a;
              Vs[(lane + 64 * q) * VLD + j + 1] = c;
              Vt[(int64_t)jj * BR + lane + 64 * q] = a;
              Vt[(int64_t)jj1 * BR + lane + 64 * q] = c;
            }
            if (lane == 0) {
              const T t0 = taus[jj], t1 = taus[jj1], t12 = pairt[owv];
              tau[jj] = t0; tau[jj1] = t1;
              // the pair's own block of T = (strict_upper(V^T V) + diag(1 / tau))^-1 and of S = V^T V
              Ts[j * VLD + j] = t0; Ts[(j + 1) * VLD + j + 1] = t1; Ts[j * VLD + j + 1] = -t0 * t12 * t1;
              Ss[j * VLD + j + 1] = t12;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // The phases of the owners to the right.  This wave's columns are final; instead of idling at the barriers it
        // assembles the panel's compact-WY factor: after barrier k it forms the 2 x 2 block S[own pair][pair k] = V_own^T V_k
        // (own reflectors from the panel image, pair k from the exchange slot; one 4-value reduction), and wave 0
        // extends T by the two columns of pair k - 1 (larft column recurrence on 16 lanes, both columns in one pass).
        // The separate S = V^T V stage (MFMA partials + block-wide reduction, two barriers) and the T construction after
        // the phases are gone from the critical path.
        // (the wave's own pair stays in registers for the S blocks: its columns are final, so the registers are free -- re-reading
        // it from the panel image cost 16 LDS reads per phase)
        V2 u0[NH], u1[NH];
        {
          const V4* const xo = reinterpret_cast<const V4*>(xb);
          const V4 o0 = xo[0], o1 = xo[1], o2 = xo[2], o3 = xo[3];
          u0[0] = V2{o0.x, o0.y}; u0[1] = V2{o0.z, o0.w}; u0[2] = V2{o1.x, o1.y}; u0[3] = V2{o1.z, o1.w};
          u1[0] = V2{o2.x, o2.y}; u1[1] = V2{o2.z, o2.w}; u1[2] = V2{o3.x, o3.y}; u1[3] = V2{o3.z, o3.w};
        }
        for (int k = owv + 1; k < nph; ++k) {
          lds_barrier();
          {
            const V4* const xr = reinterpret_cast<const V4*>(Xp + (k & 1) * (64 * XLD) + lane * XLD);
            const V4 r0 = xr[0], r1 = xr[1], r2 = xr[2], r3 = xr[3];
            const V2 w0[NH] = {V2{r0.x, r0.y}, V2{r0.z, r0.w}, V2{r1.x, r1.y}, V2{r1.z, r1.w}};
            const V2 w1[NH] = {V2{r2.x, r2.y}, V2{r2.z, r2.w}, V2{r3.x, r3.y}, V2{r3.z, r3.w}};
            V2 e0 = u0[0] * w0[0], e1 = u0[0] * w1[0], e2 = u1[0] * w0[0], e3 = u1[0] * w1[0];
#pragma unroll
            for (int h = 1; h < NH; ++h) {
              e0 = u0[h] * w0[h] + e0; e1 = u0[h] * w1[h] + e1;
              e2 = u1[h] * w0[h] + e2; e3 = u1[h] * w1[h] + e3;
            }
            T d4[4] = {e0.x + e0.y, e1.x + e1.y, e2.x + e2.y, e3.x + e3.y};
            wave_sum4(d4);
            if (lane == 0) {
              Ss[j * VLD + 2 * k] = d4[0]; Ss[j * VLD + 2 * k + 1] = d4[1];
              Ss[(j + 1) * VLD + 2 * k] = d4[2]; Ss[(j + 1) * VLD + 2 * k + 1] = d4[3];
            }
          }
          if (owv == 0 && k >= 2) t_columns(k - 1);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    } else {
    for (int owv = 0; owv < NW; ++owv) {
      if (owv * CPW < nsteps) {  // block-uniform
        if (wave_id == owv) {  // wave-uniform: local Householder QR of columns CPW*owv .. CPW*owv + CPW-1
          auto local = [&](auto OC) {
            constexpr int oc = decltype(OC)::value;
            const int j = owv * CPW + oc;
            if (j < nsteps) {
              const int jj = j0 + j;
              // ONE reduction round per step: the column norm below the diagonal and the dot products of that
              // sub-column x with the owner's remaining columns travel together; v^T c = scale * x^T c + c[jj]
              // (v = scale * x below the diagonal, 1 on it) needs no second round.
              T x[NW];
              x[0] = (lane > jj) ? pc[oc][0] : T(0);
#pragma unroll
              for (int q = 1; q < NW; ++q) x[q] = pc[oc][q];
              T d4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
              for (int cc = oc; cc < CPW; ++cc) {
                d4[cc] = x[0] * pc[cc][0];
#pragma unroll
                for (int q = 1; q < NW; ++q) d4[cc] += x[q] * pc[cc][q];
              }
              wave_sum_dpp4(d4);
              const T ss = d4[oc];
              const T alpha = lane_get(pc[oc][0], jj);
              T beta, tj, scale;
              if (ss < Num<T>::larfg_floor()) {  // LAPACK larfg: H = I (x = 0, or below 2^-50 of the block)
                beta = alpha; tj = T(0); scale = T(0);
              } else {
                larfg_scalars(alpha, ss, beta, tj, scale);
              }
              T v[NW];
              v[0] = (lane > jj) ? x[0] * scale : (lane == jj ? T(1) : T(0));
#pragma unroll
              for (int q = 1; q < NW; ++q) v[q] = x[q] * scale;
#pragma unroll
              for (int q = 0; q < NW; ++q) {
                Vs[(lane + 64 * q) * VLD + j] = v[q];
                Vt[(int64_t)jj * BR + lane + 64 * q] = v[q];  // coalesced, fire and forget
              }
              if constexpr (oc < CPW - 1) {  // apply to the owner's remaining columns
#pragma unroll
                for (int cc = oc + 1; cc < CPW; ++cc) {
                  const T f = tj * (scale * d4[cc] + lane_get(pc[cc][0], jj));
#pragma unroll
                  for (int q = 0; q < NW; ++q) pc[cc][q] -= f * v[q];
                }
              }
              if (lane == jj) pc[oc][0] = beta;  // R[jj][jj]
              if (lane == 0) { tau[jj] = tj; taus[jj] = tj; }
            }
          };
          local(IC<0>{}); local(IC<1>{});
          if constexpr (CPW > 2) { local(IC<2>{}); local(IC<3>{}); }
        }
        lds_barrier();  // the owner's reflectors (columns of Vs) and their taus are visible
        if (wave_id > owv && j0 + wave_id * CPW < n) {  // waves right of the owner apply the reflectors
#pragma unroll
          for (int rf = 0; rf < CPW; ++rf) {
            const int j = owv * CPW + rf;
            if (j < nsteps) {
              T v[NW];
#pragma unroll
              for (int q = 0; q < NW; ++q) v[q] = Vs[(lane + 64 * q) * VLD + j];
              const T tj = taus[j0 + j];
              T d4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
              for (int cc = 0; cc < CPW; ++cc) {
                d4[cc] = v[0] * pc[cc][0];
#pragma unroll
                for (int q = 1; q < NW; ++q) d4[cc] += v[q] * pc[cc][q];
              }
              wave_sum_dpp4(d4);
#pragma unroll
              for (int cc = 0; cc < CPW; ++cc) {
                const T f = tj * d4[cc];  // columns >= n are all-zero: d = 0, no effect
#pragma unroll
                for (int q = 0; q < NW; ++q) pc[cc][q] -= f * v[q];
              }
            }
          }
        }
      }
    }
    }
    stamp();
    {
      // R rows of this panel: row i (< 64) of column 4w+cc sits in lane i, q = 0
      T* __restrict__ Ro = p.Rout + bt * p.strideR + (p.top ? 0 : (int64_t)b * n * p.ldr);
      const int rr = p.top ? kb : n;
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        const int c = j0 + wave_id * CPW + cc;
        if (c < n && lane < rr) {
          T rv;
          if constexpr (PAIR) rv = (lane <= c && lane < kb) ? (cc == 0 ? pcv[0][0].x : pcv[1][0].x) : T(0);
          else rv = (lane <= c && lane < kb) ? pc[cc][0] : T(0);
          if constexpr (sizeof(T) == 4) { if (bexp != 0 && !(!PUSHED && p.top && p.expo_acc)) rv = ldexpf((float)rv, bexp); }
          Ro[(int64_t)lane * p.ldr + c] = rv;
        }
      }
    }
    if constexpr (PAIR) {
      // the owners stored their pairs themselves; columns no phase covered are identity reflectors
      // (a rank-skipped panel is recognised by its zero taus: the apply kernel never reads its reflectors or its T)
      if (!rank_skip)
        for (int j = (nsteps + 1) & ~1; j < PW; ++j) Vt[(int64_t)(j0 + j) * BR + tid] = T(0);
      if (tid < PW && tid >= ((nsteps + 1) & ~1)) tau[j0 + tid] = T(0);
    } else {
      for (int j = nsteps; j < PW; ++j) {  // identity reflectors: keep the stored factors well defined
        Vt[(int64_t)(j0 + j) * BR + tid] = T(0);
        if (tid == 0) tau[j0 + j] = T(0);
      }
    }
    lds_barrier();
    // (5) the triangular factor T of the panel
    if constexpr (PAIR) {
      // S and T were assembled during the phases; what is left are the two columns of the LAST pair (their S blocks were
      // written after the last phase barrier, i.e. before the barrier above) and the copy for the apply kernel
      if (wave_id == 0) {
        const int nph = (nsteps + 1) >> 1;
        if (nph >= 2) t_columns(nph - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!rank_skip)
          for (int e = lane; e < PW * PW; e += 64) Tg[pnl * PW * PW + e] = Ts[(e >> 4) * VLD + (e & 15)];
      }
    } else {
    // S = V^T V over the block (MFMA, K = BR split over the waves), then T
    {
      Acc s4[4] = {M::zero(), M::zero(), M::zero(), M::zero()};  // four chains: a dependent MFMA waits ~40 cycles
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const T a = Vs[(wave * 64 + ks * 4 + g) * VLD + cl];
        s4[ks & 3] = M::mma(a, a, s4[ks & 3]);
      }
      const Acc s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) Wp[wave][M::row(lane, r)][cl] = s[r];
    }
    lds_barrier();
    if (tid < PW * PW) {
      const int i = tid >> 4, k = tid & 15;
      T acc_s = (Wp[0][i][k] + Wp[1][i][k]) + (Wp[2][i][k] + Wp[3][i][k]);
      if constexpr (NW == 8) acc_s += (Wp[4][i][k] + Wp[5][i][k]) + (Wp[6][i][k] + Wp[7][i][k]);
      Ss[i * VLD + k] = acc_s;
    }
    lds_barrier();
    if (wave_id == 0) {
      // The triangular factor T (larft): T = (strict_upper(S) + diag(1 / tau))^-1, built by recursive doubling instead
      // of the column recurrence (16 dependent steps of up to 15 FMAs on 16 lanes: ~5 k cycles per panel, measured):
      // two diagonal blocks T11, T22 of width h are joined by  T12 = -T11 S12 T22  for h = 1, 2, 4, 8 -- two small
      // products per level, one output element per lane, wave-local LDS exchange (no workgroup barrier).
      for (int e = lane; e < PW * PW; e += 64) {
        const int i = e >> 4, k = e & 15;
        Ts[i * VLD + k] = (i == k) ? taus[j0 + i] : T(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int h = 1; h < PW; h <<= 1) {
        const int hh = h * h;
        const int bq = lane / hh, rr2 = lane % hh, i = rr2 / h, jx = rr2 % h;
        const int o = bq * 2 * h;           // the block pair covers rows / columns o .. o + 2h - 1
        const bool act = lane < 8 * h;      // 16 / (2h) block pairs x h^2 outputs
        if (act) {                          // X = S12 T22
          T x = T(0);
#pragma unroll
          for (int k = 0; k < h; ++k) x += Ss[(o + i) * VLD + o + h + k] * Ts[(o + h + k) * VLD + o + h + jx];
          Xs[(o + i) * VLD + jx] = x;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (act) {                          // T12 = -T11 X
          T t = T(0);
#pragma unroll
          for (int k = 0; k < h; ++k) t += Ts[(o + i) * VLD + o + k] * Xs[(o + k) * VLD + jx];
          Ts[(o + i) * VLD + o + h + jx] = -t;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      for (int e = lane; e < PW * PW; e += 64) Tg[pnl * PW * PW + e] = Ts[(e >> 4) * VLD + (e & 15)];
    }
    }
    // (6) W = V^T A2, per-wave partial over its 64 rows; the accumulator registers are the B operand.  W does not
    // depend on T: it is formed in the same barrier interval as the (serial, 16-lane) larft recurrence above, so
    // the other waves' MFMAs run under wave 0's recurrence instead of waiting for it.
    const bool trailing = pnl < NT - 1 && (pnl + 1) * PW < n && !rank_skip;   // (H = I: nothing to update)
    if (trailing) {
#pragma unroll
      for (int tn = pnl + 1; tn < NT; ++tn) {
        Acc wa = M::zero();
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int s = 0; s < 4; ++s) wa = M::mma(Vs[rowl(tm, s) * VLD + cl], acc[tm][tn][s], wa);
#pragma unroll
        for (int r = 0; r < 4; ++r) Wp[wave][M::row(lane, r)][(tn - 1) * PW + cl] = wa[r];
      }
    }
    lds_barrier();
    stamp();
    if (trailing) {
      // W2 = -T^T (sum of the partials), in two block-wide stages: every thread sums the NW partials of its elements
      // (in place in the first partial), then forms its outputs -- it used to be 128 LDS reads per thread on a
      // quarter of the block (~2 k cycles per panel)
      {
        constexpr int c0 = pnl * PW;             // first trailing column in Wp's column index
        constexpr int ncols = NP - (pnl + 1) * PW > 0 ? NP - (pnl + 1) * PW : 1;  // (> 0 whenever `trailing`)
        for (int e = tid; e < PW * ncols; e += NTH) {
          const int k = e / ncols, jc = c0 + e % ncols;
          T acc_w = (Wp[0][k][jc] + Wp[1][k][jc]) + (Wp[2][k][jc] + Wp[3][k][jc]);
          if constexpr (NW == 8) acc_w += (Wp[4][k][jc] + Wp[5][k][jc]) + (Wp[6][k][jc] + Wp[7][k][jc]);
          Wp[0][k][jc] = acc_w;
        }
        lds_barrier();
        for (int e = tid; e < PW * ncols; e += NTH) {
          const int i = e / ncols, jc = c0 + e % ncols;
          T a2 = 0;
#pragma unroll
          for (int k = 0; k < PW; ++k) a2 += Ts[k * VLD + i] * Wp[0][k][jc];
          W2s[i][jc + PW] = -a2;
        }
      }
      lds_barrier();
      // (7) A2 += V W2
#pragma unroll
      for (int tn = pnl + 1; tn < NT; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            acc[tm][tn] = M::mma(Vs[(wave * 64 + tm * 16 + cl) * VLD + ks * 4 + g], W2s[ks * 4 + g][tn * PW + cl],
                                 acc[tm][tn]);
    }
    stamp();
    // the next panel's first barrier orders (7)'s LDS reads before Vs / W2s are rewritten
  };
  panel(IC<0>{});
  if constexpr (NT > 1) panel(IC<1>{});
  if constexpr (NT > 2) panel(IC<2>{});
  if constexpr (NT > 3) panel(IC<3>{});
  lds_barrier();
}

// ---------------------------------------------------------------- apply (form Q [C;0] top-down)
template <typename T>
struct QrApply {
  const T* Vt;
  const T* tau;
  const T* Tg;
  int64_t m;     // rows of this level's matrix
  int n;         // cols of the factored matrix
  int nb;
  int kcols;     // columns of the product being formed
  const T* Top;  // level above: row block b*n, n x kcols ; nullptr => identity
  int64_t ldtop, strideTop;
  T* Out;
  int64_t ldout, strideOut, ocs;  // element (row, col) at Out[row * ldout + col * ocs] (ocs != 1: transposed level-0 output)
  int pk, pI;    // > 0: level 0 of a PUSHED factorisation, local row (wave, kk) <-> global row kk * pI + NW*b + wave
  T* Gp;           // optional (level 0 of a pushed factorisation, fp32, pk = 64, kcols = 32, whole blocks): the block's share of the
                   // ROW GRAM matrix of Out as a pk x pk unfolding, Gp[(bt * nb + b)][pk][pk] = sum_i Out_i Out_i^T over the block's mode indices
  long long* dbg;  // optional: cycle stamps of block (0, 0) (diagnostics)
  const int32_t* pack_flag;  // level 0 of a PUSHED factorisation: != 0 for items the factor kernel packed (see QrLevel)
  int grid_swap;             // grid (batch, nb) instead of (nb, batch): block-major launch order (see QrLevel)
  const int32_t* half_zero;  // (round 6) level 1 above a PACKED pushed level 0 (TTR_KNOB_QR_PACK = 3, 8 leaf blocks): the leaf flags -- rows
                             // >= m / 2 of a flagged item are the absorbed leaves' R blocks, exactly zero, and so are its reflectors'
                             // rows there and the result's: waves 4 .. 7 of its block load, multiply and store nothing
  int skip_zero_rows;        // != 0: the exactly-zero rows kk >= 32 of a packed item's output are NOT written (the caller only reads
                             // the result through kernels that take the item's rows32 flag: ttr_rowgram / ttr_rotgram / ttr_project)
};

template <typename T, int NT, int NTC, int NW>
// (second launch bound = minimum waves per SIMD: 4 caps the fp32 kernel at 128 VGPRs, i.e. two 8-wave blocks per CU)
__global__ __launch_bounds__(64 * NW, (sizeof(T) == 4 && NW == 8 ? 4 : 1)) void qr_apply_kernel(QrApply<T> p) {
  using M = Mfma<T>;
  using Acc = typename M::Acc;
  constexpr int NP = PW * NT;
  constexpr int NC = PW * NTC;
  constexpr int BR = 64 * NW;
  // reflector panel [row][j]; fp32: leading dimension 20 -- the 16 values of a row go in as four 16-byte writes, and the
  // A-operand reads of both products (rows 4 g + s at column cl: banks 16 g + cl; row cl at column 4 ks + g: banks 20 cl + g)
  // are conflict-free (17 leaves the W product's reads 4-way conflicted: 68 g + cl)
  constexpr int AVLD = sizeof(T) == 4 ? 20 : VLD;
  __shared__ __attribute__((aligned(16))) T Vs[BR * AVLD];
  __shared__ T Ts[PW * VLD];   // compact-WY factor T of the current panel
  __shared__ T Wp[NW][PW][NC];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int cl = lane & 15, g = lane >> 4;
  int b;
  int64_t bt;
  block_of(p.grid_swap, p.nb, b, bt);
  int64_t row0;
  int rows;
  block_rows(p.m, p.nb, b, row0, rows);
  const int n = p.n;
  const int kb = rows < n ? rows : n;
  const int kc = p.kcols;
  auto rowl = [&](int tm, int reg) { return wave * 64 + tm * 16 + M::row(lane, reg); };

  const int64_t blk = bt * p.nb + b;
  const T* __restrict__ Vt = p.Vt + blk * (int64_t)NP * BR;
  const T* __restrict__ Tg = p.Tg + blk * (int64_t)NT * PW * PW;

#ifdef TTR_QR_WSTAMPS
  int dbgi = 0;
  auto astamp = [&]() { if (p.dbg && b == 0 && bt == 0 && tid == 0) p.dbg[dbgi++] = (long long)clock64(); };
#else
  auto astamp = [&]() {};
#endif
  astamp();
  // the first panel's reflectors are requested before C is initialised: their HBM latency hides the Top loads
  // (round 6: the block's taus are requested TOGETHER with the packing flag -- the flag, the taus and the reflectors were three
  // dependent memory round trips at the head of every block, 17 - 23 k of its 43 k cycles under load; every block owns a tau slot,
  // an absorbed one just drops the value)
  const T* __restrict__ tq = p.tau + blk * (int64_t)NP;
  const T tv = lane < NP ? tq[lane] : T(0);
  const int pmode = p.pack_flag ? p.pack_flag[bt] : 0;
  const bool packed = pmode != 0;   // block-uniform
  const int half_nb = p.nb >> 1;
  const bool pmode2 = pmode == 2;
  if (packed && (pmode == 3 ? (b >= half_nb) : pmode2 ? ((b & 1) != 0) != ((bt & 1) != 0) : ((b >= half_nb) != ((bt & 1) != 0)))) return; // absorbed block: its rows are written by its partner block
  int npanels = (kb + PW - 1) / PW;
  {
    // trailing panels whose taus are all zero are H = I (rank-skipped by the factor kernel, or never factored): they are not
    // loaded at all.  Every wave looks at the block's taus itself (lane = reflector index): wave-uniform without LDS.
    const unsigned long long livem = __ballot(tv != T(0));
    const int nlive = livem ? (63 - __builtin_clzll(livem)) / PW + 1 : 0;
    npanels = npanels < nlive ? npanels : nlive;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(tid) >> 6;
  const bool idle_wave = NW == 8 && p.half_zero && p.half_zero[bt] == 3 && wave_u >= NW / 2;   // (wave-uniform; see QrApply::half_zero)
  T vreg[PW], treg = T(0);
  if (npanels > 0) {
#pragma unroll
    for (int j = 0; j < PW; ++j) vreg[j] = idle_wave ? T(0) : Vt[(int64_t)((npanels - 1) * PW + j) * BR + tid];
    if (tid < PW * PW) treg = Tg[(npanels - 1) * PW * PW + tid];
  }
  Acc C[4][NTC];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tc = 0; tc < NTC; ++tc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowl(tm, r), col = tc * PW + cl;
        T v = T(0);
        if (col < kc) {
          if (p.Top) {
            if (row < n && row < rows) v = p.Top[bt * p.strideTop + ((int64_t)b * n + row) * p.ldtop + col];
          } else {
            v = (row == col) ? T(1) : T(0);
          }
        }
        C[tm][tc][r] = v;
      }

  for (int pnl = npanels - 1; pnl >= 0; --pnl) {
    // stage the panel's reflectors ([row][j]) and its T factor; the next panel's loads fly under the MFMAs
#pragma unroll
    for (int j = 0; j < PW; ++j) Vs[tid * AVLD + j] = vreg[j];
    if (tid < PW * PW) Ts[(tid >> 4) * VLD + (tid & 15)] = treg;
    if (pnl > 0) {
      if (!idle_wave) {
#pragma unroll
        for (int j = 0; j < PW; ++j) vreg[j] = Vt[(int64_t)((pnl - 1) * PW + j) * BR + tid];
      }
      if (tid < PW * PW) treg = Tg[(pnl - 1) * PW * PW + tid];
    }
    lds_barrier();
    astamp();
    // W = V^T C (per-wave partial over its 64 rows).  Before the first panel is applied C is [Top; 0]: a wave whose 64 rows
    // lie below the n rows of Top contributes exactly zero and skips its MFMAs (7 of the 8 waves of a 512-row block).
    const bool c_is_zero = idle_wave || (pnl == npanels - 1 && wave_u * 64 >= n);  // wave-uniform
#pragma unroll
    for (int tc = 0; tc < NTC; ++tc) {
      Acc wa = M::zero();
      if (!c_is_zero) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int s = 0; s < 4; ++s) wa = M::mma(Vs[rowl(tm, s) * AVLD + cl], C[tm][tc][s], wa);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) Wp[wave][M::row(lane, r)][tc * PW + cl] = wa[r];
    }
    lds_barrier();
    astamp();
    // the waves' partial W's are summed in place (one entry per thread)
    {
      for (int e = tid; e < PW * NC; e += BR) {
        const int k = e / NC, jc = e % NC;
        T acc_w = (Wp[0][k][jc] + Wp[1][k][jc]) + (Wp[2][k][jc] + Wp[3][k][jc]);
        if constexpr (NW == 8) acc_w += (Wp[4][k][jc] + Wp[5][k][jc]) + (Wp[6][k][jc] + Wp[7][k][jc]);
        Wp[0][k][jc] = acc_w;
      }
    }
    lds_barrier();
    astamp();
    // W2 = -T W on the matrix cores, by every wave for itself (8 MFMAs), and C += V W2 with the W2 accumulators as the B operand
    // of the update (K step s = register s of every lane group, i.e. reflector index M::row(lane, s): V is read at that column).
    // The block-wide second stage (16 FMAs and 32 LDS reads per thread), the W2 image and one barrier per panel are gone.
    {
      T ta[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ta[ks] = -Ts[cl * VLD + ks * 4 + g];
      if (!idle_wave) {   // (an idle wave's reflector rows are zero: its rows of the product stay zero)
#pragma unroll
        for (int tc = 0; tc < NTC; ++tc) {
          Acc w2 = M::zero();
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) w2 = M::mma(ta[ks], Wp[0][ks * 4 + g][tc * PW + cl], w2);
#pragma unroll
          for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int sI = 0; sI < 4; ++sI)
              C[tm][tc] = M::mma(Vs[(wave * 64 + tm * 16 + cl) * AVLD + M::row(lane, sI)], w2[sI], C[tm][tc]);
        }
      }
    }
    lds_barrier();  // Vs / Ts / Wp are rewritten by the next panel
    astamp();
  }

  // (round 6) 16-byte stores of whole 128-byte rows for the 32-column results of the sweep (fp32, NTC = 2, ocs = 1): the wave's
  // tiles go through a wave-private LDS image [32 rows][36] (Vs is free after the last panel; ld 36: conflict-free 4-byte
  // writes, 16-byte aligned rows) one row-tile PAIR at a time and leave as 8 rows x 128 bytes per instruction -- 8 stores per
  // lane instead of 32 that each touch four 64-byte pieces 8 KB apart (cycle stamps: the store tail was 9.4 k of a block's 43 k
  // cycles under load, profiles/r06_apply_stamps.txt).  `orow(kk)`: element offset of result row kk of this wave's mode index.
  constexpr bool kWideOut = sizeof(T) == 4 && NTC == 2;
  auto store_pair_wide = [&](int tm0, T* __restrict__ obase, int64_t kstride, int kk0, bool zero_hi) {
    if constexpr (kWideOut) {
      constexpr int OLD = 36;
      T* img = Vs + wave * (32 * OLD);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tc = 0; tc < NTC; ++tc) img[(h * 16 + M::row(lane, r)) * OLD + tc * PW + cl] = C[tm0 + h][tc][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-private image: no workgroup barrier)
      typedef T V4 __attribute__((ext_vector_type(4)));
      const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rr;
        const V4 v = *reinterpret_cast<const V4*>(&img[row * OLD + c4]);
        *reinterpret_cast<V4*>(obase + (int64_t)(kk0 + row) * kstride + c4) = v;
        if (zero_hi) *reinterpret_cast<V4*>(obase + (int64_t)(kk0 + row + 32) * kstride + c4) = V4{T(0), T(0), T(0), T(0)};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is rewritten by the next pair
    }
  };
  const bool wide_ok = kWideOut && p.ocs == 1 && (p.ldout & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.Out) & 15) == 0) &&
                       ((p.strideOut & 3) == 0);

  if (p.pI > 0 && packed) {
    // packed item (factor kernel: QrLevel::pack_flag): local rows 0..31 of wave w are (kk, i0 = NW b + w), rows 32..63 are
    // (kk, i1 = NW partner + w), kk = 0..31; the rows kk >= 32 of both mode indices were dropped as zeros -- written here
    T* __restrict__ Out = p.Out + bt * p.strideOut;
    const int64_t kstride = (int64_t)p.pI * p.ldout;
    const int i0 = b * NW + wave, i1 = (pmode2 ? (b ^ 1) : (b < half_nb ? b + half_nb : b - half_nb)) * NW + wave;
    if (kc == NC && wide_ok) {
      store_pair_wide(0, Out + (int64_t)i0 * p.ldout, kstride, 0, !p.skip_zero_rows);
      store_pair_wide(2, Out + (int64_t)i1 * p.ldout, kstride, 0, !p.skip_zero_rows);
    } else if (kc == NC) {  // every store valid: one lane pointer per mode index, wave-uniform row offsets (as the unpacked fast path)
      T* __restrict__ o0 = Out + (int64_t)i0 * p.ldout + (int64_t)M::row(lane, 0) * kstride + cl;
      T* __restrict__ o1 = Out + (int64_t)i1 * p.ldout + (int64_t)M::row(lane, 0) * kstride + cl;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        T* __restrict__ o = tm < 2 ? o0 : o1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tc = 0; tc < NTC; ++tc) {
            o[(int64_t)((tm & 1) * 16 + M::row(0, r)) * kstride + tc * PW] = C[tm][tc][r];
            if (!p.skip_zero_rows) o[(int64_t)((tm & 1) * 16 + M::row(0, r) + 32) * kstride + tc * PW] = T(0);
          }
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        T* __restrict__ o = Out + (int64_t)(tm < 2 ? i0 : i1) * p.ldout + cl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t kk = (tm & 1) * 16 + M::row(lane, r);
#pragma unroll
          for (int tc = 0; tc < NTC; ++tc) {
            if (tc * PW + cl < kc) {
              o[kk * kstride + tc * PW] = C[tm][tc][r];
              if (!p.skip_zero_rows) o[(kk + 32) * kstride + tc * PW] = T(0);
            }
          }
        }
      }
    }
    astamp();
    return;
  }
  if (p.pI > 0) {
    T* __restrict__ Out = p.Out + bt * p.strideOut;
    const int imode = b * NW + wave;
    const bool full_tile = p.pk == 64 && kc == NC && (b + 1) * NW <= p.pI;  // block-uniform
    if (full_tile) {
      // every store valid: one lane pointer, wave-uniform row offsets (the guarded element-wise loop below costs 3x the cycles)
      const int64_t kstride = (int64_t)p.pI * p.ldout;
      if (wide_ok && !p.Gp) {
        store_pair_wide(0, Out + (int64_t)imode * p.ldout, kstride, 0, false);
        store_pair_wide(2, Out + (int64_t)imode * p.ldout, kstride, 32, false);
      } else {
        T* __restrict__ o = Out + (int64_t)imode * p.ldout + (int64_t)M::row(lane, 0) * kstride + cl;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int tc = 0; tc < NTC; ++tc) o[(int64_t)(tm * 16 + M::row(0, r)) * kstride + tc * PW] = C[tm][tc][r];
      }
      if constexpr (sizeof(T) == 4 && NTC == 2 && NW == 8) {
        if (p.Gp) {
          // Row Gram matrix of the block's output while it is still in registers: G_b = sum over the 8 waves (mode indices) of
          // C_w C_w^T (64 x 64, K = 8 x 32).  The C tiles hold columns on the lanes; both MFMA operands of a Gram tile want
          // ROWS on the lanes, so each wave writes its tile to an LDS image [64 rows][16 columns] (stride 20: conflict-free b32
          // writes, 16-byte aligned conflict-free b128 reads), 16 columns at a time; the 10 upper-triangle tiles are owned by
          // waves (wave w: tile w, waves 0 and 1 also tiles 8 and 9) and summed over all 8 images -- no cross-wave reduction.
          constexpr int GLD = 20, IMG = 64 * GLD;
          static_assert(6 * IMG <= BR * AVLD && 2 * IMG <= NW * PW * NC, "Gram images must fit Vs / Wp");
          auto image = [&](int w) -> T* { return w < 6 ? &Vs[w * IMG] : &Wp[0][0][0] + (w - 6) * IMG; };
          auto tile_of = [](int idx, int& ti, int& tj) {  // row-major over the upper triangle of the 4 x 4 tile grid
            ti = 0;
            while (idx >= 4 - ti) { idx -= 4 - ti; ++ti; }
            tj = ti + idx;
          };
          int ti0, tj0, ti1 = 0, tj1 = 0;
          tile_of(wave_u, ti0, tj0);
          const bool two = wave_u < 2;
          if (two) tile_of(wave_u + 8, ti1, tj1);
          Acc ga = M::zero(), gb = M::zero();
          T* mine = image(wave_u);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (h) lds_barrier();  // every reader of the first column half is done (h = 0: the panel loop ended on a barrier)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
              for (int r = 0; r < 4; ++r) mine[(tm * 16 + 4 * g + r) * GLD + cl] = C[tm][h][r];
            lds_barrier();
            for (int w = 0; w < NW; ++w) {
              const T* im = image(w);
              const float4 a = *reinterpret_cast<const float4*>(&im[(16 * ti0 + cl) * GLD + 4 * g]);
              const float4 bb = *reinterpret_cast<const float4*>(&im[(16 * tj0 + cl) * GLD + 4 * g]);
              ga = M::mma(a.x, bb.x, ga); ga = M::mma(a.y, bb.y, ga); ga = M::mma(a.z, bb.z, ga); ga = M::mma(a.w, bb.w, ga);
              if (two) {
                const float4 a1 = *reinterpret_cast<const float4*>(&im[(16 * ti1 + cl) * GLD + 4 * g]);
                const float4 b1 = *reinterpret_cast<const float4*>(&im[(16 * tj1 + cl) * GLD + 4 * g]);
                gb = M::mma(a1.x, b1.x, gb); gb = M::mma(a1.y, b1.y, gb); gb = M::mma(a1.z, b1.z, gb); gb = M::mma(a1.w, b1.w, gb);
              }
            }
          }
          T* __restrict__ Go = p.Gp + (bt * p.nb + b) * (int64_t)(64 * 64);
          auto put = [&](const Acc& t, int ti, int tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * ti + 4 * g + r, col = 16 * tj + cl;
              Go[row * 64 + col] = t[r];
              if (ti != tj) Go[col * 64 + row] = t[r];
            }
          };
          put(ga, ti0, tj0);
          if (two) put(gb, ti1, tj1);
        }
      }
      astamp();
      return;
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tc = 0; tc < NTC; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = tm * 16 + M::row(lane, r), col = tc * PW + cl;
          if (kk < p.pk && imode < p.pI && col < kc) Out[((int64_t)kk * p.pI + imode) * p.ldout + col] = C[tm][tc][r];
        }
  } else {
    T* __restrict__ Out = p.Out + bt * p.strideOut + row0 * p.ldout;
    if (idle_wave) {
      // (rows of absorbed leaves: the level below never reads them)
    } else if (wide_ok && kc == NC && rows == BR) {   // a whole block of a plain level (the sweep's level 1): the wave's 64 rows as two pairs
      T* __restrict__ ow = Out + (int64_t)(wave * 64) * p.ldout;
      store_pair_wide(0, ow, p.ldout, 0, false);
      store_pair_wide(2, ow, p.ldout, 32, false);
    } else {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tc = 0; tc < NTC; ++tc)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = rowl(tm, r), col = tc * PW + cl;
            if (row < rows && col < kc) Out[(int64_t)row * p.ldout + (int64_t)col * p.ocs] = C[tm][tc][r];
          }
    }
  }
  astamp();
}

// ---------------------------------------------------------------- host-side tree
struct QrPlan {
  int levels;
  int npad;  // 16 * NT
  int64_t m[16];
  int nb[16];
  int nw[16];  // waves per block on this level (block rows = 64 * nw)
  // workspace offsets in elements
  int64_t off_vt[16], off_tau[16], off_tg[16], off_x[16], off_out[16];
  int64_t off_flag;  // one int32 per item (stored in an element slot): level-0 row packing of a pushed factorisation
  int64_t total;  // elements
};

static int nt_for(int64_t n) { return n <= 16 ? 1 : (n <= 32 ? 2 : 4); }
int g_qr_f64_nw4 = 0;  // ttr_debug_set_knob(TTR_KNOB_QR_F64_NW4): fp64 trees out of 256-row (4-wave) blocks only
// fp64: the 8-wave block needs 132 KB of LDS and 237 VGPRs (ONE block, i.e. one Householder chain, per CU); two 4-wave
// blocks would fit (75 KB, 2 waves per SIMD).  Measured on config C2's resident batch (tools/c2_ab.py): 256 trains in
// 40.9 ms with the 4-wave blocks against 34.0 ms with the 8-wave blocks (qr_factor 37.1 vs 25.5 ms) -- the pair phases of
// the 8-wave block and the shallower tree outweigh the second chain per CU.  Kept as a switch, off.
// (bit 1 of the switch: the same for fp32 -- round 4's A/B of "256-row leaves x 4 waves, four blocks per CU" on the metric)
static bool nw4_forced(bool f64) { return f64 ? (g_qr_f64_nw4 & 1) != 0 : (g_qr_f64_nw4 & 2) != 0; }
// (bit 2 of the knob, A/B: fp32 matrices of <= 256 rows on the 8-wave PAIR kernel too -- its panels are rank-revealing, the 4-wave kernel's are not)
static int nw_for(int64_t rows, bool f64) { return ((rows > BR4 || (!f64 && (g_qr_f64_nw4 & 4) != 0)) && !nw4_forced(f64)) ? 8 : 4; }

static QrPlan make_plan(int64_t m, int64_t n, int64_t batch, bool f64) {
  QrPlan pl{};
  const int NT = nt_for(n);
  pl.npad = NT * PW;
  int64_t cur = m;
  int L = 0;
  for (;;) {
    pl.m[L] = cur;
    pl.nw[L] = nw_for(cur, f64);
    pl.nb[L] = (int)ceil_div(cur, 64 * pl.nw[L]);
    ++L;
    if (pl.nb[L - 1] <= 1) break;
    cur = (int64_t)pl.nb[L - 1] * n;
  }
  pl.levels = L;
  int64_t off = 0;
  for (int l = 0; l < L; ++l) {
    pl.off_vt[l] = off; off += batch * pl.nb[l] * pl.npad * 64 * pl.nw[l];
    pl.off_tau[l] = off; off += align_up(batch * pl.nb[l] * pl.npad, 64);
    pl.off_tg[l] = off; off += batch * pl.nb[l] * NT * PW * PW;
    if (l > 0) {
      pl.off_x[l] = off; off += batch * pl.m[l] * n;    // stacked R factors (input of level l)
      pl.off_out[l] = off; off += batch * pl.m[l] * n;  // Q slab of level l (m_l x kcols, kcols <= n)
    }
  }
  pl.off_flag = off; off += align_up(batch, 64);
  pl.total = off;
  return pl;
}

// The batch is a grid dimension (<= 65535): larger batches are processed in slices of kMaxBatchSlice items, each with its
// own plan; the slices' workspaces follow each other, so the factor and the apply calls of one (m, n, batch) agree on the
// layout without any extra state.
constexpr int64_t kMaxBatchSlice = 65535;

int64_t qr_workspace_bytes(int dtype, int64_t m, int64_t n, int64_t batch) {
  if (m <= 0 || n <= 0 || batch <= 0) return 0;
  const int64_t elem = dtype == TTR_F64 ? 8 : 4;
  int64_t total = 0;
  for (int64_t b0 = 0; b0 < batch; b0 += kMaxBatchSlice)
    total += make_plan(m, n, batch - b0 < kMaxBatchSlice ? batch - b0 : kMaxBatchSlice, dtype == TTR_F64).total * elem;
  return total;
}

long long* g_qr_dbg = nullptr;  // set through ttr_debug_set_qr_stamps (diagnostics only)
int g_qr_dbg_bx = 0, g_qr_dbg_by = 0;  // which level-0 block stamps (ttr_debug_set_knob: a block in the middle of the grid shows the
                                        // steady state -- block (0, 0) starts together with every other first-wave block)
// ttr_debug_set_knob(TTR_KNOB_QR_PACK): pushed level-0 blocks pack two mode indices per wave when Rm has numerical rank <= 32.
// 3 (default): blocks b >= nb / 2 are absorbed by b - nb / 2 and the launch is BLOCK-MAJOR (grid (batch, nb)): all working
// blocks are dispatched first, the absorbed ones (which return after ~12 k cycles) last.  1 / 2 (item-major grid, the working
// half alternating with the item / even-odd blocks): measured WITHOUT gain -- the dispatcher stalls when long and short
// workgroups alternate in launch order (level-0 launch 1.78 ms against 1.60 ms unpacked, although the working blocks cost
// what an unpacked block costs and there are half as many); block-major: 1.1 ms, the metric step 20.9 -> 17.5 ms
// (tools/probes/qr_pack_stamps.py, profiles/r04_qr_pack_ab.txt).  0 = never pack.
int g_qr_pack = 3;
int g_qr_stagger = 0;      // ttr_debug_set_knob(TTR_KNOB_QR_STAGGER): see QrLevel::stagger_kc
int g_qr_interleave = 1;   // ttr_debug_set_knob(TTR_KNOB_QR_INTERLEAVE): see block_of (0 = round 4's block-major order, A/B)
int g_rank_skip_c = 8;   // ttr_debug_set_knob(TTR_KNOB_QR_RANK_SKIP, c): threshold factor of the rank-revealing early exit (0 = off)
int g_qr_variant = 1;           // ttr_debug_set_knob(TTR_KNOB_QR_PANEL): 1 = pair steps in the 8-wave blocks (default), 0 = one reflector at a time

struct Pushed {  // level-0 operands of a fused push (nullptr Rm: plain factorisation)
  const void* Rm = nullptr;
  int64_t ldrm = 0, strideRm = 0;
  const void* Cn = nullptr;
  int64_t strideCn = 0;
  const void* Cn2 = nullptr;  // block-diagonal core: second block
  int64_t strideCn2 = 0;
  int sumRa = 0, sumCa = 0;
  int k = 0, Rin = 0, I = 0;
  int32_t* expo_acc = nullptr;  // QrLevel::expo_acc of the top level (plain factorisations too)
};

// R <- R 2^-e in place, e = exponent of the item's largest entry, expo_acc[item] += e: the top level of a PUSHED factorisation
// that has a single level (a handful of mode indices: never the hot path) -- see QrLevel::expo_acc
template <typename T>
__global__ __launch_bounds__(256) void r_expo_kernel(T* __restrict__ R, int rows, int n, int64_t ldr, int64_t strideR,
                                                     int32_t* __restrict__ expo_acc) {
  __shared__ float red[4];
  const int64_t bt = blockIdx.x;
  T* __restrict__ Rb = R + bt * strideR;
  float mx = 0.f;
  for (int idx = threadIdx.x; idx < rows * n; idx += 256) mx = fmaxf(mx, fabsf((float)Rb[(int64_t)(idx / n) * ldr + idx % n]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  int e = 0;
  if (mx > 0.f && mx < 3e38f) (void)frexpf(mx, &e);
  if (e != 0)
    for (int idx = threadIdx.x; idx < rows * n; idx += 256) {
      T* q = Rb + (int64_t)(idx / n) * ldr + idx % n;
      *q = (T)ldexpf((float)*q, -e);
    }
  if (threadIdx.x == 0) expo_acc[bt] += e;
}

// pack_flag[item] = pack_ok when rows 32 .. 63 of the item's 64 x 64 Rm hold at most (c eps)^2 of its squared Frobenius norm (the
// packing criterion of the fused push, QrLevel::pack_flag), else 0: one wave per item, ahead of the level-0 launch (QrLevel::pack_pre)
template <typename T>
__global__ __launch_bounds__(512) void pack_flags_kernel(const T* __restrict__ Rm, int64_t ldrm, int64_t strideRm, int Rin, int64_t batch,
                                                         int rank_skip_c, int pack_ok, int32_t* __restrict__ flag) {
  // one 512-thread workgroup per item, summing in EXACTLY the order of the factor kernel's own test (thread (wave w, lane l): rows
  // w + 8 e of column l, e ascending; wave_sum2; the waves' sums added in wave order): a train takes the same decision whether it
  // travels in a batch that decides ahead of the launch or in a small one whose blocks decide for themselves
  __shared__ T sa_w[8], sl_w[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t bt = blockIdx.x;
  const T* __restrict__ R = Rm + bt * strideRm;
  T sq_all = T(0), sq_low = T(0);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kk = w + 8 * e;
    const T rv = lane < Rin ? R[(int64_t)kk * ldrm + lane] : T(0);
    sq_all += rv * rv;
    if (kk >= 32) sq_low += rv * rv;
  }
  wave_sum2(sq_all, sq_low);
  if (lane == 0) { sa_w[w] = sq_all; sl_w[w] = sq_low; }
  __syncthreads();
  if (tid == 0) {
    T sa = sa_w[0], sl = sl_w[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) { sa += sa_w[k]; sl += sl_w[k]; }
    const T ce = T(rank_skip_c) * Num<T>::eps();
    flag[bt] = (sl <= ce * ce * sa) ? pack_ok : 0;
  }
}

int g_qr_l1_idle = 1;     // (round 6) level-1 apply of packed items: waves 4 .. 7 idle (QrApply::half_zero); 0 with TTR_KNOB_QR_PACK_PRE = 2 (A/B)
int g_qr_pack_pre_min = 256;   // smallest batch that takes the packing decision ahead of the launch
int g_qr_pack_pre = 1;   // ttr_debug_set_knob(TTR_KNOB_QR_PACK_PRE): 0 = every block derives the packing decision from Rm itself (round 5)

template <typename T, int NT>
static int factor_run(int64_t m, int n, int64_t batch, const T* A, int64_t lda, int64_t strideA, T* R, int64_t ldr,
                      int64_t strideR, T* ws, const QrPlan& pl, const Pushed& pu, hipStream_t stream, int64_t a_cs = 1) {
  const int L = pl.levels;
  for (int l = 0; l < L; ++l) {
    QrLevel<T> p;
    p.dbg = (l == 0) ? g_qr_dbg : nullptr;
    p.dbg_bx = g_qr_dbg_bx; p.dbg_by = g_qr_dbg_by;
    p.rank_skip_c = g_rank_skip_c;
    p.pack_flag = (l == 0 && pu.Rm) ? reinterpret_cast<int32_t*>(ws + pl.off_flag) : nullptr;
    p.pack_ok = (g_qr_pack && pu.Cn2 == nullptr) ? g_qr_pack : 0;
    p.Rm = (const T*)pu.Rm; p.ldrm = pu.ldrm; p.strideRm = pu.strideRm;
    p.Cn = (const T*)pu.Cn; p.strideCn = pu.strideCn; p.pk = pu.k; p.pRin = pu.Rin; p.pI = pu.I;
    p.Cn2 = (const T*)pu.Cn2; p.strideCn2 = pu.strideCn2; p.sumRa = pu.sumRa; p.sumCa = pu.sumCa;
    p.X = l == 0 ? A : ws + pl.off_x[l];
    p.ldx = l == 0 ? lda : n;
    p.xcs = l == 0 ? a_cs : 1;
    p.strideX = l == 0 ? strideA : pl.m[l] * n;
    p.m = pl.m[l]; p.n = n; p.nb = pl.nb[l];
    p.Vt = ws + pl.off_vt[l];
    p.tau = ws + pl.off_tau[l];
    p.Tg = ws + pl.off_tg[l];
    p.top = (l == L - 1);
    p.expo_acc = (p.top && !(l == 0 && pu.Rm)) ? pu.expo_acc : nullptr;
    p.stagger_kc = g_qr_stagger;
    if (p.top) { p.Rout = R; p.ldr = ldr; p.strideR = strideR; }
    else { p.Rout = ws + pl.off_x[l + 1]; p.ldr = n; p.strideR = pl.m[l + 1] * n; }
    const bool pushed = (l == 0 && pu.Rm);
    // the packing decision ahead of the launch: exactly when the kernel's `packed` path can be taken at all (8-wave PAIR blocks, whole
    // shapes: every mode index valid, all 16 NT columns, Rin a multiple of 16; a 64-row R; an even number of blocks)
    // (from 256 items on: below, the absorbed blocks of a launch run beside its working blocks -- no tail to save -- and the extra
    // launch costs a latency-bound small-batch sweep 5 us per core)
    p.pack_pre = (pushed && g_qr_pack_pre && batch >= g_qr_pack_pre_min && p.pack_ok && p.pack_flag && g_rank_skip_c > 0 && pl.nw[l] == 8 && g_qr_variant != 0 &&
                  pu.k == 64 && pu.Rin <= 64 && (pu.Rin & 15) == 0 && n == PW * NT && (pl.nb[l] & 1) == 0 && pu.I == 8 * pl.nb[l] &&
                  pu.ldrm >= pu.Rin) ? 1 : 0;
    {
      ProfScope prof(TTR_PROF_QR_FACTOR, stream);
      if (p.pack_pre)
        hipLaunchKernelGGL((pack_flags_kernel<T>), dim3((unsigned)batch), dim3(512), 0, stream, (const T*)pu.Rm, pu.ldrm,
                           pu.strideRm, pu.Rin, batch, g_rank_skip_c, p.pack_ok, p.pack_flag);
      p.grid_swap = (p.pack_ok == 3 && l == 0 && pu.Rm) ? ((g_qr_interleave && (pl.nb[l] & 1) == 0) ? 2 : 1) : 0;
      const dim3 grid = p.grid_swap ? dim3((unsigned)batch, (unsigned)pl.nb[l]) : dim3((unsigned)pl.nb[l], (unsigned)batch);
      if (pl.nw[l] == 8 && g_qr_variant != 0) {
        if (pushed) hipLaunchKernelGGL((qr_factor_kernel<T, NT, true, 8, true>), grid, dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((qr_factor_kernel<T, NT, false, 8, true>), grid, dim3(512), 0, stream, p);
      } else if (pl.nw[l] == 8) {
        if (pushed) hipLaunchKernelGGL((qr_factor_kernel<T, NT, true, 8, false>), grid, dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((qr_factor_kernel<T, NT, false, 8, false>), grid, dim3(512), 0, stream, p);
      } else {
        if (pushed) hipLaunchKernelGGL((qr_factor_kernel<T, NT, true, 4, false>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((qr_factor_kernel<T, NT, false, 4, false>), grid, dim3(256), 0, stream, p);
      }
    }
    if (work_census_on()) {   // what this launch executed, from the taus / flags it left behind (outside the timed scope)
      work_qr_taus(TTR_PROF_QR_FACTOR, p.tau, sizeof(T) == 8, batch * pl.nb[l], pl.nb[l], pl.npad, pl.m[l], 64 * pl.nw[l], n, 0, !pushed, stream);
      if (pushed) {
        // the fused push Rm (k x Rin) x core (Rin x I n): 16 x 16 (row tile, K group) products; an upper-triangular Rm skips the
        // tiles below the diagonal (square Rm: T (T + 1) / 2 of T^2); a packed item only forms rows 0 .. 31 (7 of its 8 tiles)
        const double full = 2.0 * pu.k * pu.Rin * (double)pu.I * n;
        const int Tt = (pu.k + 15) / 16;
        const double tri = (pu.k == pu.Rin && Tt >= 1) ? (double)(Tt + 1) / (2.0 * Tt) : 1.0;
        const double fl[4] = {full * tri, 2.0 * 32 * pu.Rin * (double)pu.I * n * (7.0 / 8.0), 0.0, 0.0};
        const double rd = (double)sizeof(T) * ((double)pu.Rin * pu.I * n + (double)pu.k * pu.Rin);   // the core and Rm, read once
        const double by[4] = {rd, rd, 0.0, 0.0};
        work_items(TTR_PROF_QR_FACTOR, p.pack_flag, nullptr, batch, fl, by, stream);
      }
    }
  }
  if (pu.expo_acc && pu.Rm && L == 1) {   // (the pushed kernel was its own top level)
    if constexpr (sizeof(T) == 4) {
      const int64_t mt = (int64_t)pu.k * pu.I;
      ProfScope prof(TTR_PROF_QR_FACTOR, stream);
      hipLaunchKernelGGL(r_expo_kernel<T>, dim3((unsigned)batch), dim3(256), 0, stream, R, (int)(mt < n ? mt : n), n, ldr, strideR,
                         pu.expo_acc);
    }
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

template <typename T, int NT, int NTC>
static int apply_run(int64_t m, int n, int64_t batch, const T* ws, T* wsw, const QrPlan& pl, const T* C, int64_t ldc,
                     int64_t strideC, int kc, T* Out, int64_t ldo, int64_t strideO, int pk, int pI, T* Gp, hipStream_t stream,
                     int64_t o_cs = 1, int skipz = 0) {
  const int L = pl.levels;
  for (int l = L - 1; l >= 0; --l) {
    QrApply<T> p;
    p.ocs = l == 0 ? o_cs : 1;
    p.Vt = ws + pl.off_vt[l];
    p.tau = ws + pl.off_tau[l];
    p.Tg = ws + pl.off_tg[l];
    p.m = pl.m[l]; p.n = n; p.nb = pl.nb[l]; p.kcols = kc;
    if (l == L - 1) { p.Top = C; p.ldtop = ldc; p.strideTop = strideC; }  // C == nullptr: identity
    else { p.Top = ws + pl.off_out[l + 1]; p.ldtop = kc; p.strideTop = pl.m[l + 1] * n; }
    if (l == 0) { p.Out = Out; p.ldout = ldo; p.strideOut = strideO; }
    else { p.Out = wsw + pl.off_out[l]; p.ldout = kc; p.strideOut = pl.m[l] * n; }
    p.pk = (l == 0) ? pk : 0; p.pI = (l == 0) ? pI : 0;
    p.Gp = (l == 0) ? Gp : nullptr;
    p.pack_flag = (l == 0 && pk > 0) ? reinterpret_cast<const int32_t*>(ws + pl.off_flag) : nullptr;
    p.half_zero = (l == 1 && L == 2 && pk > 0 && g_qr_pack == 3 && g_qr_l1_idle && pl.nb[0] == 8 && pl.nw[1] == 8 && n == 64 && pl.m[1] == 512)
                      ? reinterpret_cast<const int32_t*>(ws + pl.off_flag) : nullptr;
    p.skip_zero_rows = (l == 0) ? skipz : 0;
    p.dbg = (l == 0) ? g_qr_dbg : nullptr;
    {
      ProfScope prof(TTR_PROF_QR_APPLY, stream);
      p.grid_swap = (g_qr_pack == 3 && p.pack_flag) ? ((g_qr_interleave && (pl.nb[l] & 1) == 0) ? 2 : 1) : 0;
      const dim3 grid = p.grid_swap ? dim3((unsigned)batch, (unsigned)pl.nb[l]) : dim3((unsigned)pl.nb[l], (unsigned)batch);
      if (pl.nw[l] == 8) hipLaunchKernelGGL((qr_apply_kernel<T, NT, NTC, 8>), grid, dim3(512), 0, stream, p);
      else hipLaunchKernelGGL((qr_apply_kernel<T, NT, NTC, 4>), grid, dim3(256), 0, stream, p);
    }
    if (work_census_on())
      work_qr_taus(TTR_PROF_QR_APPLY, p.tau, sizeof(T) == 8, batch * pl.nb[l], pl.nb[l], pl.npad, pl.m[l], 64 * pl.nw[l], n, kc, false, stream);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

template <typename T>
static int factor_typed(int64_t m, int64_t n, int64_t batch, const void* A, int64_t lda, int64_t strideA, void* R,
                        int64_t ldr, int64_t strideR, void* ws, int64_t ws_bytes, const Pushed& pu, hipStream_t stream,
                        int64_t a_cs = 1) {
  if (batch > kMaxBatchSlice) {  // slices of the batch, one after the other (see qr_workspace_bytes)
    TTR_REQUIRE(ws_bytes >= qr_workspace_bytes(sizeof(T) == 8 ? TTR_F64 : TTR_F32, m, n, batch), TTR_E_WORKSPACE,
                "ttr_qr: workspace too small for batch %lld", (long long)batch);
    char* wsp = (char*)ws;
    for (int64_t b0 = 0; b0 < batch; b0 += kMaxBatchSlice) {
      const int64_t nb = batch - b0 < kMaxBatchSlice ? batch - b0 : kMaxBatchSlice;
      const int64_t wsb = make_plan(m, n, nb, sizeof(T) == 8).total * (int64_t)sizeof(T);
      Pushed ps = pu;
      if (ps.Rm) ps.Rm = (const T*)ps.Rm + b0 * ps.strideRm;
      if (ps.Cn) ps.Cn = (const T*)ps.Cn + b0 * ps.strideCn;
      if (ps.Cn2) ps.Cn2 = (const T*)ps.Cn2 + b0 * ps.strideCn2;
      if (ps.expo_acc) ps.expo_acc += b0;
      const int rc = factor_typed<T>(m, n, nb, A ? (const void*)((const T*)A + b0 * strideA) : nullptr, lda, strideA,
                                     (T*)R + b0 * strideR, ldr, strideR, wsp, wsb, ps, stream, a_cs);
      if (rc != TTR_OK) return rc;
      wsp += wsb;
    }
    return TTR_OK;
  }
  const QrPlan pl = make_plan(m, n, batch, sizeof(T) == 8);
  TTR_REQUIRE(ws_bytes >= pl.total * (int64_t)sizeof(T), TTR_E_WORKSPACE, "ttr_qr: workspace %lld < %lld bytes",
              (long long)ws_bytes, (long long)(pl.total * (int64_t)sizeof(T)));
  switch (nt_for(n)) {
    case 1: return factor_run<T, 1>(m, (int)n, batch, (const T*)A, lda, strideA, (T*)R, ldr, strideR, (T*)ws, pl, pu, stream, a_cs);
    case 2: return factor_run<T, 2>(m, (int)n, batch, (const T*)A, lda, strideA, (T*)R, ldr, strideR, (T*)ws, pl, pu, stream, a_cs);
    default: return factor_run<T, 4>(m, (int)n, batch, (const T*)A, lda, strideA, (T*)R, ldr, strideR, (T*)ws, pl, pu, stream, a_cs);
  }
}

template <typename T, int NT>
static int apply_nt(int64_t m, int n, int64_t batch, T* ws, const QrPlan& pl, const T* C, int64_t ldc, int64_t strideC,
                    int kc, T* Out, int64_t ldo, int64_t strideO, int pk, int pI, T* Gp, hipStream_t stream, int64_t o_cs, int skipz) {
  if (kc <= 16) return apply_run<T, NT, 1>(m, n, batch, ws, ws, pl, C, ldc, strideC, kc, Out, ldo, strideO, pk, pI, Gp, stream, o_cs, skipz);
  if (kc <= 32) return apply_run<T, NT, 2>(m, n, batch, ws, ws, pl, C, ldc, strideC, kc, Out, ldo, strideO, pk, pI, Gp, stream, o_cs, skipz);
  return apply_run<T, NT, 4>(m, n, batch, ws, ws, pl, C, ldc, strideC, kc, Out, ldo, strideO, pk, pI, Gp, stream, o_cs, skipz);
}

template <typename T>
static int apply_typed(int64_t m, int64_t n, int64_t batch, void* ws, int64_t ws_bytes, const void* C, int64_t ldc,
                       int64_t strideC, int64_t kc, void* Out, int64_t ldo, int64_t strideO, int pk, int pI, void* Gp,
                       hipStream_t stream, int64_t o_cs = 1, int skipz = 0) {
  if (batch > kMaxBatchSlice) {
    TTR_REQUIRE(ws_bytes >= qr_workspace_bytes(sizeof(T) == 8 ? TTR_F64 : TTR_F32, m, n, batch), TTR_E_WORKSPACE,
                "ttr_qr_apply: workspace too small for batch %lld", (long long)batch);
    char* wsp = (char*)ws;
    for (int64_t b0 = 0; b0 < batch; b0 += kMaxBatchSlice) {
      const int64_t nb = batch - b0 < kMaxBatchSlice ? batch - b0 : kMaxBatchSlice;
      const int64_t wsb = make_plan(m, n, nb, sizeof(T) == 8).total * (int64_t)sizeof(T);
      const int rc = apply_typed<T>(m, n, nb, wsp, wsb, C ? (const void*)((const T*)C + b0 * strideC) : nullptr, ldc, strideC, kc,
                                    (T*)Out + b0 * strideO, ldo, strideO, pk, pI,
                                    Gp ? (void*)((T*)Gp + b0 * make_plan(m, n, nb, sizeof(T) == 8).nb[0] * (int64_t)(64 * 64)) : nullptr, stream, o_cs, skipz);
      if (rc != TTR_OK) return rc;
      wsp += wsb;
    }
    return TTR_OK;
  }
  const QrPlan pl = make_plan(m, n, batch, sizeof(T) == 8);
  TTR_REQUIRE(ws_bytes >= pl.total * (int64_t)sizeof(T), TTR_E_WORKSPACE, "ttr_qr_apply: workspace %lld < %lld bytes",
              (long long)ws_bytes, (long long)(pl.total * (int64_t)sizeof(T)));
  TTR_REQUIRE(kc >= 1 && kc <= 64 && kc <= n, TTR_E_UNSUPPORTED, "ttr_qr_apply: kcols = %lld outside [1, min(n, 64)]",
              (long long)kc);
  switch (nt_for(n)) {
    case 1: return apply_nt<T, 1>(m, (int)n, batch, (T*)ws, pl, (const T*)C, ldc, strideC, (int)kc, (T*)Out, ldo, strideO, pk, pI, (T*)Gp, stream, o_cs, skipz);
    case 2: return apply_nt<T, 2>(m, (int)n, batch, (T*)ws, pl, (const T*)C, ldc, strideC, (int)kc, (T*)Out, ldo, strideO, pk, pI, (T*)Gp, stream, o_cs, skipz);
    default: return apply_nt<T, 4>(m, (int)n, batch, (T*)ws, pl, (const T*)C, ldc, strideC, (int)kc, (T*)Out, ldo, strideO, pk, pI, (T*)Gp, stream, o_cs, skipz);
  }
}

int qr_max_cols(int) { return 64; }

int qr_factor_dispatch(int dtype, int64_t m, int64_t n, int64_t batch, const void* A, int64_t lda, int64_t strideA,
                       void* R, int64_t ldr, int64_t strideR, void* ws, int64_t ws_bytes, hipStream_t stream, int64_t a_cs,
                       int32_t* expo_acc) {
  TTR_REQUIRE(n <= qr_max_cols(dtype), TTR_E_UNSUPPORTED, "ttr_qr: n = %lld exceeds the %d-column panel kernel",
              (long long)n, qr_max_cols(dtype));
  TTR_REQUIRE(!expo_acc || dtype == TTR_F32, TTR_E_UNSUPPORTED, "ttr_qr_factor: expo_acc is an fp32 facility");
  Pushed none;
  none.expo_acc = expo_acc;
  if (dtype == TTR_F32) return factor_typed<float>(m, n, batch, A, lda, strideA, R, ldr, strideR, ws, ws_bytes, none, stream, a_cs);
  return factor_typed<double>(m, n, batch, A, lda, strideA, R, ldr, strideR, ws, ws_bytes, none, stream, a_cs);
}

int qr_apply_dispatch(int dtype, int64_t m, int64_t n, int64_t batch, void* ws, int64_t ws_bytes, const void* C,
                      int64_t ldc, int64_t strideC, int64_t kc, void* Out, int64_t ldo, int64_t strideO,
                      hipStream_t stream, int64_t o_cs) {
  TTR_REQUIRE(n <= qr_max_cols(dtype), TTR_E_UNSUPPORTED, "ttr_qr_apply: n = %lld exceeds the %d-column panel kernel",
              (long long)n, qr_max_cols(dtype));
  if (dtype == TTR_F32) return apply_typed<float>(m, n, batch, ws, ws_bytes, C, ldc, strideC, kc, Out, ldo, strideO, 0, 0, nullptr, stream, o_cs);
  return apply_typed<double>(m, n, batch, ws, ws_bytes, C, ldc, strideC, kc, Out, ldo, strideO, 0, 0, nullptr, stream, o_cs);
}

// Pushed variants: the factored matrix is the (k*I) x n left unfolding of Rm * C; level 0 has ceil(I/NW)
// zero-padded blocks of 64*NW rows (one mode index per wave), so the plan is that of a (64*NW*ceil(I/NW)) x n
// matrix -- NW = 8 as soon as there are more than four mode indices (consistent with nw_for; fp64: 4-wave blocks, see there).
static int64_t pushed_rows(int64_t I, int dtype) {
  if (nw4_forced(dtype == TTR_F64)) return 256 * ceil_div(I, 4);
  return I > 4 ? 512 * ceil_div(I, 8) : 256;
}

// Byte offset, inside the workspace of a pushed factorisation, of the per-item int32 flags "this item's level-0 blocks packed
// their rows: rows kk >= 32 of everything ttr_qr_apply_pushed produces from it are exactly zero"; -1 when the batch is processed
// in slices (> 65535 items: one flag array per slice).
int64_t qr_pushed_flag_offset(int dtype, int64_t I, int64_t n, int64_t batch) {
  if (batch > kMaxBatchSlice) return -1;
  return make_plan(pushed_rows(I, dtype), n, batch, dtype == TTR_F64).off_flag * (dtype == TTR_F64 ? 8 : 4);
}

int64_t qr_pushed_workspace_bytes(int dtype, int64_t I, int64_t n, int64_t batch) {
  return qr_workspace_bytes(dtype, pushed_rows(I, dtype), n, batch);
}

static int pushed_ok(int dtype, int64_t k, int64_t Rin, int64_t I, int64_t n) {
  TTR_REQUIRE(k >= 1 && k <= 64 && Rin >= 1 && Rin <= 64 && I >= 1 && n >= 1 && n <= qr_max_cols(dtype), TTR_E_UNSUPPORTED,
              "ttr_qr_*_pushed: k, Rin, n must be <= 64 (got %lld, %lld, %lld)", (long long)k, (long long)Rin, (long long)n);
  TTR_REQUIRE(k * I >= n, TTR_E_UNSUPPORTED, "ttr_qr_*_pushed: needs k*I >= n (tall unfolding)");
  TTR_REQUIRE(Rin * I * n < (int64_t(1) << 31), TTR_E_UNSUPPORTED,
              "ttr_qr_*_pushed: core of %lld elements exceeds the kernel's 32-bit element offsets", (long long)(Rin * I * n));
  return TTR_OK;
}

int qr_factor_pushed_dispatch(int dtype, int64_t k, int64_t Rin, int64_t I, int64_t n, int64_t batch, const void* Rm,
                              int64_t ldrm, int64_t strideRm, const void* Cn, int64_t strideCn, void* R, int64_t ldr,
                              int64_t strideR, void* ws, int64_t ws_bytes, hipStream_t stream, int32_t* expo_acc) {
  const int rc = pushed_ok(dtype, k, Rin, I, n);
  if (rc != TTR_OK) return rc;
  TTR_REQUIRE(!expo_acc || dtype == TTR_F32, TTR_E_UNSUPPORTED, "ttr_qr_factor_pushed: expo_acc is an fp32 facility");
  Pushed pu;
  pu.Rm = Rm; pu.ldrm = ldrm; pu.strideRm = strideRm; pu.Cn = Cn; pu.strideCn = strideCn;
  pu.k = (int)k; pu.Rin = (int)Rin; pu.I = (int)I;
  pu.expo_acc = expo_acc;
  const int64_t m = pushed_rows(I, dtype);
  if (dtype == TTR_F32) return factor_typed<float>(m, n, batch, nullptr, 0, 0, R, ldr, strideR, ws, ws_bytes, pu, stream);
  return factor_typed<double>(m, n, batch, nullptr, 0, 0, R, ldr, strideR, ws, ws_bytes, pu, stream);
}

// Fused push of a block-diagonal core: QR of the left unfolding of Rm * blockdiag(a, b) (a: [ra][I][ca], b: [rb][I][cb]).
int qr_factor_pushed_sum_dispatch(int dtype, int64_t k, int64_t I, int64_t batch, const void* Rm, int64_t ldrm,
                                  int64_t strideRm, const void* Ca, int64_t ra, int64_t ca, int64_t strideCa,
                                  const void* Cb, int64_t rb, int64_t cb, int64_t strideCb, void* R, int64_t ldr,
                                  int64_t strideR, void* ws, int64_t ws_bytes, hipStream_t stream) {
  const int64_t n = ca + cb, Rin = ra + rb;
  const int rc = pushed_ok(dtype, k, Rin, I, n);
  if (rc != TTR_OK) return rc;
  TTR_REQUIRE(ra >= 1 && rb >= 1 && ca >= 1 && cb >= 1, TTR_E_INVALID, "ttr_qr_factor_pushed_sum: empty block");
  Pushed pu;
  pu.Rm = Rm; pu.ldrm = ldrm; pu.strideRm = strideRm; pu.Cn = Ca; pu.strideCn = strideCa;
  pu.Cn2 = Cb; pu.strideCn2 = strideCb; pu.sumRa = (int)ra; pu.sumCa = (int)ca;
  pu.k = (int)k; pu.Rin = (int)Rin; pu.I = (int)I;
  const int64_t m = pushed_rows(I, dtype);
  if (dtype == TTR_F32) return factor_typed<float>(m, n, batch, nullptr, 0, 0, R, ldr, strideR, ws, ws_bytes, pu, stream);
  return factor_typed<double>(m, n, batch, nullptr, 0, 0, R, ldr, strideR, ws, ws_bytes, pu, stream);
}

// Number of row-Gram partials ttr_qr_apply_pushed_gram writes per batch item (0: the fused Gram epilogue does not cover this shape)
int64_t qr_apply_pushed_gram_parts(int dtype, int64_t k, int64_t I, int64_t n, int64_t kc) {
  if (dtype != TTR_F32 || k != 64 || kc != 32 || n < 32 || n > 64 || I < 8 || (I % 8) != 0) return 0;
  return I / 8;  // one 512-row level-0 block per 8 mode indices
}

int qr_apply_pushed_dispatch(int dtype, int64_t k, int64_t I, int64_t n, int64_t batch, void* ws, int64_t ws_bytes,
                             const void* C, int64_t ldc, int64_t strideC, int64_t kc, void* Out, int64_t ldo,
                             int64_t strideO, void* G, hipStream_t stream, int skip_zero_rows) {
  const int rc = pushed_ok(dtype, k, 1, I, n);
  if (rc != TTR_OK) return rc;
  TTR_REQUIRE(!G || (qr_apply_pushed_gram_parts(dtype, k, I, n, kc) > 0 && ldo == kc), TTR_E_UNSUPPORTED,
              "ttr_qr_apply_pushed_gram: shape not covered by the fused Gram epilogue (see ttr_qr_apply_pushed_gram_parts)");
  // The Gram epilogue walks the UNPACKED row map of the level-0 blocks: a factorisation whose items may have packed their rows
  // (TTR_KNOB_QR_PACK != 0 when ttr_qr_factor_pushed ran) cannot feed it -- an absorbed block would leave its partial unwritten.
  // The knob has to be 0 from before the factorisation until after this call (tntorch_amd/_hipops.py: TTR_FUSE_APPLY_GRAM=1).
  TTR_REQUIRE(!G || g_qr_pack == 0, TTR_E_UNSUPPORTED,
              "ttr_qr_apply_pushed_gram: row packing is enabled (TTR_KNOB_QR_PACK = %d); the fused Gram epilogue needs it off "
              "for the factorisation AND the apply", g_qr_pack);
  const int64_t m = pushed_rows(I, dtype);
  if (dtype == TTR_F32)
    return apply_typed<float>(m, n, batch, ws, ws_bytes, C, ldc, strideC, kc, Out, ldo, strideO, (int)k, (int)I, G, stream, 1, skip_zero_rows);
  return apply_typed<double>(m, n, batch, ws, ws_bytes, C, ldc, strideC, kc, Out, ldo, strideO, (int)k, (int)I, nullptr, stream, 1, skip_zero_rows);
}

}  // namespace ttr
