
// Batched symmetric eigensolver + truncated_svd's rank rule, on device (gfx950).
//
// One workgroup per matrix runs a parallel-order cyclic two-sided Jacobi iteration on
// G (n x n, symmetric PSD Gram matrix) with the eigenvector matrix V accumulated
// alongside; both live in LDS (leading dimension n+1: row and column walks are both
// bank-conflict free) when they fit, otherwise in a caller workspace that stays
// L2-resident.  A round rotates n/2 disjoint (p,q) pairs (round-robin tournament):
//   phase 1  one thread per pair computes (c, s) from G_pp, G_qq, G_pq
//   phase 2  G <- J^T G J block-wise: one thread owns the 2x2 block (row pair, column pair) and applies
//            both rotations in place; V <- V J column-wise in the same barrier interval
// (two barriers per round)
// ~500 rotations touch every entry of the accumulated eigenvector matrix V, and float accumulation
// leaves V^T V = I + O(3e-6), which shows up one-to-one as reconstruction error of the truncation
// (measured 3.5e-6 per bond vs 3e-7 for LAPACK).  V is nevertheless kept in the matrix precision in
// LDS (half the footprint: FOUR fp32 64x64 problems per CU instead of three, and the kernel is
// occupancy/latency bound) and orthogonality is restored once at the end by one Newton-Schulz step
// V <- V (I - E/2), E = V^T V - I, with E accumulated in double: V^T V = I + O(E^2) + O(eps).
// Rotations are skipped when |G_pq| <= eps * sqrt(G_pp G_qq) (the relative criterion
// that gives Jacobi its high relative accuracy on graded Gram matrices); the sweep
// loop ends when a whole sweep rotates nothing.
//
// Epilogue = round.py:118-158 of the reference, fused: clamp, sqrt, sort by
// decreasing sigma (counting sort, one thread per eigenvalue), permute V's columns,
// tail-energy scan (accumulated in double like torch.cumsum on CPU), rank selection.
#include <type_traits>

#include "ttr_common.h"

namespace ttr {

template <typename T>
struct EighArgs {
  int n;
  const T* G;
  int64_t ldg, strideG;
  int gparts;            // G[b] = sum of `gparts` partial matrices spaced stride_gpart apart (split-K partials of a Gram kernel)
  int64_t stride_gpart;
  T* V;
  int64_t ldv, strideV;
  T* sigma;
  int64_t stride_sigma;
  int32_t* info;
  int eig_mode;
  int use_delta;
  double delta2;
  int64_t rmax;
  T* ws;  // global-memory variant: per matrix 2 * n * (n + 1) elements (V, then G)
  int max_sweeps;
  int abs_floor;     // 1: also skip rotations with |G_pq| <= tol * max|G_ii| (plain Gram input: its entries are
                     //    only accurate to eps*||G||, below that level rotations chase rounding noise forever)
  int32_t* sweeps;   // optional [batch]: sweeps used (diagnostics / convergence tests)
  int top_pre;       // eigh_tridiag_kernel<T, *, 64>: the 32-row launch ran before this one (info[b] != -1: item done)
  // rank rule with the bound on the DEVICE (one double per launch; overrides delta2): eps-mode sweeps enqueue every bond
  // without reading the norm back
  const double* delta2_dev;
  int noise_c;           // TTR_KNOB_RANK_NOISE_FLOOR (ttr_common.h: rank_rule)
  // block-Jacobi pair problems (ttr_bj_solve): grid = pairs_per_item * items; problem (item, pair) is the 2b x 2b matrix
  // [[G_ii, G_ij], [G_ji, G_jj]] of blocks i = pair_tab[2 pair], j = pair_tab[2 pair + 1] of the item's n x n matrix
  const int32_t* pair_tab;
  int pair_b, pairs_per_item;
  // pass 2 of the two-pass truncation, items whose kept spectrum is flat (ttr_spectrum_flat): the first Gram matrix already
  // carries every kept singular value to a few eps, so the item passes through -- V = I, sigma = sigma_in (pass 1's, sorted),
  // rank rule as usual -- and its rotated Gram matrix (which ttr_rotgram did not write) is never read
  const int32_t* skip_items;
  const T* sigma_in;
  int64_t stride_sigma_in;
  // eigh_tridiag_kernel<T, true>: top_r largest eigenpairs by the top-r path when the kept spectrum is flat within top_thr and has
  // no close pair (top_flat[b] = 1, optional), the QL phase otherwise (top_flat[b] = 0)
  int32_t* top_flat;
  double top_thr;
  int top_r;
  int top_need_all;          // != 0: the top-r path only serves items of which it computes EVERY eigenpair (top_r >= the live size)
  const int32_t* skip_flag;  // != 0 on the device: the driver has converged, the launch returns at once
  int32_t* rot_count;        // incremented once per problem that rotated anything (the driver's "a whole sweep found nothing")
};

constexpr int kMaxPairs = 2048;  // n <= 4096 (fp32) / 2048 (fp64) in the global-memory variant: its rotation table and
                                // sort scratch still have to fit the 160 KB LDS

template <typename T>
__host__ __device__ constexpr size_t rot_table_bytes(int npad) {
  return (size_t)npad * (2 * sizeof(double) + 4 * sizeof(int) + 2 * sizeof(T));
}

// Jacobi rotation that annihilates G_pq.  Computed in the matrix precision with the fast hardware
// reciprocal / rsqrt (the angle only has to be accurate enough for convergence), then (c, s) is pulled back
// onto the unit circle in double with one Newton step so that the accumulated V stays orthogonal.
__device__ __forceinline__ void jacobi_cs(float app, float aqq, float apq, double& c, double& s) {
  const float theta = (aqq - app) * __builtin_amdgcn_rcpf(2.0f * apq);
  float t;
  if (fabsf(theta) > 1e18f) t = 0.5f * __builtin_amdgcn_rcpf(theta);
  else t = copysignf(1.0f, theta) * __builtin_amdgcn_rcpf(fabsf(theta) + __builtin_amdgcn_sqrtf(1.0f + theta * theta));
  const float cf = __builtin_amdgcn_rsqf(1.0f + t * t);
  const double cd = (double)cf, sd = (double)(t * cf);
  const double corr = 1.5 - 0.5 * (cd * cd + sd * sd);
  c = cd * corr;
  s = sd * corr;
}
__device__ __forceinline__ void jacobi_cs(double app, double aqq, double apq, double& c, double& s) {
  const double theta = (aqq - app) / (2.0 * apq);
  double t;
  if (fabs(theta) > 1e150) t = 0.5 / theta;
  else t = copysign(1.0, theta) / (fabs(theta) + sqrt(1.0 + theta * theta));
  c = 1.0 / sqrt(1.0 + t * t);
  s = t * c;
}
constexpr int UNR = 8;          // rotation items batched per thread (n = 64: exactly one batch per phase)

template <typename T, bool LDSRES, int NTH = kThreads>
__global__ __launch_bounds__(NTH) void eigh_jacobi_kernel(EighArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (p.skip_flag && *p.skip_flag != 0) return;  // (wave-uniform: one word for the whole launch)
  const int tid = threadIdx.x;
  const int64_t bt = blockIdx.x;
  const int nf = p.n;
  if (p.skip_items && p.skip_items[bt] != 0) {  // (block-uniform) pass-through item, see EighArgs
    const int n = nf;
    T* __restrict__ Vo = p.V + bt * p.strideV;
    for (int idx = tid; idx < n * n; idx += NTH) {
      const int i = idx / n, j = idx - i * n;
      Vo[(int64_t)i * p.ldv + j] = (i == j) ? T(1) : T(0);
    }
    const T* __restrict__ si = p.sigma_in + bt * p.stride_sigma_in;
    for (int i = tid; i < n; i += NTH) p.sigma[bt * p.stride_sigma + i] = si[i];
    if (tid == 0) {  // the rank rule of the regular epilogue, on pass 1's sigma (ttr_spectrum_flat made sure pass 2 would agree)
      const T d2 = p.use_delta ? (T)(p.delta2_dev ? *p.delta2_dev : p.delta2) : T(0);
      p.info[bt] = rank_rule<T>(si, n, n, p.rmax, p.use_delta, d2, p.noise_c);
      if (p.sweeps) p.sweeps[bt] = 0;
    }
    return;
  }
  // TTR_SOLVER_JACOBI_LIVE, n <= 64 (round 4): the solve is restricted to the LIVE PREFIX.  Indices with G_ii <= (n eps)^2 max G_ii
  // are frozen (below): they never rotate, their eigenvector stays e_i and nothing that is read afterwards depends on their
  // rows / columns -- so when every index from nl on is frozen (pass 1 sorts: the frozen ones are the tail), working on the
  // leading nl x nl block gives the same result with (nl / n)^3 of the work.  A bond whose carry has zero rows 32.. (packed QR)
  // has nl <= 32; a decaying bond sigma_j ~ 2^(-j/2) has nl = 34 of 64.  Decided from the diagonal (read from global memory by
  // every wave, lane = index), before the LDS is carved; the tail's diagonal entries stay in registers for the epilogue's sort.
  int nl = nf;
  T dtail = T(0);
  if (LDSRES && p.abs_floor == TTR_SOLVER_JACOBI_LIVE && !p.pair_tab && nf <= 64 && nf > 2) {
    const int l = tid & 63;
    if (l < nf) {
      const T* __restrict__ Gd = p.G + bt * p.strideG + (int64_t)l * p.ldg + l;
      dtail = Gd[0];
      for (int pt = 1; pt < p.gparts; ++pt) dtail += Gd[pt * p.stride_gpart];   // (the summation order of the matrix load below)
    }
    T gmx = l < nf ? fabs(dtail) : T(0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmx = fmax(gmx, __shfl_xor(gmx, off, 64));
    const T thr0 = (T)nf * Num<T>::eps();
    const unsigned long long lv = __ballot(l < nf && !(dtail <= thr0 * thr0 * gmx));
    nl = lv ? 64 - __clzll(lv) : 1;
    if (nl < 2) nl = 2;
  }
  const int n = nl;
  const int ld = n + 1;
  const int ne = n + (n & 1);  // even number of players (phantom index n if n is odd)
  const int np = ne / 2;

  // carve the dynamic LDS: rotation table, sigma scratch, then G / V.
  const int npad = (np + 7) & ~7;
  unsigned char* tab = smem_raw;  // one table shared by the workgroup
  double* cs_c = reinterpret_cast<double*>(tab);           // [npad]
  double* cs_s = cs_c + npad;
  int* pq_p = reinterpret_cast<int*>(cs_s + npad);
  int* pq_q = pq_p + npad;
  int* prow = pq_q + npad;                                 // p * ld, q * ld (row offsets of the pair)
  int* qrow = prow + npad;
  T* ct = reinterpret_cast<T*>(qrow + npad);               // (c, s) once more in the matrix type
  T* st = ct + npad;
  unsigned char* after = smem_raw + rot_table_bytes<T>(npad);
  int* flags = reinterpret_cast<int*>(after);              // [0]: rotated-this-sweep
  int* deadv = flags + 16;                                 // [n4] 1 = numerically null index, never rotated (TTR_SOLVER_JACOBI_LIVE)
  const int n4 = (n + 4) & ~3;
  T* sg = reinterpret_cast<T*>(deadv + n4);                // sigma / ordering scratch [2 * n]
  T* Gs;
  T* Vs;
  if (LDSRES) {
    Vs = sg + 2 * ((nf + 1) & ~1) + 2;
    Gs = Vs + (size_t)n * ld;
  } else {
    Vs = p.ws + bt * (int64_t)2 * n * ld;
    Gs = Vs + (size_t)n * ld;
  }

  if (p.pair_tab) {
    // pair problem of the block-Jacobi driver: gather the four b x b blocks of the pair (the driver never moves blocks)
    const int64_t item = bt / p.pairs_per_item;
    const int pair = (int)(bt - item * p.pairs_per_item);
    const int b = p.pair_b;
    const int64_t ri = (int64_t)p.pair_tab[2 * pair] * b, rj = (int64_t)p.pair_tab[2 * pair + 1] * b;
    const T* __restrict__ G = p.G + item * p.strideG;
    for (int idx = tid; idx < n * n; idx += NTH) {
      const int i = idx / n, j = idx % n;
      const int64_t gi = i < b ? ri + i : rj + (i - b), gj = j < b ? ri + j : rj + (j - b);
      Gs[i * ld + j] = G[gi * p.ldg + gj];
      Vs[i * ld + j] = (i == j) ? T(1) : T(0);
    }
  } else {
  const T* __restrict__ G = p.G + bt * p.strideG;
  for (int idx = tid; idx < n * n; idx += NTH) {
    const int i = idx / n, j = idx % n;
    T gv = G[(int64_t)i * p.ldg + j];
    for (int pt = 1; pt < p.gparts; ++pt) gv += G[pt * p.stride_gpart + (int64_t)i * p.ldg + j];
    Gs[i * ld + j] = gv;
    Vs[i * ld + j] = (i == j) ? T(1) : T(0);
  }
  }
  if (tid == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0; }
  __syncthreads();
  T floor_abs = Num<T>::tiny();
  if (p.abs_floor) {
    T gmax = 0;
    for (int i = 0; i < n; ++i) gmax = fmax(gmax, fabs(Gs[i * ld + i]));  // broadcast reads
    if (p.abs_floor == TTR_SOLVER_JACOBI_LIVE) {
      // Pass 2 of the 'svd' truncation: G is the Gram matrix of ROTATED rows, i.e. graded and nearly diagonal, and
      // every entry is accurate relative to sqrt(G_pp G_qq).  The purely relative (cosine) rotation test then gives
      // every live direction the accuracy class of a backward-stable SVD, however small its sigma.  Indices whose
      // G_ii <= (n eps)^2 max G_ii are the numerical null space of the input (their rows are rounding noise of the
      // pass-1 rotation): rotating them never terminates (the noise is regenerated by every update) and changes
      // nothing above the noise level, so they are frozen.
      const T thr = (T)nf * Num<T>::eps();   // (nf: the thresholds do not depend on the live prefix)
      const T dead_below = thr * thr * gmax;
      for (int i = tid; i < n; i += NTH) deadv[i] = (Gs[i * ld + i] <= dead_below) ? 1 : 0;
      __syncthreads();
    } else {
      floor_abs = fmax(floor_abs, Num<T>::eps() * sqrt((T)n) * gmax);
    }
  }
  const bool live_mode = p.abs_floor == TTR_SOLVER_JACOBI_LIVE;

  // LAPACK xGESVJ-style tolerance sqrt(n)*eps: with a bare eps the rounding noise of the updates keeps
  // regenerating off-diagonals at the eps level and the sweep loop never terminates.
  const T eps = Num<T>::eps() * sqrt((T)nf);
  const int m1 = ne - 1;
  const int k0 = tid / n, i0 = tid % n;            // item = tid + 256*e  <->  (k, i), advanced incrementally
  const int dk = NTH / n, di = NTH % n;
  // 2x2-block phase mapping: sub-groups of cw lanes <-> column pairs, (wave, sub-group) <-> row pairs
  const int cw = np <= 32 ? 32 : 64;
  const int kc_l = (tid & 63) & (cw - 1);
  const int krs = (NTH / 64) * (64 / cw);                       // row-pair stride between a thread's blocks
  const int kr0 = (tid >> 6) * (64 / cw) + (tid & 63) / cw;

  // Convergence scan, before every sweep: does any off-diagonal entry pass the rotation test?  One pass over the
  // matrix and one barrier -- instead of a whole rotation-free sweep (63 rounds, 126 barriers) to find out.  Pass 2
  // of the 'svd' algorithm hands over an already diagonal-to-working-accuracy matrix: then no sweep runs at all, V
  // stays the identity and the re-orthogonalisation below is skipped.
  auto needs_work = [&]() {
    // exactly the test of phase 1, on exactly the element phase 1 reads (G[p][q] of the schedule's ordered pair:
    // the two-sided updates leave G symmetric only to rounding, so testing the other triangle could disagree
    // with phase 1 at the threshold and keep a finished matrix spinning until max_sweeps)
    for (int idx = tid; idx < m1 * np; idx += NTH) {
      const int r = idx / np, k = idx - r * np;
      int pp, qq;
      if (k == 0) { pp = ne - 1; qq = r % m1; }
      else { pp = (r + k) % m1; qq = (r - k + m1) % m1; }
      if (pp < n && qq < n) {
        const T aabs = fabs(Gs[pp * ld + qq]);
        if (aabs > eps * (sqrt(fabs(Gs[pp * ld + pp])) * sqrt(fabs(Gs[qq * ld + qq]))) && aabs > floor_abs &&
            !(live_mode && (deadv[pp] | deadv[qq])))
          flags[1] = 1;
      }
    }
    __syncthreads();
    const bool w = flags[1] != 0;
    __syncthreads();
    if (tid == 0) flags[1] = 0;
    return w;  // (the next write to flags[1] happens after at least one more barrier)
  };
  bool any_work = false;
  int sweeps_used = 0;
  for (int sweep = 0; sweep < p.max_sweeps && n > 1; ++sweep) {
    if (!needs_work()) break;
    any_work = true;
    sweeps_used = sweep + 1;
    for (int r = 0; r < m1; ++r) {
      // ---- phase 1: rotations of this round (one thread per pair)
      for (int k = tid; k < np; k += NTH) {
        int pp, qq;
        if (k == 0) { pp = ne - 1; qq = r % m1; }
        else { pp = (r + k) % m1; qq = (r - k + m1) % m1; }
        double c = 1.0, s = 0.0;
        if (pp < n && qq < n) {
          const T app = Gs[pp * ld + pp], aqq = Gs[qq * ld + qq], apq = Gs[pp * ld + qq];
          const T aabs = fabs(apq);
          if (aabs > eps * (sqrt(fabs(app)) * sqrt(fabs(aqq))) && aabs > floor_abs &&  // no overflow of app*aqq
              !(live_mode && (deadv[pp] | deadv[qq]))) {
            jacobi_cs(app, aqq, apq, c, s);
            flags[2 + (r & 1)] = 1;  // this round has work
          }
        }
        // n odd: the player paired with the phantom index sits this round out.  Its pair is stored as
        // (real, real) with the identity rotation: the 2x2-block update then still applies the COLUMN
        // rotations to that row (both block rows alias the same row and receive the same value).
        if (pp >= n) pp = qq;
        if (qq >= n) qq = pp;
        cs_c[k] = c; cs_s[k] = s; ct[k] = (T)c; st[k] = (T)s; pq_p[k] = pp; pq_q[k] = qq;
        prow[k] = pp * ld; qrow[k] = qq * ld;
      }
      __syncthreads();
      // Rounds in which no pair passed the rotation test (the common case once the matrix is nearly diagonal:
      // pass 2 of the 'svd' algorithm, late block-Jacobi sweeps) skip the two update passes altogether.
      const bool round_work = flags[2 + (r & 1)] != 0;
      if (tid == 0) flags[2 + ((r + 1) & 1)] = 0;  // slot of the NEXT round; nobody touches it before the barrier below
      if (round_work) {
      // ---- phase 2: G <- J^T G J on 2x2 blocks.  Thread item = (row pair kr, column pair kc): the block
      // G[{p,q}][{p',q'}] is read, rotated from both sides and written back by ONE thread, so the two-sided
      // update is a single in-place pass (4 LDS reads, 16 flops, 4 writes) instead of a column pass, a
      // barrier and a row pass.  Lanes walk the column pairs.
      // Lane sub-groups of `cw` lanes walk the column pairs (their rotation is loaded once per round),
      // the sub-groups / waves / iterations walk the row pairs, four blocks in flight per thread.
      for (int kc = kc_l; kc < np; kc += cw) {
        const T cc = ct[kc], sc = st[kc];
        const int pc = pq_p[kc], qc = pq_q[kc];
        for (int krb = kr0; krb < np; krb += 4 * krs) {
          if (krb + 3 * krs < np) {
            // fast path, branch-free: an identity rotation (c = 1, s = 0) reproduces its inputs exactly, so
            // nothing is skipped and the 16 LDS loads of the four blocks are in flight together
            T cr[4], sr[4], a[4], bq[4], c2[4], d[4];
            int pr[4], qr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int kk = krb + u * krs;
              cr[u] = ct[kk]; sr[u] = st[kk]; pr[u] = prow[kk]; qr[u] = qrow[kk];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              a[u] = Gs[pr[u] + pc]; bq[u] = Gs[pr[u] + qc]; c2[u] = Gs[qr[u] + pc]; d[u] = Gs[qr[u] + qc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const T t1 = cc * a[u] - sc * bq[u], t2 = sc * a[u] + cc * bq[u];
              const T u1 = cc * c2[u] - sc * d[u], u2 = sc * c2[u] + cc * d[u];
              a[u] = cr[u] * t1 - sr[u] * u1; bq[u] = cr[u] * t2 - sr[u] * u2;
              c2[u] = sr[u] * t1 + cr[u] * u1; d[u] = sr[u] * t2 + cr[u] * u2;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              Gs[pr[u] + pc] = a[u]; Gs[pr[u] + qc] = bq[u]; Gs[qr[u] + pc] = c2[u]; Gs[qr[u] + qc] = d[u];
            }
          } else {
            for (int kk = krb; kk < np; kk += krs) {
              const T cr = ct[kk], sr = st[kk];
              const int pr = prow[kk], qr = qrow[kk];
              const T a = Gs[pr + pc], bq = Gs[pr + qc], c2 = Gs[qr + pc], d = Gs[qr + qc];
              const T t1 = cc * a - sc * bq, t2 = sc * a + cc * bq;
              const T u1 = cc * c2 - sc * d, u2 = sc * c2 + cc * d;
              Gs[pr + pc] = cr * t1 - sr * u1; Gs[pr + qc] = cr * t2 - sr * u2;
              Gs[qr + pc] = sr * t1 + cr * u1; Gs[qr + qc] = sr * t2 + cr * u2;
            }
          }
        }
      }
      // ---- V <- V J  (columns p,q; lanes walk rows).  Independent of the G blocks: same barrier interval.
      for (int base = tid, kb_ = k0, ib_ = i0; base < np * n;) {
        if (base + (UNR - 1) * NTH < np * n) {  // fast path, branch-free (same argument as above)
          T cv[UNR], sv[UNR], vp[UNR], vq[UNR];
          int ap[UNR], aq[UNR];
          int k = kb_, i = ib_;
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            sv[u] = st[k]; cv[u] = ct[k]; ap[u] = i * ld + pq_p[k]; aq[u] = i * ld + pq_q[k];
            i += di; k += dk;
            if (i >= n) { i -= n; ++k; }
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u) { vp[u] = Vs[ap[u]]; vq[u] = Vs[aq[u]]; }
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            Vs[ap[u]] = cv[u] * vp[u] - sv[u] * vq[u];
            Vs[aq[u]] = sv[u] * vp[u] + cv[u] * vq[u];
          }
          base += UNR * NTH; kb_ = k; ib_ = i;
        } else {
          int k = kb_, i = ib_;
          for (; base < np * n; base += NTH) {
            const T sv = st[k], cv = ct[k];
            const int ap = i * ld + pq_p[k], aq = i * ld + pq_q[k];
            const T vp = Vs[ap], vq = Vs[aq];
            Vs[ap] = cv * vp - sv * vq;
            Vs[aq] = sv * vp + cv * vq;
            i += di; k += dk;
            if (i >= n) { i -= n; ++k; }
          }
        }
      }
      }  // round_work
      __syncthreads();
    }
  }

  // ---- epilogue: clamp / sqrt / sort, Newton-Schulz re-orthogonalisation, permuted write, rank rule
  // (over all nf indices: the frozen tail beyond the live prefix takes part in the sort with its untouched diagonal entries)
  T* sig = sg;             // [nf] unsorted sigma
  T* sig_sorted = sg + nf; // [nf]
  int* posv = reinterpret_cast<int*>(smem_raw);  // [nf] destination column of eigenvector i (the rotation table is free now:
                                                 // 40 bytes per pair slot, at least 8 slots >= 64 ints when nf <= 64 > n)
  for (int i = tid; i < nf; i += NTH) {
    T w = i < n ? Gs[i * ld + i] : dtail;        // (n < nf only when nf <= 64: tid = i = this lane's index)
    if (p.eig_mode == TTR_EIG_REF) { if (w < T(0)) w = T(1e-8); }
    else { if (!(w > T(0))) w = T(0); }
    sig[i] = sqrt(w);
  }
  __syncthreads();
  for (int i = tid; i < nf; i += NTH) {
    const T si = sig[i];
    int pos = 0;
    for (int j = 0; j < nf; ++j) {
      const T sj = sig[j];
      pos += (sj > si) || (sj == si && j < i);
    }
    sig_sorted[pos] = si;
    // TTR_EIG_MATCH_DIAG: Jacobi rotations (|angle| <= pi/4) never swap, eigenpair i stays in column i
    posv[i] = p.eig_mode == TTR_EIG_MATCH_DIAG ? i : pos;
  }
  __syncthreads();
  T* __restrict__ V = p.V + bt * p.strideV;
  if (sizeof(T) == 4 && any_work) {
    // E = V^T V - I (double accumulation) overwrites G, which is no longer needed
    for (int idx = tid; idx < n * n; idx += NTH) {
      const int i = idx / n, j = idx - i * n;
      double a = 0.0;
      for (int k = 0; k < n; ++k) a += (double)Vs[k * ld + i] * (double)Vs[k * ld + j];
      Gs[i * ld + j] = (T)(a - (i == j ? 1.0 : 0.0));
    }
    __syncthreads();
    // V' = V - 0.5 V E, written straight to the output with the columns permuted into sorted order
    for (int idx = tid; idx < n * n; idx += NTH) {
      const int row = idx / n, j = idx - row * n;
      T a = 0;
      for (int k = 0; k < n; ++k) a += Vs[row * ld + k] * Gs[k * ld + j];
      V[(int64_t)row * p.ldv + posv[j]] = Vs[row * ld + j] - T(0.5) * a;
    }
  } else {  // double accumulation is already orthogonal to ~1e-14 (or V is still exactly the identity)
    for (int idx = tid; idx < n * n; idx += NTH) {
      const int row = idx / n, j = idx - row * n;
      V[(int64_t)row * p.ldv + posv[j]] = Vs[row * ld + j];
    }
  }
  if (n < nf) {  // the frozen tail: unit eigenvectors, zeros elsewhere
    for (int idx = tid; idx < nf * nf; idx += NTH) {
      const int row = idx / nf, j = idx - row * nf;
      if (row >= n || j >= n) V[(int64_t)row * p.ldv + posv[j]] = (row == j) ? T(1) : T(0);
    }
  }
  T* __restrict__ sout = p.sigma + bt * p.stride_sigma;
  for (int i = tid; i < nf; i += NTH) sout[i] = p.eig_mode == TTR_EIG_MATCH_DIAG ? sig[i] : sig_sorted[i];
  if (tid == 0) {
    // (the rank rule sees the whole spectrum; zero guard: round.py:137-145)
    const T d2 = p.use_delta ? (T)(p.delta2_dev ? *p.delta2_dev : p.delta2) : T(0);
    p.info[bt] = rank_rule<T>(sig_sorted, nf, nf, p.rmax, p.use_delta, d2, p.noise_c);
    if (p.sweeps) p.sweeps[bt] = sweeps_used;
    if (p.rot_count && sweeps_used > 0) atomicAdd(p.rot_count, 1);
  }
}

// ====================================================================================================
// Tridiagonal QL eigensolver (n <= 64): one workgroup of TWO waves per matrix (see the kernel below).
// (Measured alternative: vectors / d / e / rotations in registers with v_readlane broadcasts instead of LDS
// broadcast reads -- 0.81 -> 1.1 ms per launch of 2048 matrices; the LDS broadcast is the cheaper one.)
//
// Householder reduction to tridiagonal form, accumulation of Q on the matrix cores, implicit-shift QL with the
// rotations applied to Q's columns: ~10x fewer flops than cyclic Jacobi (n^3-class constant 4/3+4/3+~3
// instead of ~50), so many problems run per CU.  Absolute accuracy
// O(eps * ||G||) (LAPACK steqr class): this is the pass-1 / 'eig' solver.  Pass 2 of the 'svd' algorithm
// needs the relative accuracy of Jacobi on a graded matrix and keeps the Jacobi kernel.
//   lane = row      for the reduction (symmetric mat-vec, rank-2 update) and for applying QL rotations
// The matrix is scaled by 1/max|entry| on load (fp32 Gram matrices of the metric workload reach 1e24, whose
// squares overflow) and the eigenvalues are scaled back at the end.
// r = sqrt(f^2 + g^2) and 1/r for the QL rotations.  The recurrence is computed redundantly by all 64 lanes of its wave and
// sits on the kernel's critical chain, so fp32 uses the hardware rsq (1 ulp, 3 instructions) instead
// of the IEEE sqrt and divide expansions (~25).
__device__ __forceinline__ bool givens_norm(float f, float g, float& r, float& rinv) {
  // The matrix is scaled to max |G_ii| = 1, so x cannot overflow; below the normal range (x == 0 included) the rotation is
  // the "underflow" case of the QL sweep: returns false, r = 0, and rinv is NOT usable (inf / garbage: the caller leaves
  // the sweep after this rotation and must not let values derived from rinv reach d / e).
  const float x = f * f + g * g;
  rinv = __builtin_amdgcn_rsqf(x);  // 1 ulp; a Newton step on top (4 more instructions ON the chain) did not improve
                                    // |V^T V - I|, the residual or the eigenvalues (tools/eigh_orth_probe.py: 5e-6 / 1e-6 / 1.5e-6 either way)
  const bool ok = x > 1e-36f;
  r = ok ? x * rinv : 0.f;
  return ok;
}
__device__ __forceinline__ bool givens_norm(double f, double g, double& r, double& rinv) {
  // fp64: the hardware rsq estimate + two Newton steps for 1 / sqrt(x) (12 instructions; the IEEE sqrt and the division
  // that used to sit here expand to ~30 on the recurrence's chain -- the fp64 rotation was 75 instructions).  x <= 1 by the
  // scaling; below 1e-290 the rotation is the sweep's underflow case, as in fp32.
  const double x = fma(f, f, g * g);
  const bool ok = x > 1e-290;
  double y = __builtin_amdgcn_rsq(x);
  double h = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, h, y);
  h = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, h, y);
  rinv = y;
  r = ok ? x * y : 0.;
  return ok;
}

// beta = -sign(alpha) sqrt(alpha^2 + ss), tau = (beta - alpha) / beta, scale = 1 / (alpha - beta) (LAPACK larfg).  fp32: the
// 1-ulp hardware sqrt / reciprocal -- the IEEE expansions are ~45 dependent instructions per reflector on the wave's serial
// chain; H = I - tau v v^T stays orthogonal to O(eps) (same choice as in the QR kernels).
__device__ __forceinline__ void householder_scalars(float alpha, float ss, float& beta, float& tau, float& scale) {
  const float n2 = alpha * alpha + ss;
  if (n2 > 1e-30f && n2 < 1e30f) {  // (wave-uniform) the hardware sqrt / rcp flush denormals: graded matrices get there
    beta = -copysignf(__builtin_amdgcn_sqrtf(n2), alpha);
    tau = (beta - alpha) * __builtin_amdgcn_rcpf(beta);
    scale = __builtin_amdgcn_rcpf(alpha - beta);
  } else {
    beta = -copysignf(sqrtf(n2), alpha);
    tau = (beta - alpha) / beta;
    scale = 1.0f / (alpha - beta);
  }
}
__device__ __forceinline__ void householder_scalars(double alpha, double ss, double& beta, double& tau, double& scale) {
  beta = -copysign(sqrt(alpha * alpha + ss), alpha);
  tau = (beta - alpha) / beta;
  scale = 1.0 / (alpha - beta);
}

// Wilkinson-type shift of a QL sweep: g = d[m] - d[l] + e[l] / (t + sign(t) sqrt(t^2 + 1)),  t = (d[l+1] - d[l]) / (2 e[l]).  Only the
// convergence rate depends on its accuracy, so fp32 takes the 1-ulp hardware reciprocal / sqrt: the two IEEE divisions and the
// sqrt are ~30 instructions per sweep on the wave's serial chain.  e[l] is not deflated here (> eps * max(|d|, |e|) of a matrix
// scaled to max |entry| = 1), so neither reciprocal sees a denormal.
__device__ __forceinline__ float ql_shift(float dl, float dl1, float dm, float el) {
  const float t = (dl1 - dl) * __builtin_amdgcn_rcpf(2.0f * el);
  const float r = __builtin_amdgcn_sqrtf(fmaf(t, t, 1.0f));
  return dm - dl + el * __builtin_amdgcn_rcpf(t + copysignf(r, t));
}
__device__ __forceinline__ double ql_shift(double dl, double dl1, double dm, double el) {
  const double t = (dl1 - dl) / (2.0 * el);
  const double r = sqrt(t * t + 1.0);
  return dm - dl + el / (t + copysign(r, t));
}

// a / b with the 1-ulp hardware reciprocal in fp32 (pivots of LDL^T factorisations: the Sturm counts and twisted
// factorisations below only need them to a few ulp)
__device__ __forceinline__ float fdiv_fast(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ double fdiv_fast(double a, double b) { return a / b; }

// The r LARGEST eigenpairs of a symmetric n x n matrix, n <= 64, r <= 32, for pass 1 of the batch-mode 'svd' truncation when the
// kept spectrum is flat (ttr_spectrum_flat's criterion, decided HERE from the eigenvalues) -- what the QL kernel above spends
// its time on, one serial rotation chain through ALL n eigenvectors (316 k of its 560 k cycles), is replaced by work that is
// parallel over the eigenvalues.  Same two waves, same tridiagonalisation (wave 0) with the forward Q formation on the matrix
// cores under it (wave 1: Q^T stays in its accumulators); then
//   * the r largest eigenvalues by Sturm-count MULTISECTION: 128 / r shifts per eigenvalue and round, every lane runs the
//     n-step count recurrence for its own shift; the groups of an eigenvalue never straddle the waves, so the waves do not
//     talk (11 rounds for r = 32);
//   * their eigenvectors from the TWISTED factorisation of T - lambda I (one lane per vector, D+ / D- of the stationary qd
//     transforms in 64 registers, twist where |gamma| is smallest);
//   * orthonormalisation and back-transformation on the matrix cores without a transposition: with Z (n x r) as the A and the
//     B operand, S = Z^T Z, C = (Q Z)^T = Z^T Q^T (Q^T from the accumulators as the B operand), two (fp64: three) Newton-Schulz steps on
//     the r x r matrices (P = 1.5 I - 0.5 S is symmetric, so its accumulator registers ARE its A operand) and V^T = P C.
// An item qualifies when sigma_r >= thr sigma_1 > 0 and no two of the r eigenvalues are closer than 512 eps lambda_1 (the
// twisted vectors of closer pairs are too parallel for two Newton-Schulz steps); V[:, r:] and sigma[r:] are then written as
// zeros.  The others FALL THROUGH to the QL phase of the same launch (one more barrier; wave 1 stores Q^T from the same
// accumulators) and get the full decomposition: eigh_tridiag_kernel<T, true> is the QL kernel with this path in front.
template <int I>
using IC2 = std::integral_constant<int, I>;

// Two waves per matrix.  A single wave issues one VALU instruction per ~8 clocks whatever its dependencies (tools/microbench:
// independent and dependent FMA chains both 8.0-8.5; four waves per SIMD: 2.4 per instruction), and at B <= 2048 there are at
// most two matrices per SIMD -- the one-wave kernel of rounds 1-2 (everything below on one wave, Q formed backwards after the
// tridiagonalisation) was bound by the INSTRUCTION COUNT of its single wave (cycle stamps, B = 1 and B = 2048 alike: 163 k
// tridiagonalisation + 75 k Q formation + 309 k QL recurrence + 96 k rotation replay = 318 us for one matrix), not by the
// chip.  Here the work that does not sit on the recurrence's chain runs on a second wave of the same workgroup (255 us;
// profiles/r03_eigh_two_wave_ab.txt, A/B at commit a699024):
//   wave 0   tridiagonalisation, then the QL recurrence on (d, e); each sweep's rotations are published to LDS (two buffers)
//   wave 1   Q^T = H_{n-2} ... H_0, accumulated FORWARD (X <- (I - V_b T_b^T V_b^T) X for b = 0, 1, ...: block b only needs the
//            reflectors of the 16 steps wave 0 has just finished, so the Q formation hides under the tridiagonalisation
//            except for its last block), stored transposed; then it replays the published sweeps on Q's rows.
// Workgroup barriers are the only synchronisation: one per reflector block, then one per QL sweep (wave 0 arrives after
// computing sweep s, wave 1 before replaying it, so wave 0 runs at most two sweeps ahead and never overwrites a buffer
// that is still being replayed); the number of sweeps is data dependent, so every barrier's sweep carries a `done` word.
//
// NMAX = 32 (round 4; top-r launches only): the SAME kernel instantiated for problems of at most 32 rows -- the zero-tail items of a
// 64 x 64 launch, i.e. every bond of a rank-inflated train.  One register allocation serves both waves and both problem sizes:
// with 64 accumulator registers of Q^T and 64 pivots of the twisted factorisation the 64-row instance spills 107 VGPRs under its
// four-waves-per-SIMD cap, and every unrolled loop walks 64 guarded steps.  At NMAX = 32 everything is half the size (16 + 32
// registers), nothing spills, and wave 0 keeps its row of the matrix in REGISTERS through the fully unrolled tridiagonalisation.
// It runs first (top-r launches and plain QL launches on Gram matrices alike); items without a zero tail get info[b] = -1 and are
// solved by the NMAX = 64 launch that follows, which skips everything else (items the top-r path declines fall through to the QL phase of their own launch, as ever: handing them to the
// second launch made it as long as its slowest block -- 171 us per launch at B = 2048 for ~1 % of the items, measured).
template <typename T, bool TOP = false, int NMAX = 64, int MINW = 0>
__global__ __launch_bounds__(2 * kWave, (MINW > 0 ? MINW : (sizeof(T) == 4 && NMAX == 64 ? 4 : 2))) void eigh_tridiag_kernel(EighArgs<T> p) {
  static_assert(NMAX == 64 || NMAX == 32, "64-row kernel and its 32-row instance");
  constexpr int NT = NMAX / 16;   // 16-row tiles
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid) >> 6;  // provably wave-uniform
  const int64_t bt = blockIdx.x;
  const int nf = p.n;
  if constexpr (NMAX == 64) {
    if (p.top_pre && p.info[bt] != -1) return;   // (block-uniform) solved by the 32-row launch (which leaves info = -1 on the others)
  }
  // Gram matrices (every launch but the block-Jacobi driver's TTR_EIG_MATCH_DIAG pair problems; G_ii = 0 means a zero row /
  // column of a positive semi-definite matrix): a 64 x 64 matrix whose
  // diagonal is exactly zero from index 32 on -- the carry of a bond whose QR packed its rows (ttr_qr_pushed_flag_offset) -- is
  // solved as its leading 32 x 32 block: half the reflectors at half the length, half the Sturm / twisted recurrences.  V comes
  // out as blockdiag(V11, I) (qualified items: zeros beyond the r kept columns), sigma[32..] = 0.  Decided by every wave from the
  // diagonal (lane = index), before the LDS is carved.
  int n_shrunk = nf;
  if ((TOP || p.eig_mode != TTR_EIG_MATCH_DIAG) && nf == 64) {   // (RAW / REF clamp negative eigenvalues: their input is a Gram matrix)
    const T* __restrict__ Gd = p.G + bt * p.strideG + (int64_t)lane * p.ldg + lane;
    T dg = T(0);
    if (lane >= 32)
      for (int pt = 0; pt < p.gparts; ++pt) dg += fabs(Gd[pt * p.stride_gpart]);
    if (__ballot(dg != T(0)) == 0ull) n_shrunk = 32;
  }
  if constexpr (NMAX == 32) {
    if (n_shrunk != 32) {   // (block-uniform) not this launch's item
      if (tid == 0) p.info[bt] = -1;
      return;
    }
  }
  const int n = n_shrunk;
  const int ld = n + 1;
  // beyond the solved block (n < nf): unit eigenvectors with zero eigenvalues, or nothing at all for an item of the top-r path
  auto pad_tail = [&](bool ident, int t0, int nt) {
    if (n == nf) return;
    T* __restrict__ Vp = p.V + bt * p.strideV;
    for (int idx = t0; idx < nf * nf; idx += nt) {
      const int row = idx / nf, j = idx - row * nf;
      if (row >= n || j >= n) Vp[(int64_t)row * p.ldv + j] = (ident && row == j) ? T(1) : T(0);
    }
    for (int i = n + t0; i < nf; i += nt) p.sigma[bt * p.stride_sigma + i] = T(0);
  };
  T* A = reinterpret_cast<T*>(smem_raw);     // [n][ld]  G -> reflectors -> Q -> eigenvectors
  T* vs = A + n * ld + 3;                    // scratch region of 408 elements:
  T* wsv = vs + 64;                          //   wave 0: the broadcast arrays v / w of the tridiagonalisation (first 131 elements), then
  T* cv = wsv + 64;                          //           rotation buffer 0 of the QL phase
  T* sv = cv + 64;                           //   wave 1: S_b of the Q formation (elements 132 .. 403)
  int* posv = reinterpret_cast<int*>(sv + 64);  //   epilogue (after the last barrier): sigma, sorted sigma, diag(G), column order, sort positions
  T* const Sq = vs + 132;                    // [16][17]
  T* dv = vs + 408;                          // [66] diagonal / eigenvalues   } the QL phase keeps d / e in registers: dv .. ev is
  T* ev = dv + 66;                           // [66] sub-diagonal             } rotation buffer 1 meanwhile
  T* tauv = ev + 66;                         // [64]
  T* Tb = tauv + 64;                         // [16][17] compact-WY factor of a reflector block (Q formation)
  volatile int* meta = reinterpret_cast<volatile int*>(Tb + 16 * 17);  // [2][4]: m, ilast, done of the sweep in buffer 0 / 1
  // top-r path: d_i and e_{i-1}^2 (clamped to the smallest normal; [0] = 0) for the Sturm counts, 16-byte aligned so that four
  // steps' values come with two LDS reads
  constexpr int kSel16 = 16 / (int)sizeof(T);
  T* const dsel = A + ((n * ld + 3 + 408 + 66 * 2 + 64 + 16 * 17 + 8 + kSel16 - 1) / kSel16) * kSel16;   // [64]
  T* const e2s = dsel + 64;                                                                                 // [64]

  const T* __restrict__ G = p.G + bt * p.strideG;
  // ---- load + scale (both waves)
  if ((n == 64 || (n == 32 && nf == 64)) && p.ldg == 64 && (p.stride_gpart & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(G) & (4 * sizeof(T) - 1)) == 0) {
    // a contiguous 64 x 64 matrix (+ split partials, e.g. the 8 per-block Gram partials of ttr_qr_apply_pushed_gram): four
    // elements per lane and load, the partials' loads of one position issued four at a time -- an element-wise loop has
    // one dependent load in flight per lane, which costs 20 us per partial and launch at B = 2048
    typedef T VT __attribute__((ext_vector_type(4)));
    const int q4 = n >> 2;  // vectors of four per (stored) row of the block
    for (int it = 0; it < (n * n) / (8 * kWave); ++it) {
      const int q = it * 2 * kWave + tid, qr = q / q4;
      const int e = qr * 64 + (q - qr * q4) * 4;   // element offset in the 64-column source
      const T* __restrict__ src = G + e;
      VT acc = *reinterpret_cast<const VT*>(src);
      int pt = 1;
      for (; pt + 3 < p.gparts; pt += 4) {
        const VT x0 = *reinterpret_cast<const VT*>(src + (int64_t)pt * p.stride_gpart);
        const VT x1 = *reinterpret_cast<const VT*>(src + (int64_t)(pt + 1) * p.stride_gpart);
        const VT x2 = *reinterpret_cast<const VT*>(src + (int64_t)(pt + 2) * p.stride_gpart);
        const VT x3 = *reinterpret_cast<const VT*>(src + (int64_t)(pt + 3) * p.stride_gpart);
        acc += x0; acc += x1; acc += x2; acc += x3;  // same summation order as the element-wise loop
      }
      for (; pt < p.gparts; ++pt) acc += *reinterpret_cast<const VT*>(src + (int64_t)pt * p.stride_gpart);
      T* dst = &A[(e >> 6) * ld + (e & 63)];
      dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = acc[2]; dst[3] = acc[3];
    }
  } else {
    for (int idx = tid; idx < n * n; idx += 2 * kWave) {
      const int i = idx / n, j = idx - i * n;
      T gv = G[(int64_t)i * p.ldg + j];
      for (int pt = 1; pt < p.gparts; ++pt) gv += G[pt * p.stride_gpart + (int64_t)i * p.ldg + j];
      A[i * ld + j] = gv;
    }
  }
  __syncthreads();
  const T gdiag = (lane < n) ? A[lane * ld + lane] : T(0);
  // scale by the largest |entry| (= the largest diagonal entry for a Gram matrix): afterwards every entry is <= 1 and
  // ||A||_F <= n, so the squares formed by the QL rotations cannot overflow (their fast path has no range branch)
  T gmax = T(0);  // (each wave over the whole matrix: the same value in both, no exchange)
  for (int idx = lane; idx < n * n; idx += kWave) {
    const int i = idx / n, j = idx - i * n;
    gmax = fmax(gmax, fabs(A[i * ld + j]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, off, 64));
  const T ginv = gmax > T(0) ? T(1) / gmax : T(0);
  __syncthreads();  // both waves have read the unscaled matrix
  for (int idx = tid; idx < n * n; idx += 2 * kWave) {
    const int i = idx / n, j = idx - i * n;
    A[i * ld + j] *= ginv;
  }
  __syncthreads();

#ifdef TTR_EIGH_STAMPS
  long long* const dbg = reinterpret_cast<long long*>(p.ws);  // diagnostics build: cycle stamps of matrix 0 (wave 0)
  int dbgi = 0;
#define TTR_ESTAMP() do { if (dbg && bt == 0 && tid == 0) dbg[dbgi++] = (long long)clock64(); } while (0)
  int dbgj = 32;  // wave 1's stamps (lane 0 of wave 1) go to slots 32..
#define TTR_ESTAMP1() do { if (dbg && bt == 0 && tid == kWave) dbg[dbgj++] = (long long)clock64(); } while (0)
#else
#define TTR_ESTAMP() do {} while (0)
#define TTR_ESTAMP1() do {} while (0)
#endif
  TTR_ESTAMP();
  T* const rowp = A + lane * ld;
  constexpr int CH = 8;
  typedef T T2 __attribute__((ext_vector_type(2)));
  typedef T T4 __attribute__((ext_vector_type(4)));
  constexpr int kPer16 = 16 / (int)sizeof(T);
  // broadcast arrays of the tridiagonalisation (v, w), 16-byte aligned (offset arithmetic on the LDS pointer: a round trip
  // through uintptr_t loses the address space and turns the reads into flat loads)
  T* const vsh = vs + (kPer16 - (n * ld + 3) % kPer16) % kPer16;
  T* const wsh = vsh + 64;
  T2* const cs0 = reinterpret_cast<T2*>(vsh);                                // rotation buffers: (c, s) of rotation i at [i + 1]
  T2* const cs1 = reinterpret_cast<T2*>(dv + ((n * ld + 3 + 408) & 1));
  const int nblk = (n - 1 + 15) / 16;  // blocks of 16 reflectors (0 for n = 1)
  T dreg = T(0), ereg = T(0);
  int total_iter = 0;
#ifdef TTR_EIGH_STAMPS
  long long nrot_dbg = 0, qlrec_dbg = 0;
#endif
  if (n == 1 && tid == 0) { dv[0] = A[0]; ev[0] = T(0); tauv[0] = T(0); }
  // ---- selection set-up of the top-r path (TOP; everything below is dead code otherwise)
  const int rsel = TOP ? (p.top_r < n ? (p.top_r < 1 ? 1 : p.top_r) : n) : 1;  // eigenpairs wanted (host: <= 32)
  int kp = 1;
  while (kp < rsel) kp <<= 1;
  const int GL = kp >= 2 ? (2 * kWave) / kp : kWave;            // lanes per eigenvalue; a group never straddles the two waves
  const int gj = kp >= 2 ? tid / GL : (wv == 0 ? 0 : 1);        // eigenvalue of this lane's group
  const int gs = kp >= 2 ? tid % GL : lane;
  const int nvec = kp >= 2 ? kp / 2 : 1;                       // vectors per wave: wave w owns eigenvalues w * nvec + (0 .. nvec - 1)
  const int jv = wv * nvec + lane;                             // this lane's vector (lane < nvec)
  const bool vlive = lane < nvec && jv < rsel;
  T* const lamv = vsh;                                         // [64] eigenvalues (the tridiagonalisation's broadcast arrays are free by then)
  volatile int* const badf = meta + 3;                         // [1] some vector overflowed (a spare control word)
  const int ZLD = n >= 40 ? 36 : n + 1;                        // Z[i][j] at A[i * ZLD + j] (A is free once Q sits in registers; n rows of <= 32 vectors fit A's n (n + 1))
  const T eps = Num<T>::eps();
  if (TOP && tid == 0) badf[0] = 0;
  // eigenvalues of this lane's group + the twisted factorisation of this lane's vector (everything a wave needs comes from
  // d / e in LDS, which are final before the tridiagonalisation's last barrier)
  T Dp[NMAX];    // D+_i for i <= twist, D-_i above it
  int twist = 0;
  T lam_vec = T(0), pivmin = T(0);
  auto select = [&]() __attribute__((always_inline)) {
    T gl = Num<T>::big_theta(), gu = -Num<T>::big_theta(), emax2 = T(0);
    if (lane < n) {
      const T di = dv[lane], ei = lane + 1 < n ? ev[lane] : T(0), em = lane > 0 ? ev[lane - 1] : T(0);
      const T rad = fabs(ei) + fabs(em);
      gl = di - rad; gu = di + rad; emax2 = ei * ei;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      gl = fmin(gl, __shfl_xor(gl, off, 64)); gu = fmax(gu, __shfl_xor(gu, off, 64)); emax2 = fmax(emax2, __shfl_xor(emax2, off, 64));
    }
    const T tnorm = fmax(fabs(gl), fabs(gu));
    pivmin = Num<T>::tiny() * fmax(T(1), emax2) / eps;
    gl -= T(2.1) * tnorm * eps * n + T(2.1) * pivmin;
    gu += T(2.1) * tnorm * eps * n + T(2.1) * pivmin;
    // number of eigenvalues < sigma: the signs of the pivots q_i = d_i - sigma - e_{i-1}^2 / q_{i-1} of the LDL^T factorisation.  No
    // pivot guard: a pivot that is exactly zero (or flushed to zero) gives e^2 / 0 = +inf, the next pivot is -inf (counted: the
    // zero pivot stands for an eigenvalue AT sigma), the one after that d - sigma -- IEEE arithmetic does what the guard did
    // (LAPACK dstebz's ieee variant); e^2 is clamped away from zero so that 0 * inf cannot occur.  Four steps per trip with two
    // 16-byte LDS reads (cycle stamps, 32 x 32 problem: 57 k of 207 k cycles in this phase at ~127 cycles per step, 16 issue slots
    // of the lone wave, before)
    auto count_below = [&](T sigma) __attribute__((always_inline)) {
      typedef T Q4 __attribute__((ext_vector_type(4)));
      T q = T(1);
      int cnt = 0;
      int i = 0;
      for (; i + 4 <= n; i += 4) {
        Q4 d4, e4;
        if constexpr (sizeof(T) == 4) {
          d4 = *reinterpret_cast<const Q4*>(&dsel[i]); e4 = *reinterpret_cast<const Q4*>(&e2s[i]);
        } else {
          typedef T Q2 __attribute__((ext_vector_type(2)));
          const Q2 da = *reinterpret_cast<const Q2*>(&dsel[i]), db = *reinterpret_cast<const Q2*>(&dsel[i + 2]);
          const Q2 ea = *reinterpret_cast<const Q2*>(&e2s[i]), eb = *reinterpret_cast<const Q2*>(&e2s[i + 2]);
          d4 = Q4{da[0], da[1], db[0], db[1]}; e4 = Q4{ea[0], ea[1], eb[0], eb[1]};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          q = (d4[u] - sigma) - fdiv_fast(e4[u], q);
          cnt += q < T(0) ? 1 : 0;
        }
      }
      for (; i < n; ++i) {
        q = (dsel[i] - sigma) - fdiv_fast(e2s[i], q);
        cnt += q < T(0) ? 1 : 0;
      }
      return cnt;
    };
    const bool live = gj < rsel;
    const int idx = n - 1 - (live ? gj : 0);  // ascending index of the gj-th largest eigenvalue
    const int jloc = gj - wv * nvec;          // group number inside this wave
    T lo = gl, hi = gu;
    const int max_rounds = sizeof(T) == 4 ? 40 : 90;
    for (int it = 0; it < max_rounds; ++it) {
      const T wdt = hi - lo;
      const bool done = !(wdt > T(4) * eps * fmax(fabs(lo), fabs(hi)) + T(2) * pivmin);
      if (__ballot(live && !done) == 0ull) break;
      const T sigma = lo + wdt * (T)(gs + 1) / (T)(GL + 1);
      const bool above = count_below(sigma) >= idx + 1;
      const unsigned long long mask = __ballot(above);
      const unsigned long long grp = (mask >> ((jloc < 0 ? 0 : jloc) * GL)) & (GL == 64 ? ~0ull : ((1ull << GL) - 1ull));
      const int f = grp ? __ffsll((long long)grp) - 1 : GL;
      const T nlo = f == 0 ? lo : lo + wdt * (T)f / (T)(GL + 1);
      const T nhi = f == GL ? hi : lo + wdt * (T)(f + 1) / (T)(GL + 1);
      if (!done) { lo = nlo; hi = nhi; }
    }
    const T lam_grp = T(0.5) * (lo + hi);
    if (live && gs == 0) lamv[gj] = lam_grp;
    lam_vec = __shfl(lam_grp, (lane < nvec ? lane : 0) * (kp >= 2 ? GL : 0), 64);
    // twisted factorisation: D+ from the top, gamma from the bottom, then D- kept above the twist
    const T lam = lam_vec;
    T q = dv[0] - lam;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      if (i < n) {
        if (fabs(q) < pivmin) q = -pivmin;
        Dp[i] = q;
        if (i + 1 < n) { const T e = ev[i]; q = dv[i + 1] - lam - fdiv_fast(e * e, q); }
      } else {
        Dp[i] = T(1);
      }
    }
    T best = fabs(Dp[0]);
    twist = 0;
    {
      T qm = T(0);
#pragma unroll
      for (int i = NMAX - 1; i >= 0; --i) {
        if (i < n) {
          if (i == n - 1) qm = dv[i] - lam;
          else { const T e = ev[i]; if (fabs(qm) < pivmin) qm = -pivmin; qm = dv[i] - lam - fdiv_fast(e * e, qm); }
          const T gam = Dp[i] + qm - (dv[i] - lam);
          if (i == n - 1 || fabs(gam) < best) { best = fabs(gam); twist = i; }
        }
      }
    }
    {
      T qm = T(0);
#pragma unroll
      for (int i = NMAX - 1; i >= 1; --i) {
        if (i < n) {
          if (i == n - 1) qm = dv[i] - lam;
          else { const T e = ev[i]; qm = dv[i] - lam - fdiv_fast(e * e, qm); }
          if (fabs(qm) < pivmin) qm = -pivmin;
          if (i > twist) Dp[i] = qm;
        }
      }
    }
  };
  // x_twist = 1;  x_i = -(e_i / D+_i) x_{i+1} (i < twist);  x_{i+1} = -(e_i / D-_{i+1}) x_i (i >= twist).  WRITE = false: only ||x||^2.
  auto vector_pass = [&](auto WRITE, T scale) __attribute__((always_inline)) -> T {
    constexpr bool kWrite = decltype(WRITE)::value != 0;
    T nrm2 = T(1), xc = T(0);
    if (kWrite && twist < n) A[twist * ZLD + jv] = scale;
#pragma unroll
    for (int i = NMAX - 2; i >= 0; --i) {  // upwards
      if (i + 1 < n) {
        if (i + 1 == twist) xc = T(1);
        if (i < twist) {
          xc = -fdiv_fast(ev[i], Dp[i]) * xc;
          nrm2 += xc * xc;
          if (kWrite) A[i * ZLD + jv] = xc * scale;
        }
      }
    }
    xc = T(0);
#pragma unroll
    for (int i = 0; i < NMAX - 1; ++i) {  // downwards
      if (i + 1 < n) {
        if (i == twist) xc = T(1);
        if (i >= twist) {
          xc = -fdiv_fast(ev[i], Dp[i + 1]) * xc;
          nrm2 += xc * xc;
          if (kWrite) A[(i + 1) * ZLD + jv] = xc * scale;
        }
      }
    }
    return nrm2;
  };
  // after both waves have their eigenvalues in lamv: does the item qualify?  (same answer in every thread)
  auto qualifies = [&]() __attribute__((always_inline)) -> bool {
    T gmin = Num<T>::big_theta();
    if (lane + 1 < rsel) gmin = lamv[lane] - lamv[lane + 1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmin = fmin(gmin, __shfl_xor(gmin, off, 64));
    const T l0 = lamv[0], lr = lamv[rsel - 1];
    const T thr = (T)p.top_thr;
    return l0 > T(0) && lr >= thr * thr * l0 && (rsel == 1 || gmin >= T(512) * eps * l0) && (!p.top_need_all || rsel == n);
  };
  T* const invn = wsh;   // [<= 32] 1 / ||x_j|| of the twisted vectors (the tridiagonalisation's second broadcast array is free by then)

  if (wv == 0) {
    // ---- 1. Householder tridiagonalisation (lower): reflector k annihilates A[k+2:, k]; a barrier after every block of 16
    // reflectors hands the block to wave 1.
    // lane = row, and every lane only ever touches ITS OWN row of A here: the reflector v and the vector w stay in
    // registers (one entry per lane) and are broadcast through LDS arrays, the row is walked in
    // chunks of eight (loads, arithmetic, stores -- the element-by-element read-modify-write through LDS was serialised at
    // the full LDS latency by the possible aliasing of A with the broadcast arrays).
    if constexpr (NMAX == 32) {
      // The lane's ROW lives in registers (16 pairs) and the k loop is fully unrolled: column k of the own row, the chunks a step
      // still touches (columns > k) and every register index are compile-time constants -- no row loads / stores through LDS
      // (cycle stamps of one generic step at n = 32: 964 of 3.1 k cycles in the matrix-vector product, 1060 in the rank-2 update,
      // nearly all of it 4-byte LDS traffic: the row stride n + 1 is odd).  Only the reflector (for wave 1's Q formation), d, e
      // and tau go to LDS.  Same arithmetic as the generic loop below.
      typedef T R2 __attribute__((ext_vector_type(2)));
      R2 a2[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const T lo = (lane < n && 2 * j < n) ? rowp[2 * j] : T(0), hi = (lane < n && 2 * j + 1 < n) ? rowp[2 * j + 1] : T(0);
        a2[j] = R2{lo, hi};
      }
#pragma unroll
      for (int k = 0; k < 31; ++k) {
        if (k + 1 < n) {   // (wave-uniform)
          const bool below = lane >= k + 2 && lane < n;
          const T ak = a2[k >> 1][k & 1];
          const T xr = (lane >= k + 1 && lane < n) ? ak : T(0);
          const T x = below ? xr : T(0);
          const T alpha = lane_get(xr, k + 1);
          const T xn2 = wave_sum_dpp(x * x);
          T beta = alpha, t = T(0), v = (lane == k + 1) ? T(1) : T(0);
          if (xn2 != T(0)) {
            T scale;
            householder_scalars(alpha, xn2, beta, t, scale);
            if (below) v = x * scale;
          }
          if (lane == 0) { ev[k] = beta; tauv[k] = t; }
          if (t != T(0)) {
            const bool act = lane >= k + 1 && lane < n;
            vsh[lane] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            R2 p0 = {T(0), T(0)}, p1 = {T(0), T(0)};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              if (4 * c + 3 >= k + 1 && 4 * c < n) {   // (columns <= k: v = 0 exactly)
                const T4 vv = *reinterpret_cast<const T4*>(&vsh[4 * c]);
                p0 += a2[2 * c] * R2{vv[0], vv[1]};
                p1 += a2[2 * c + 1] * R2{vv[2], vv[3]};
              }
            }
            T pr = (p0[0] + p0[1]) + (p1[0] + p1[1]);
            pr = act ? pr * t : T(0);
            const T dot = wave_sum_dpp(pr * v);
            const T w = pr - T(0.5) * t * dot * v;
            wsh[lane] = w;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const R2 mv = {-v, -v}, mw = {-w, -w};   // (lanes outside the trailing block: v = w = 0, their rows stay)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              if (4 * c + 3 >= k + 1 && 4 * c < n) {
                const T4 vv = *reinterpret_cast<const T4*>(&vsh[4 * c]), ww = *reinterpret_cast<const T4*>(&wsh[4 * c]);
                a2[2 * c] += mv * R2{ww[0], ww[1]};
                a2[2 * c + 1] += mv * R2{ww[2], ww[3]};
                a2[2 * c] += mw * R2{vv[0], vv[1]};
                a2[2 * c + 1] += mw * R2{vv[2], vv[3]};
              }
            }
          }
          if (below) rowp[k] = v;  // the reflector below the sub-diagonal, for wave 1
          if (lane == k) dv[k] = a2[k >> 1][k & 1];
          if (k == n - 2 && lane == n - 1) { dv[k + 1] = a2[(k + 1) >> 1][(k + 1) & 1]; ev[k + 1] = T(0); }
          if (TOP && k == n - 2) {   // d / e are final: the Sturm counts' copies (both waves read them after the barrier below)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const T el = (lane >= 1 && lane < n) ? ev[lane - 1] : T(0);
            dsel[lane] = lane < n ? dv[lane] : T(0);
            e2s[lane] = lane == 0 ? T(0) : fmax(el * el, Num<T>::tiny());
          }
          if ((k & 15) == 15 || k == n - 2) __syncthreads();
        }
      }
    } else
    for (int k = 0; k + 1 < n; ++k) {
      const bool below = lane >= k + 2 && lane < n;
      const T xr = (lane >= k + 1 && lane < n) ? rowp[k] : T(0);  // column k of the trailing block (symmetric: own row)
      const T x = below ? xr : T(0);
      const T alpha = lane_get(xr, k + 1);
      const T xn2 = wave_sum_dpp(x * x);
      T beta = alpha, t = T(0), v = (lane == k + 1) ? T(1) : T(0);
      if (xn2 != T(0)) {
        T scale;
        householder_scalars(alpha, xn2, beta, t, scale);
        if (below) v = x * scale;
      }
      if (lane == 0) { ev[k] = beta; tauv[k] = t; }
      if (t != T(0)) {
        const bool act = lane >= k + 1 && lane < n;
        // v (and below w) are broadcast through two 16-byte aligned LDS arrays: one ds_write per lane, then every lane reads
        // the values of a chunk of eight columns with two 16-byte broadcast reads -- v_readlane costs one instruction (plus a
        // hazard slot) per VALUE, 24 per chunk of the rank-2 update.  Chunks start at a multiple of eight at or below k + 1:
        // v and w are exactly zero in the columns <= k, which are therefore re-written unchanged.
        vsh[lane] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int j0 = (k + 1) & ~(CH - 1), nfull = n & ~(CH - 1);
        T pr = 0;
        {
          T2 p0 = {T(0), T(0)}, p1 = {T(0), T(0)};  // four chains
          int j = j0;
          for (; j < nfull; j += CH) {
            T a8[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) a8[u] = rowp[j + u];
            const T4 va = *reinterpret_cast<const T4*>(&vsh[j]), vb = *reinterpret_cast<const T4*>(&vsh[j + 4]);
            p0 += T2{a8[0], a8[1]} * T2{va[0], va[1]};
            p1 += T2{a8[2], a8[3]} * T2{va[2], va[3]};
            p0 += T2{a8[4], a8[5]} * T2{vb[0], vb[1]};
            p1 += T2{a8[6], a8[7]} * T2{vb[2], vb[3]};
          }
          for (j = (j < k + 1) ? k + 1 : j; j < n; ++j) pr += rowp[j] * vsh[j];
          pr += (p0[0] + p0[1]) + (p1[0] + p1[1]);
        }
        pr = act ? pr * t : T(0);
        const T dot = wave_sum_dpp(pr * v);
        const T w = pr - T(0.5) * t * dot * v;
        wsh[lane] = w;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
          // a[j] -= v w_j + w v_j
          struct Chunk { T a[CH]; T4 v0, v1, w0, w1; };
          auto load = [&](int jj, Chunk& c) {
#pragma unroll
            for (int u = 0; u < CH; ++u) c.a[u] = rowp[jj + u];
            c.v0 = *reinterpret_cast<const T4*>(&vsh[jj]); c.v1 = *reinterpret_cast<const T4*>(&vsh[jj + 4]);
            c.w0 = *reinterpret_cast<const T4*>(&wsh[jj]); c.w1 = *reinterpret_cast<const T4*>(&wsh[jj + 4]);
          };
          const T2 mv = {-v, -v}, mw = {-w, -w};
          auto proc = [&](int jj, Chunk& c) {
            T2 r0 = T2{c.a[0], c.a[1]} + mv * T2{c.w0[0], c.w0[1]};
            T2 r1 = T2{c.a[2], c.a[3]} + mv * T2{c.w0[2], c.w0[3]};
            T2 r2 = T2{c.a[4], c.a[5]} + mv * T2{c.w1[0], c.w1[1]};
            T2 r3 = T2{c.a[6], c.a[7]} + mv * T2{c.w1[2], c.w1[3]};
            r0 += mw * T2{c.v0[0], c.v0[1]};
            r1 += mw * T2{c.v0[2], c.v0[3]};
            r2 += mw * T2{c.v1[0], c.v1[1]};
            r3 += mw * T2{c.v1[2], c.v1[3]};
            if (act) {
              rowp[jj + 0] = r0[0]; rowp[jj + 1] = r0[1]; rowp[jj + 2] = r1[0]; rowp[jj + 3] = r1[1];
              rowp[jj + 4] = r2[0]; rowp[jj + 5] = r2[1]; rowp[jj + 6] = r3[0]; rowp[jj + 7] = r3[1];
            }
          };
          int j = j0;
          for (; j + 2 * CH <= nfull; j += 2 * CH) {  // sixteen columns per trip: all loads before the first store
            Chunk ca, cb;
            load(j, ca);
            load(j + CH, cb);
            proc(j, ca);
            proc(j + CH, cb);
          }
          if (j < nfull) {
            Chunk ca;
            load(j, ca);
            proc(j, ca);
            j += CH;
          }
          for (j = (j < k + 1) ? k + 1 : j; j < n; ++j) {
            const T a1 = rowp[j] - (v * wsh[j] + w * vsh[j]);
            if (act) rowp[j] = a1;
          }
        }
      }
      if (below) rowp[k] = v;  // keep the reflector below the sub-diagonal
      if (lane == k) dv[k] = rowp[k];
      if (k == n - 2 && lane == n - 1) { dv[n - 1] = rowp[n - 1]; ev[n - 1] = T(0); }  // (before the barrier: wave 1 overwrites A with Q)
      if (TOP && k == n - 2) {   // d / e are final: the Sturm counts' copies (both waves read them after the barrier below)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const T el = (lane >= 1 && lane < n) ? ev[lane - 1] : T(0);
        dsel[lane] = lane < n ? dv[lane] : T(0);
        e2s[lane] = lane == 0 ? T(0) : fmax(el * el, Num<T>::tiny());
      }
      if ((k & 15) == 15 || k == n - 2) __syncthreads();
    }
    TTR_ESTAMP();
    TTR_ESTAMP();
    if constexpr (TOP) {
      // ---- 2t. this wave's eigenvalues and twisted factorisations
      select();
      TTR_ESTAMP();
      TTR_ESTAMP();
      __syncthreads();  // B1: all eigenvalues in lamv, wave 1 done with the reflectors in A
      TTR_ESTAMP();
      bool taken = qualifies();
      if (taken) {
        // ONE pass over each vector: written unnormalised (x_twist = 1) together with its norm; 1 / ||x|| is applied where wave 1
        // loads Z for the matrix cores.  (Round 4: a norm pass before the barrier and a scaled pass after it -- 14 k of a 32 x 32
        // problem's 207 k cycles, on both waves.)  A vector that overflowed is only known afterwards: checked behind B2.
        if (vlive) {
          const T nrm2 = vector_pass(IC2<1>{}, T(1));
          if (!(nrm2 < Num<T>::big_theta())) badf[0] = 1;
          invn[jv] = T(1) / sqrt(nrm2);
        }
        TTR_ESTAMP();
        __syncthreads();  // B2: Z complete
        taken = badf[0] == 0;
      }
      if (taken) {
        // epilogue of wave 0: sigma, rank, flag, zero columns
        if (lane < n) {
          T w = lane < rsel ? lamv[lane] * gmax : T(0);
          if (!(w > T(0))) w = T(0);
          p.sigma[bt * p.stride_sigma + lane] = sqrt(w);
        }
        if (lane == 0) {
          T w0 = lamv[0] * gmax;
          p.info[bt] = (sqrt(w0 > T(0) ? w0 : T(0)) < T(1e-13)) ? 0 : rsel;
          if (p.top_flat) p.top_flat[bt] = 1;
          if (p.sweeps) p.sweeps[bt] = 0;
        }
        T* __restrict__ Vo = p.V + bt * p.strideV;
        const int zc = n - rsel;
        for (int idx = lane; idx < n * zc; idx += kWave) {
          const int row = idx / zc, c = rsel + idx - row * zc;
          Vo[(int64_t)row * p.ldv + c] = T(0);
        }
        pad_tail(false, lane, kWave);
        TTR_ESTAMP();
        return;
      }
      if (lane == 0 && p.top_flat) p.top_flat[bt] = 0;
      __syncthreads();  // B1': both waves have read lamv (= rotation buffer 0 of the QL phase below)
    }
    // ---- 3. implicit-shift QL on (d, e); every sweep's rotations are published and replayed on Q's rows by wave 1.
    // The recurrence is one serial chain per matrix (the wave repeats it in all lanes): it runs at the issue rate of a
    // single wave.  d[k] / e[k] and the recorded rotation (c, s)[k-1] therefore live in REGISTERS, one index per lane:
    // operands are fetched with v_readlane (uniform index), results are merged with a lane-select -- no LDS access, no wait
    // and no exec-masked store inside the rotation loop; d[i], d[i+1] are carried from rotation to rotation.
    ereg = (lane < n) ? ev[lane] : T(0);
    dreg = (lane < n) ? dv[lane] : T(0);
    T creg = T(1), sreg = T(0);  // rotation i is kept in lane i + 1
    const T eps = Num<T>::eps();
    T an = fmax(fabs(dreg), fabs(ereg));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) an = fmax(an, __shfl_xor(an, off, 64));
    const T floor_abs = eps * an;
    int sweep = 0;
    for (int l = 0; l < n; ++l) {
      for (int iter = 0; iter < 64; ++iter) {
        // m = first index >= l with a negligible sub-diagonal (n-1 if none)
        const T dnext = __shfl_down(dreg, 1, 64);
        bool small = true;
        if (lane >= l && lane < n - 1) {
          const T el = fabs(ereg);
          small = (el <= eps * (fabs(dreg) + fabs(dnext))) || (el <= floor_abs);
        }
        unsigned long long mask = __ballot(small && lane >= l);
        const int m = __ffsll((long long)mask) - 1;  // lanes >= n-1 always report "small"
        if (m <= l) break;
        ++total_iter;
        const T dl = lane_get(dreg, l), el0 = lane_get(ereg, l);
        T g = ql_shift(dl, lane_get(dreg, l + 1), lane_get(dreg, m), el0);
        T r;
        T sn = T(1), cs = T(1), pp = T(0);
        bool underflow = false;
        int ilast = l;
#ifdef TTR_EIGH_STAMPS
        const long long tq0 = clock64();
        nrot_dbg += m - l;
#endif
        T d_ip1 = lane_get(dreg, m);  // d[i + 1] (not yet touched by this sweep)
        // Branch-free body, exit tests at the end.  Underflow (givens_norm returns false, rn = 0) needs no repair code: the body
        // itself stores e[i+1] = rn = 0 and d[i+1] = g = d[i+1] - p_old -- exactly tql2's "d[i+1] -= p, e[m] = 0, abandon the
        // sweep"; the garbage rotation it records is never replayed (ilast = i + 1) and the carried values die with the sweep.
        // With the repair on an early exit every carried value went through a copy at the back edge and the rotation took
        // three branches: 48 instructions; now ~36.
        int i = m - 1;
        bool zero;
        for (;;) {
          const T e_i = lane_get(ereg, i), d_i = lane_get(dreg, i);  // lanes <= i are untouched by this sweep so far
          const T f = sn * e_i, b = cs * e_i;
          T rn, rinv;
          const bool ok = givens_norm(f, g, rn, rinv);  // rn = sqrt(f^2 + g^2) (0 in the underflow case), rinv ~ 1 / rn
          const bool here = lane == i + 1;
          sn = f * rinv; cs = g * rinv;
          g = d_ip1 - pp;
          r = fma(cs, b + b, (d_i - g) * sn);   // (d_i - g) s + 2 c b with 2 b off the chain
          pp = sn * r;
          ereg = here ? rn : ereg;                         // e[i + 1]
          dreg = here ? (ok ? g + pp : g) : dreg;          // d[i + 1]   (underflow: d[i + 1] - p_old, nothing derived from rinv)
          creg = here ? cs : creg;
          sreg = here ? sn : sreg;
          g = fma(cs, r, -b);
          d_ip1 = d_i;
          zero = !ok;
          if (zero) break;
          if (i == l) break;
          --i;
        }
        if (zero) {
          if (lane == m) ereg = T(0);
          underflow = true;
          ilast = i + 1;
        } else {  // d_ip1 = d[l] here
          if (lane == l) { dreg = d_ip1 - pp; ereg = g; }
          if (lane == m) ereg = T(0);
        }
#ifdef TTR_EIGH_STAMPS
        qlrec_dbg += clock64() - tq0;
#endif
        {
          const int par = sweep & 1;
          T2* const buf = par ? cs1 : cs0;
          if (lane < n) buf[lane] = T2{creg, sreg};
          if (lane == 0) { meta[par * 4 + 0] = m; meta[par * 4 + 1] = ilast; meta[par * 4 + 2] = 0; }
          __syncthreads();
          ++sweep;
        }
      }
    }
    if (lane == 0) meta[(sweep & 1) * 4 + 2] = 1;
    __syncthreads();
    if (lane < n) dv[lane] = dreg;  // eigenvalues for the epilogue (wave 1 is past its last replay: buffer 1 is free)
  } else {
    // ---- 2. X = Q^T = H_{n-2} ... H_0 on the matrix cores, forward over the reflector blocks:  X <- (I - V_b T_b^T V_b^T) X
    {
      using MF = Mfma<T>;
      using Acc = typename MF::Acc;
      const int cl = lane & 15;
      T* const Ss = Sq;  // [16][17] V_b^T V_b; its unused lower-left blocks are the scratch of the T construction
      constexpr int SLD = 17;
      auto vb = [&](int row, int c) -> T {  // V[row][c], c = reflector index
        if (row >= n || c >= n - 1 || row <= c) return T(0);
        return row == c + 1 ? T(1) : A[row * ld + c];
      };
      Acc Z[NT][NT];
#pragma unroll
      for (int tm = 0; tm < NT; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r) Z[tm][tn][r] = (16 * tm + MF::row(lane, r) == 16 * tn + cl) ? T(1) : T(0);
      for (int b = 0; b < nblk; ++b) {
        __syncthreads();  // wave 0 has finished the block's reflectors (columns 16 b .. of A, tauv)
        const int c0 = 16 * b;
        {  // S = V_b^T V_b (rows <= c0 of V_b are zero: K starts at the block's first row tile)
          Acc s4[4] = {MF::zero(), MF::zero(), MF::zero(), MF::zero()};
          for (int tm = b; tm < NT; ++tm)
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
              const T a = vb(16 * tm + MF::row(lane, sI), c0 + cl);
              s4[sI] = MF::mma(a, a, s4[sI]);
            }
          const Acc sacc = (s4[0] + s4[1]) + (s4[2] + s4[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) Ss[MF::row(lane, r) * SLD + cl] = sacc[r];
        }
        for (int e = lane; e < 256; e += kWave) {
          const int i = e >> 4, k = e & 15;
          Tb[i * SLD + k] = (i == k && c0 + i < n - 1) ? tauv[c0 + i] : T(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int h = 1; h < 16; h <<= 1) {  // T12 = -T11 S12 T22 (see the QR kernel); X = S12 T22 goes to S's lower-left block
          const int hh = h * h;
          const int bq = lane / hh, rr2 = lane % hh, i = rr2 / h, jx = rr2 % h;
          const int o = bq * 2 * h;
          const bool act = lane < 8 * h;
          if (act) {
            T x = T(0);
#pragma unroll
            for (int k = 0; k < h; ++k) x += Ss[(o + i) * SLD + o + h + k] * Tb[(o + h + k) * SLD + o + h + jx];
            Ss[(o + h + i) * SLD + o + jx] = x;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (act) {
            T t = T(0);
#pragma unroll
            for (int k = 0; k < h; ++k) t += Tb[(o + i) * SLD + o + k] * Ss[(o + h + k) * SLD + o + jx];
            Tb[(o + i) * SLD + o + h + jx] = -t;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // W = V_b^T Z (16 x 64): A[i][k] = V_b[row k][i], B = Z tiles from the accumulators
        Acc W[NT];
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) W[tn] = MF::zero();
        for (int tm = b; tm < NT; ++tm)
#pragma unroll
          for (int sI = 0; sI < 4; ++sI) {
            const T a = vb(16 * tm + MF::row(lane, sI), c0 + cl);
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
              // (static register index for Z: tm is a runtime loop variable, so select the tile explicitly)
              T zb;
              if constexpr (NT == 4) zb = tm == 0 ? Z[0][tn][sI] : (tm == 1 ? Z[1][tn][sI] : (tm == 2 ? Z[2][tn][sI] : Z[3][tn][sI]));
              else zb = tm == 0 ? Z[0][tn][sI] : Z[1][tn][sI];
              W[tn] = MF::mma(a, zb, W[tn]);
            }
          }
        // W2 = T_b^T W
        Acc W2[NT];
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) W2[tn] = MF::zero();
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
          const T a = Tb[MF::row(lane, sI) * SLD + cl];
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) W2[tn] = MF::mma(a, W[tn][sI], W2[tn]);
        }
        // Z -= V_b W2 (row tiles above the block's first row are untouched: V_b is zero there)
#pragma unroll
        for (int tm = 0; tm < NT; ++tm) {
          if (tm < b) continue;  // wave-uniform
#pragma unroll
          for (int sI = 0; sI < 4; ++sI) {
            const T a = -vb(16 * tm + cl, c0 + MF::row(lane, sI));
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) Z[tm][tn] = MF::mma(a, W2[tn][sI], Z[tm][tn]);
          }
        }
      }
      TTR_ESTAMP1();
      if constexpr (TOP) {
        // ---- 2t. this wave's eigenvalues and twisted factorisations (Q^T stays in the accumulators Z)
        select();
        TTR_ESTAMP1();
        __syncthreads();  // B1
        bool taken = qualifies();
        if (taken) {
          if (vlive) {
            const T nrm2 = vector_pass(IC2<1>{}, T(1));
            if (!(nrm2 < Num<T>::big_theta())) badf[0] = 1;
            invn[jv] = T(1) / sqrt(nrm2);
          }
          TTR_ESTAMP1();
          __syncthreads();  // B2: Z complete in LDS
          TTR_ESTAMP1();
          taken = badf[0] == 0;
        }
        if (taken) {
          // ---- 4. V^T = P (Q Z)^T on the matrix cores
          T zr[NT][4][2];  // Z[16 tm + row(lane, s)][16 tv + cl]
#pragma unroll
          for (int tm = 0; tm < NT; ++tm)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int tv = 0; tv < 2; ++tv) {
                const int row = 16 * tm + MF::row(lane, s), col = 16 * tv + cl;
                zr[tm][s][tv] = (row < n && col < rsel) ? A[row * ZLD + col] * invn[col] : T(0);
              }
          Acc C[2][NT];  // (Q Z)^T: rows = vector, columns = matrix row
#pragma unroll
          for (int ti = 0; ti < NT; ++ti) {
#pragma unroll
            for (int tv = 0; tv < 2; ++tv) C[tv][ti] = MF::zero();
#pragma unroll
            for (int tm = 0; tm < NT; ++tm)
#pragma unroll
              for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tv = 0; tv < 2; ++tv) C[tv][ti] = MF::mma(zr[tm][s][tv], Z[tm][ti][s], C[tv][ti]);
          }
          TTR_ESTAMP1();
          Acc S[2][2];
#pragma unroll
          for (int tv = 0; tv < 2; ++tv)
#pragma unroll
            for (int tw = 0; tw < 2; ++tw) {
              S[tv][tw] = MF::zero();
#pragma unroll
              for (int tm = 0; tm < NT; ++tm)
#pragma unroll
                for (int s = 0; s < 4; ++s) S[tv][tw] = MF::mma(zr[tm][s][tv], zr[tm][s][tw], S[tv][tw]);
            }
          // Newton-Schulz: P = 1.5 I - 0.5 S (padding rows / columns: S = 0 there, P = 1.5 I -- harmless, C is zero there)
          auto newton = [&](const Acc (&Sx)[2][2], Acc (&Px)[2][2]) {
#pragma unroll
            for (int tv = 0; tv < 2; ++tv)
#pragma unroll
              for (int tw = 0; tw < 2; ++tw)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                  Px[tv][tw][rr] = T(-0.5) * Sx[tv][tw][rr] + ((16 * tv + MF::row(lane, rr) == 16 * tw + cl) ? T(1.5) : T(0));
          };
          // R = X Y for symmetric r x r matrices held in accumulators: X's registers are its A operand (X[m][k] = X[k][m]), Y's its B operand
          auto symmul = [&](const Acc (&X)[2][2], const Acc (&Y)[2][2], Acc (&R)[2][2]) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
              for (int tn = 0; tn < 2; ++tn) {
                R[tm][tn] = MF::zero();
#pragma unroll
                for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                  for (int s = 0; s < 4; ++s) R[tm][tn] = MF::mma(X[tk][tm][s], Y[tk][tn][s], R[tm][tn]);
              }
          };
          // Two steps in fp32, three in fp64: the mutual contamination theta of the twisted vectors of a pair at the admitted
          // distance (512 eps lambda_1) is ~ 1e-2 in EITHER precision, and a step squares it (theta -> 0.75 theta^2): 4e-9 after two
          // steps is below eps of fp32 only (measured in fp64 with two steps: 1.4e-10 on triples 600 eps apart).
          Acc P[2][2], T1[2][2], S1[2][2], P1[2][2], Pt[2][2];
          newton(S, P);
          symmul(S, P, T1);    // S P
          symmul(P, T1, S1);   // S1 = P S P
          newton(S1, P1);
          symmul(P1, P, Pt);   // Pt = P1 P  (all polynomials in S: symmetric, commuting)
          if constexpr (sizeof(T) == 8) {
            Acc S2[2][2], P2[2][2], Pu[2][2];
            symmul(S1, P1, T1);
            symmul(P1, T1, S2);  // S2 = P1 S1 P1
            newton(S2, P2);
            symmul(P2, Pt, Pu);  // P2 P1 P
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b2 = 0; b2 < 2; ++b2) Pt[a][b2] = Pu[a][b2];
          }
          TTR_ESTAMP1();
          T* __restrict__ Vo = p.V + bt * p.strideV;
#pragma unroll
          for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tv = 0; tv < 2; ++tv) {
              Acc X = MF::zero();
#pragma unroll
              for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int s = 0; s < 4; ++s) X = MF::mma(Pt[tk][tv][s], C[tk][ti][s], X);
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                const int v = 16 * tv + MF::row(lane, rr), i = 16 * ti + cl;
                if (v < rsel && i < n) Vo[(int64_t)i * p.ldv + v] = X[rr];
              }
            }
          TTR_ESTAMP1();
          return;
        }
        __syncthreads();  // B1'
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // Q[j][i] = X[i][j]: the transposed store (wave 0 no longer reads A: its last access is before the last block's barrier)
#pragma unroll
      for (int tm = 0; tm < NT; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * tm + MF::row(lane, r), j = 16 * tn + cl;
            if (i < n && j < n) A[j * ld + i] = Z[tm][tn][r];
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- 3'. replay wave 0's sweeps on the rows of Q (lane = row)
    for (int s = 0;; ++s) {
      __syncthreads();
      const int par = s & 1;
      if (__builtin_amdgcn_readfirstlane(meta[par * 4 + 2])) break;
      const int m = __builtin_amdgcn_readfirstlane(meta[par * 4 + 0]);
      const int ilast = __builtin_amdgcn_readfirstlane(meta[par * 4 + 1]);
      const T2* const buf = par ? cs1 : cs0;
      if (lane < n) {
        T hi = rowp[m];
        int i = m - 1;
        for (; i - (CH - 1) >= ilast; i -= CH) {
          T lo8[CH], o8[CH];
          T2 cs8[CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) { lo8[u] = rowp[i - u]; cs8[u] = buf[i - u + 1]; }
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const T c2 = cs8[u][0], s2 = cs8[u][1];
            const T cl = c2 * lo8[u], sl = s2 * lo8[u];  // off the chain
            o8[u] = fma(c2, hi, sl);
            hi = fma(-s2, hi, cl);                        // the chain: one FMA per rotation
          }
#pragma unroll
          for (int u = 0; u < CH; ++u) rowp[i - u + 1] = o8[u];
        }
        for (; i >= ilast; --i) {
          const T lo = rowp[i];
          const T2 cs2 = buf[i + 1];
          rowp[i + 1] = cs2[1] * lo + cs2[0] * hi;
          hi = cs2[0] * lo - cs2[1] * hi;
        }
        rowp[ilast] = hi;
      }
    }
  }
  __syncthreads();
  TTR_ESTAMP();
#ifdef TTR_EIGH_STAMPS
  if (dbg && bt == 0 && tid == 0) { dbg[dbgi++] = total_iter; dbg[dbgi++] = nrot_dbg; dbg[dbgi++] = qlrec_dbg; dbg[dbgi++] = 0; }
#endif

  // ---- 4. epilogue: un-scale, clamp / sqrt / sort, permuted write, rank rule (round.py:118-158)
  T* sig = vs;          // reuse
  T* sig_sorted = wsv;
  if (tid < n) {
    T w = dv[tid] * gmax;
    if (p.eig_mode == TTR_EIG_REF) { if (w < T(0)) w = T(1e-8); }
    else { if (!(w > T(0))) w = T(0); }
    sig[tid] = sqrt(w);
  }
  __syncthreads();
  if (tid < n) {
    const T si = sig[tid];
    int pos = 0;
    for (int j = 0; j < n; ++j) {
      const T sj = sig[j];
      pos += (sj > si) || (sj == si && j < tid);
    }
    sig_sorted[pos] = si;
    posv[tid] = pos;
  }
  __syncthreads();
  T* __restrict__ sout = p.sigma + bt * p.stride_sigma;
  if (p.eig_mode == TTR_EIG_MATCH_DIAG) {
    // Column order for block-Jacobi drivers: the eigenvector of the r-th largest eigenvalue goes to the
    // column holding the r-th largest diagonal entry of G, so that V -> I as G -> diagonal (no sorting
    // swaps; this is what makes the outer block iteration converge).
    int* colof = reinterpret_cast<int*>(sv);
    if (tid < kWave) cv[tid] = gdiag;
    __syncthreads();
    if (tid < n) {
      int dpos = 0;
      for (int j = 0; j < n; ++j) {
        const T dj = cv[j];
        dpos += (dj > gdiag) || (dj == gdiag && j < tid);
      }
      colof[dpos] = tid;
    }
    __syncthreads();
    if (tid < n) {
      const int c = colof[posv[tid]];
      posv[tid] = c;
      sout[c] = sig[tid];
    }
    __syncthreads();
  } else if (tid < n) {
    sout[tid] = sig_sorted[tid];
  }
  T* __restrict__ V = p.V + bt * p.strideV;
  for (int idx = tid; idx < n * n; idx += 2 * kWave) {
    const int row = idx / n, j = idx - row * n;
    V[(int64_t)row * p.ldv + posv[j]] = A[row * ld + j];
  }
  pad_tail(true, tid, 2 * kWave);
  if (tid == 0) {
    // (nf: the zero eigenvalues beyond a shrunk block count; zero guard: round.py:137-145)
    const T d2 = p.use_delta ? (T)(p.delta2_dev ? *p.delta2_dev : p.delta2) : T(0);
    p.info[bt] = rank_rule<T>(sig_sorted, n, nf, p.rmax, p.use_delta, d2, p.noise_c);
    if (p.sweeps) p.sweeps[bt] = total_iter;
    if constexpr (TOP) {
      // an item the top-r path declined: ttr_spectrum_flat's batch-mode test on the full decomposition's sigma, so that the flags
      // of a ttr_eigh_top launch are the pass-through flags of the bond (no separate launch for the test and for merging the two)
      if (p.top_flat) {
        const int kq = p.top_r < n ? (p.top_r < 1 ? 1 : p.top_r) : n;
        p.top_flat[bt] = (sig_sorted[0] > T(0) && sig_sorted[kq - 1] >= (T)p.top_thr * sig_sorted[0]) ? 2 : 0;
      }
    }
  }
}

static size_t eigh_tridiag_lds_bytes(size_t elem, int64_t n) {  // A, 3 pad, 408 scratch, d / e (66 each), tau, T_b, 8 control words, d / e^2 of the Sturm counts (2 x 64, 16-byte aligned)
  return (((size_t)n * (n + 1) + 3 + 408 + 66 * 2 + 64 + 16 * 17 + 8 + 4 + 128) * elem + 15) & ~size_t(15);
}


static size_t eigh_lds_bytes(size_t elem, int64_t n, bool ldsres) {
  const int np = (int)((n + 1) / 2), npad = (np + 7) & ~7;
  const size_t tab = (size_t)npad * (2 * sizeof(double) + 4 * sizeof(int) + 2 * elem);
  size_t bytes = tab + 16 * sizeof(int);
  bytes += (size_t)((n + 4) & ~3) * sizeof(int);  // dead-index flags
  bytes += (2 * ((n + 1) & ~1) + 2) * elem;
  if (ldsres) bytes += 2 * (size_t)n * (n + 1) * elem;  // G and V
  return (bytes + 15) & ~size_t(15);
}

int eigh_max_n_lds(int dtype) {
  // 160 KiB LDS per CU; G and V are (n x (n+1)) each
  const size_t elem = dtype == TTR_F64 ? 8 : 4;
  int n = 8;
  while (eigh_lds_bytes(elem, n + 1, true) <= 160 * 1024) ++n;
  return n;
}

int eigh_max_n(int dtype) { return dtype == TTR_F64 ? kMaxPairs : 2 * kMaxPairs; }

int64_t eigh_workspace_bytes(int dtype, int64_t n, int64_t batch) {
  if (n <= eigh_max_n_lds(dtype)) return 0;
  return batch * 2 * n * (n + 1) * (dtype == TTR_F64 ? 8 : 4);
}

int g_rank_noise_c = 1;  // ttr_debug_set_knob(TTR_KNOB_RANK_NOISE_FLOOR, c): see rank_rule (ttr_common.h); 1 = the reference's ranks (round 6: the default)
// ttr_debug_set_knob(TTR_KNOB_JACOBI_LIVE_WAVE, 1): pass 2 with ONE wave per matrix.  REFUTED by measurement (round 5,
// profiles/r05_decay_probe.txt): 1.25 instead of 0.80 ms per launch of 2048 matrices at n_live = 35 (eigh 14.3 instead of 10.0 ms
// per step), 0.38 instead of 0.26 ms at n_live = 18 -- a round of the parallel-order Jacobi is ~7000 scattered LDS accesses, not
// synchronisation: a quarter of the lanes take longer over them than the barriers of four waves cost.  Kept for the A/B.
int g_jacobi_live_wave = 0;
int g_eigh_big_occ = 0;   // ttr_debug_set_knob(TTR_KNOB_EIGH_BIG_OCC): waves per SIMD the 64-row top-r instance is built for (0 = 4)
int g_eigh_small = 2;   // ttr_debug_set_knob(TTR_KNOB_EIGH_SMALL): 0 = no separate 32-row launch, 1 = the 32-row instance, 2 (default) / 3 = and its
                        // large fp32 top-r launches at three / four waves per SIMD (A/B)
template <typename T>
static int eigh_typed(int dtype, int64_t n, int64_t batch, const void* G, int64_t ldg, int64_t strideG, int64_t gparts,
                      int64_t stride_gpart, void* V,
                      int64_t ldv, int64_t strideV, void* sigma, int64_t stride_sigma, int32_t* info, int eig_mode,
                      int use_delta, double delta2, int64_t rmax, int abs_floor, int32_t* sweeps, void* ws,
                      int64_t ws_bytes, hipStream_t stream, const double* delta2_dev = nullptr,
                      const int32_t* skip_items = nullptr, const void* sigma_in = nullptr, int64_t stride_sigma_in = 0) {
  const int64_t nmax = eigh_max_n(dtype);
  TTR_REQUIRE(n >= 1 && n <= nmax, TTR_E_UNSUPPORTED, "ttr_eigh_trunc: n = %lld outside [1, %lld]", (long long)n,
              (long long)nmax);
  EighArgs<T> p{};
  p.delta2_dev = delta2_dev;
  p.noise_c = g_rank_noise_c;
  p.skip_items = skip_items; p.sigma_in = (const T*)sigma_in; p.stride_sigma_in = stride_sigma_in;
  TTR_REQUIRE(!skip_items || (sigma_in && !(abs_floor == TTR_SOLVER_TRIDIAG && n <= 64)), TTR_E_INVALID,
              "ttr_eigh_trunc: skip_items needs sigma_in and a Jacobi solver");
  p.n = (int)n;
  p.G = (const T*)G; p.ldg = ldg; p.strideG = strideG;
  p.gparts = (int)gparts; p.stride_gpart = stride_gpart;
  p.V = (T*)V; p.ldv = ldv; p.strideV = strideV;
  p.sigma = (T*)sigma; p.stride_sigma = stride_sigma;
  p.info = info;
  p.eig_mode = eig_mode; p.use_delta = use_delta; p.delta2 = delta2; p.rmax = rmax;
  p.ws = (T*)ws;
  p.max_sweeps = sizeof(T) == 8 ? 40 : 30;
  p.abs_floor = abs_floor;
  p.sweeps = sweeps;
  if (abs_floor == TTR_SOLVER_TRIDIAG && n <= 64) {  // tridiagonal QL, two waves per matrix
    ProfScope prof(TTR_PROF_EIGH, stream);
    if (n == 64 && g_eigh_small && eig_mode != TTR_EIG_MATCH_DIAG) {   // zero-tail Gram matrices first, in the 32-row instance
      hipLaunchKernelGGL((eigh_tridiag_kernel<T, false, 32>), dim3((unsigned)batch), dim3(2 * kWave), eigh_tridiag_lds_bytes(sizeof(T), 32), stream, p);
      TTR_HIP_CHECK(hipGetLastError());
      p.top_pre = 1;
    }
    hipLaunchKernelGGL(eigh_tridiag_kernel<T>, dim3((unsigned)batch), dim3(2 * kWave), eigh_tridiag_lds_bytes(sizeof(T), n), stream, p);
    TTR_HIP_CHECK(hipGetLastError());
    return TTR_OK;
  }
  if (abs_floor == TTR_SOLVER_TRIDIAG) p.abs_floor = TTR_SOLVER_JACOBI_ABS;  // larger problems: Jacobi with the absolute floor
  const bool ldsres = n <= eigh_max_n_lds(dtype);
  if (!ldsres) {
    const int64_t need = eigh_workspace_bytes(dtype, n, batch);
    TTR_REQUIRE(ws && ws_bytes >= need, TTR_E_WORKSPACE, "ttr_eigh_trunc: workspace %lld < %lld bytes",
                (long long)ws_bytes, (long long)need);
  }
  const size_t lds = eigh_lds_bytes(sizeof(T), n, ldsres);
  ProfScope prof(TTR_PROF_EIGH, stream);
  if (ldsres && abs_floor == TTR_SOLVER_JACOBI_LIVE && n <= 64 && g_jacobi_live_wave) {
    // (diagnostic variant, see g_jacobi_live_wave: ONE wave per matrix -- measured slower than the four-wave kernel)
    auto kern = eigh_jacobi_kernel<T, true, kWave>;
    if (lds > 64 * 1024)
      TTR_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(kWave), lds, stream, p);
  } else if (ldsres) {
    auto kern = eigh_jacobi_kernel<T, true>;
    if (lds > 64 * 1024)
      TTR_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(kThreads), lds, stream, p);
  } else {
    auto kern = eigh_jacobi_kernel<T, false>;
    if (lds > 64 * 1024)
      TTR_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(kThreads), lds, stream, p);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int eigh_dispatch(int dtype, int64_t n, int64_t batch, const void* G, int64_t ldg, int64_t strideG, int64_t gparts,
                  int64_t stride_gpart, void* V,
                  int64_t ldv, int64_t strideV, void* sigma, int64_t stride_sigma, int32_t* info, int eig_mode,
                  int use_delta, double delta2, int64_t rmax, int abs_floor, int32_t* sweeps, void* ws,
                  int64_t ws_bytes, hipStream_t stream, const double* delta2_dev, const int32_t* skip_items,
                  const void* sigma_in, int64_t stride_sigma_in) {
  if (dtype == TTR_F32)
    return eigh_typed<float>(dtype, n, batch, G, ldg, strideG, gparts, stride_gpart, V, ldv, strideV, sigma, stride_sigma, info, eig_mode,
                             use_delta, delta2, rmax, abs_floor, sweeps, ws, ws_bytes, stream, delta2_dev, skip_items, sigma_in,
                             stride_sigma_in);
  return eigh_typed<double>(dtype, n, batch, G, ldg, strideG, gparts, stride_gpart, V, ldv, strideV, sigma, stride_sigma, info, eig_mode,
                            use_delta, delta2, rmax, abs_floor, sweeps, ws, ws_bytes, stream, delta2_dev, skip_items, sigma_in,
                            stride_sigma_in);
}

// pass 1 of a batch-mode bond, n <= 64: the r largest eigenpairs (flat[b] = 1) or the full QL decomposition (flat[b] = 0) per item
#ifdef TTR_EIGH_STAMPS
static void* g_eigh_stamps = nullptr;   // diagnostics build: cycle stamps of matrix 0 of the next ttr_eigh_top launches (>= 64 int64)
extern "C" void ttr_debug_set_eigh_stamps(void* p) { g_eigh_stamps = p; }
#endif
template <typename T>
static int eigh_top_typed(int64_t n, int64_t batch, const void* G, int64_t ldg, int64_t strideG, int64_t gparts, int64_t stride_gpart,
                          void* V, int64_t ldv, int64_t strideV, void* sigma, int64_t stride_sigma, int32_t* info, int64_t r, double thr,
                          int32_t* flat, hipStream_t stream, int need_all) {
  EighArgs<T> p{};
  p.top_need_all = need_all;
  p.n = (int)n;
  p.G = (const T*)G; p.ldg = ldg; p.strideG = strideG; p.gparts = (int)gparts; p.stride_gpart = stride_gpart;
  p.V = (T*)V; p.ldv = ldv; p.strideV = strideV;
  p.sigma = (T*)sigma; p.stride_sigma = stride_sigma;
  p.info = info; p.rmax = n; p.top_r = (int)r; p.top_flat = flat; p.top_thr = thr;
  p.eig_mode = TTR_EIG_RAW;
#ifdef TTR_EIGH_STAMPS
  p.ws = (T*)g_eigh_stamps;
#endif
  ProfScope prof(TTR_PROF_EIGH, stream);
  if (n == 64 && g_eigh_small) {   // zero-tail items first, in the 32-row instance (see the kernel)
    // Large fp32 launches go through a build capped at 168 VGPRs (TTR_KNOB_EIGH_SMALL = 2, the default; 3: 128): the instance needs
    // 210 registers, i.e. two waves per SIMD = FOUR matrices per CU at a time, each of them latency-bound -- three waves per SIMD
    // with 44 spilled registers (four: 124) is the faster trade from 1024 matrices per launch on (B = 4096: eigh 3.75 -> 3.13 ms
    // per step of event time, the step 24.91 -> 24.77 ms, bit-identical results: profiles/r05_eigh_occ_ab.txt); a single matrix
    // keeps the spill-free build.
    if (sizeof(T) == 4 && g_eigh_small == 2 && batch >= 1024)
      hipLaunchKernelGGL((eigh_tridiag_kernel<T, true, 32, 3>), dim3((unsigned)batch), dim3(2 * kWave), eigh_tridiag_lds_bytes(sizeof(T), 32), stream, p);
    else if (sizeof(T) == 4 && g_eigh_small == 3 && batch >= 1024)
      hipLaunchKernelGGL((eigh_tridiag_kernel<T, true, 32, 4>), dim3((unsigned)batch), dim3(2 * kWave), eigh_tridiag_lds_bytes(sizeof(T), 32), stream, p);
    else
      hipLaunchKernelGGL((eigh_tridiag_kernel<T, true, 32>), dim3((unsigned)batch), dim3(2 * kWave), eigh_tridiag_lds_bytes(sizeof(T), 32), stream, p);
    TTR_HIP_CHECK(hipGetLastError());
    p.top_pre = 1;
  }
  // (round 6, TTR_KNOB_EIGH_BIG_OCC: the 64-row instance needs ~230 registers and carries 106 spilled ones under the 128-register cap of
  // four waves per SIMD; 3 / 2 = builds at three / two waves per SIMD for large fp32 launches, A/B)
  if (sizeof(T) == 4 && g_eigh_big_occ == 3 && batch >= 1024)
    hipLaunchKernelGGL((eigh_tridiag_kernel<T, true, 64, 3>), dim3((unsigned)batch), dim3(2 * kWave), eigh_tridiag_lds_bytes(sizeof(T), n), stream, p);
  else if (sizeof(T) == 4 && g_eigh_big_occ == 2 && batch >= 1024)
    hipLaunchKernelGGL((eigh_tridiag_kernel<T, true, 64, 2>), dim3((unsigned)batch), dim3(2 * kWave), eigh_tridiag_lds_bytes(sizeof(T), n), stream, p);
  else
    hipLaunchKernelGGL((eigh_tridiag_kernel<T, true>), dim3((unsigned)batch), dim3(2 * kWave), eigh_tridiag_lds_bytes(sizeof(T), n), stream, p);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}
int eigh_top_dispatch(int dtype, int64_t n, int64_t batch, const void* G, int64_t ldg, int64_t strideG, int64_t gparts,
                      int64_t stride_gpart, void* V, int64_t ldv, int64_t strideV, void* sigma, int64_t stride_sigma, int32_t* info,
                      int64_t r, double thr, int32_t* flat, hipStream_t stream, int need_all) {
  if (dtype == TTR_F32)
    return eigh_top_typed<float>(n, batch, G, ldg, strideG, gparts, stride_gpart, V, ldv, strideV, sigma, stride_sigma, info, r, thr, flat, stream, need_all);
  return eigh_top_typed<double>(n, batch, G, ldg, strideG, gparts, stride_gpart, V, ldv, strideV, sigma, stride_sigma, info, r, thr, flat, stream, need_all);
}

int g_bj_inner_sweeps = 1;  // ttr_debug_set_knob(TTR_KNOB_BJ_INNER_SWEEPS); measured on C3's share: 1 -> 72 ms, 2 -> 82, 3 -> 90, until converged -> 97

// Pair problems of one block-Jacobi round (ttr_bj_solve): `items * npairs` LDS-resident Jacobi problems of size w = 2 b <= 64
// gathered from the items' n x n matrices through the device pair table; eigenvectors with the diagonal-matched column
// order (W -> I as the pair block -> diagonal), W[(item * npairs + pair)] = w x w contiguous.  `scratch`: w + 1 elements per
// problem (sigma, rank -- unused by the driver).  `skip_flag` / `rot_count`: device words of the driver's control block.
template <typename T>
static int eigh_pairs_typed(int64_t b, int64_t npairs, int64_t items, const T* G, int64_t ldg, int64_t strideG,
                            const int32_t* pair_tab, T* W, T* scratch, const int32_t* skip_flag, int32_t* rot_count,
                            hipStream_t stream) {
  const int64_t w = 2 * b, nprob = items * npairs;
  EighArgs<T> p{};
  p.n = (int)w;
  p.G = G; p.ldg = ldg; p.strideG = strideG; p.gparts = 1;
  p.V = W; p.ldv = w; p.strideV = w * w;
  p.sigma = scratch; p.stride_sigma = w;
  p.info = reinterpret_cast<int32_t*>(scratch + nprob * w);
  p.eig_mode = TTR_EIG_MATCH_DIAG; p.use_delta = 0; p.rmax = w;
  p.max_sweeps = g_bj_inner_sweeps > 0 ? g_bj_inner_sweeps : (sizeof(T) == 8 ? 40 : 30);
  p.abs_floor = TTR_SOLVER_JACOBI_ABS;
  p.pair_tab = pair_tab; p.pair_b = (int)b; p.pairs_per_item = (int)npairs;
  p.skip_flag = skip_flag; p.rot_count = rot_count;
  const size_t lds = eigh_lds_bytes(sizeof(T), w, true);
  auto kern = eigh_jacobi_kernel<T, true>;
  if (lds > 64 * 1024)
    TTR_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope prof(TTR_PROF_EIGH, stream);
  hipLaunchKernelGGL(kern, dim3((unsigned)nprob), dim3(kThreads), lds, stream, p);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int eigh_pairs_dispatch(int dtype, int64_t b, int64_t npairs, int64_t items, const void* G, int64_t ldg, int64_t strideG,
                        const int32_t* pair_tab, void* W, void* scratch, const int32_t* skip_flag, int32_t* rot_count,
                        hipStream_t stream) {
  if (dtype == TTR_F32)
    return eigh_pairs_typed<float>(b, npairs, items, (const float*)G, ldg, strideG, pair_tab, (float*)W, (float*)scratch,
                                   skip_flag, rot_count, stream);
  return eigh_pairs_typed<double>(b, npairs, items, (const double*)G, ldg, strideG, pair_tab, (double*)W, (double*)scratch,
                                  skip_flag, rot_count, stream);
}

}  // namespace ttr

