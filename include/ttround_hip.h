/*
 * ttround_hip.h -- C ABI of libttround_hip.so, the MI355X (gfx950) kernels behind
 * tntorch's TT orthogonalisation / rounding hot path.
 *
 * The reference (rballester/tntorch) has NO native layer: the hot path reaches
 * LAPACK/BLAS through torch operators.  Each entry point below therefore replaces
 * one *operator call site* of the reference (file:line given per function), and
 * is what a maintainer would bind with ctypes from tntorch/round.py and
 * tntorch/tensor.py.
 *
 * Conventions
 *   - all matrix pointers are DEVICE pointers; matrices are row-major with an
 *     explicit leading dimension (elements) and a batch stride (elements);
 *   - `dtype` is TTR_F32 or TTR_F64; scalars that select truncation (delta^2)
 *     are passed as double;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream);
 *     every call only enqueues work on that stream, never synchronises, never
 *     allocates: workspaces are caller-provided (sizes from *_workspace_bytes);
 *   - return value: 0 on success, <0 on error (TTR_E_*); ttr_last_error() gives
 *     a thread-local message.
 */
#ifndef TTROUND_HIP_H
#define TTROUND_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTR_F32 0
#define TTR_F64 1

#define TTR_OK 0
#define TTR_E_INVALID (-1)     /* bad argument */
#define TTR_E_UNSUPPORTED (-2) /* shape outside the kernels' envelope */
#define TTR_E_HIP (-3)         /* HIP runtime error */
#define TTR_E_WORKSPACE (-4)   /* workspace too small */

/* scale modes for ttr_gemm epilogues */
#define TTR_SCALE_NONE 0
#define TTR_SCALE_MUL 1
#define TTR_SCALE_DIV 2 /* x / s, and 0 where |s| is below the smallest normal */

/* eigenvalue post-processing modes for ttr_eigh_trunc */
#define TTR_EIG_RAW 0   /* sigma = sqrt(max(w, 0)) */
#define TTR_EIG_REF 1   /* round.py:118-119: w < 0 -> 1e-8 before the sqrt */
#define TTR_EIG_MATCH_DIAG 2 /* like RAW, but NOT sorted: V -> I as G -> diagonal.  Tridiagonal solver: the
                                eigenvector of the r-th largest eigenvalue is written to the column holding
                                the r-th largest diagonal entry of G; Jacobi: eigenpair i stays in column i
                                (rotation angles <= pi/4 never swap).  Building block of the block-Jacobi
                                driver for n above the single-workgroup limit. */

/* ABI version of this header: bumped whenever an exported signature changes (round 2 inserted `gparts` / `stride_gpart` into
   ttr_eigh_trunc = 2; round 3 additions = 3 ... 7, the last one ttr_eigh_top; round 4: 8 = rows32 / skip_zero_rows, 9 = ttr_carry_rows32;
   round 5: 10 = ttr_round_tt, the whole sweep behind one call, + TTR_KNOB_RANK_NOISE_FLOOR; 11 = ttr_qr_factor_expo / ttr_qr_factor_pushed_expo).  ttr_version() returns the value the library was built with; the Python
   binding refuses to use a library whose version differs (a stale .so would take misaligned arguments silently). */
#define TTR_ABI_VERSION 11
int ttr_version(void);
const char* ttr_last_error(void);

/* Limits of the LDS/register-resident kernels (queried by the host shim). */
int ttr_qr_max_cols(int dtype);        /* widest panel ttr_qr factors (columns) */
int ttr_eigh_max_n_lds(int dtype);     /* largest n solved out of LDS; above it the solver works out of L2/HBM */
int ttr_eigh_max_n(int dtype);         /* largest n ttr_eigh_trunc accepts (4096 fp32 / 2048 fp64) */

/*
 * C[b] = scale( op(A[b]) * op(B[b]) ),  op(A) is M x K, op(B) is K x N.
 * Replaces: `R @ right_unfolding(core)` tensor.py:1826-1832 (push right),
 *           `M @ M^T` / `M^T @ M` round.py:104-109 (Gram),
 *           `left^T @ M` with the 1/sigma row scaling round.py:163-172 (projection),
 *           `einsum("ijk,kl")` tensor.py:2074-2083 (push left, with the sigma column scaling
 *           of round.py:169-172 fused), `M @ left` round.py:175-181.
 * transA/transB: 0 -> the stored matrix is op(X); 1 -> the stored matrix is op(X)^T.
 * rowscale (length M per batch item) / colscale (length N per batch item) may be NULL.
 * MFMA (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64) on LDS-staged tiles.
 * `workspace` (may be NULL) enables split-K for few-tile / very-long-K products
 * (Gram matrices of tall dense unfoldings); size from ttr_gemm_workspace_bytes.
 */
int64_t ttr_gemm_workspace_bytes(int dtype, int64_t M, int64_t N, int64_t K, int64_t batch);
int ttr_gemm(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
             const void* A, int64_t lda, int64_t strideA,
             const void* B, int64_t ldb, int64_t strideB,
             void* C, int64_t ldc, int64_t strideC,
             const void* rowscale, int64_t stride_rs, int rowscale_mode,
             const void* colscale, int64_t stride_cs, int colscale_mode,
             int64_t batch, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * C[b] <- beta * C[b] + alpha * op(A[b]) * op(B[b])   (same kernel; C is read only when beta != 0).
 * Building block of the routines that extend the envelope of the kernels above: block Gram-Schmidt
 * projections  W <- W - Q (Q^T W)  of the QR for n > ttr_qr_max_cols (torch.linalg.qr, tensor.py:1816/1853)
 * and the Newton-Schulz step  V <- 1.5 V - 0.5 V (V^T V)  of the large-n eigensolver (round.py:115).
 */
int ttr_gemm_axpby(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
                   const void* A, int64_t lda, int64_t strideA,
                   const void* B, int64_t ldb, int64_t strideB,
                   void* C, int64_t ldc, int64_t strideC,
                   double alpha, double beta,
                   int64_t batch, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Reduced Householder QR of batch tall (or square/wide) matrices: A[b] (m x n) = Q[b] (m x k) R[b] (k x n),
 * k = min(m, n), LAPACK sign convention (geqrf: beta = -sign(alpha)*norm), R upper triangular/trapezoidal.
 * Replaces: torch.linalg.qr at tensor.py:1816 (left_orthogonalize) and tensor.py:1853/1859
 * (right_orthogonalize, on the transposed unfolding).
 * Communication-avoiding TSQR: 256-row blocks held as MFMA accumulator tiles and factored with the blocked
 * compact-WY algorithm (16-column panels, trailing updates on the matrix cores), R factors reduced over a
 * 4-ary tree, Q formed by walking the tree back (pure MFMA).  n <= ttr_qr_max_cols(dtype).
 */
int64_t ttr_qr_workspace_bytes(int dtype, int64_t m, int64_t n, int64_t batch);
int ttr_qr(int dtype, int64_t m, int64_t n, int64_t batch,
           const void* A, int64_t lda, int64_t strideA,
           void* Q, int64_t ldq, int64_t strideQ,
           void* R, int64_t ldr, int64_t strideR,
           void* workspace, int64_t workspace_bytes, void* stream);
/*
 * The same factorisation with TRANSPOSED addressing -- tensor.py:1853-1863 factors the transpose of the right unfolding
 * (`torch.linalg.qr(unfolding.T)`, torch has no RQ) and transposes Q back (1866-1878).  `At` is the n x m row-major
 * matrix whose TRANSPOSE A (m x n) is factored; `Qt` receives Q^T (k x m row-major, ldqt >= m); R as for ttr_qr.  The
 * kernels read / write through (row, column) strides, nothing is copied.
 */
int ttr_qr_t(int dtype, int64_t m, int64_t n, int64_t batch,
             const void* At, int64_t ldat, int64_t strideAt,
             void* Qt, int64_t ldqt, int64_t strideQt,
             void* R, int64_t ldr, int64_t strideR,
             void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The two halves of ttr_qr, for callers that fuse work into the formation of Q:
 *   ttr_qr_factor  factors A (TSQR tree of blocked compact-WY Householder blocks on MFMA), writes R (k x n) and
 *                  leaves the reflectors / T factors in `workspace` (which must stay alive and untouched);
 *   ttr_qr_apply   forms Out[b] (m x kcols) = Q[b] * C[b], C (k x kcols, kcols <= k) -- C == NULL means the
 *                  identity, i.e. the first kcols columns of Q.  round_tt uses it to produce core * (U sigma)
 *                  directly (the "push left" einsum of tensor.py:2081-2083) without ever materialising Q.
 */
int ttr_qr_factor(int dtype, int64_t m, int64_t n, int64_t batch,
                  const void* A, int64_t lda, int64_t strideA,
                  void* R, int64_t ldr, int64_t strideR,
                  void* workspace, int64_t workspace_bytes, void* stream);
int ttr_qr_apply(int dtype, int64_t m, int64_t n, int64_t batch,
                 void* workspace, int64_t workspace_bytes,
                 const void* C, int64_t ldc, int64_t strideC, int64_t kcols,
                 void* Out, int64_t ldo, int64_t strideO, void* stream);
/*
 * ttr_qr_factor with the sweep's power-of-two normalisation folded in (ABI 11; fp32 only): R comes back as R 2^-e, e the binary
 * exponent of the largest entry of the TOP block of the TSQR tree (max |entry| of what is returned times 2^e lies within a factor
 * sqrt(rows) of 1: O(1), exact scaling), and e is ADDED to expo_acc[item] (device int32 [batch]).  The left-to-right loop of a
 * rounding (tensor.py:1905-1906) multiplies the norms of all cores into the last R: 64^8 randn cores reach 2^72 in fp32, sums and
 * products more; the sweep therefore keeps every R at O(1) and gives the exponents back to the first core at the end (exact).
 * Rounds 1 - 4 did that with one ttr_pow2_normalize launch per core; the factor kernel already scales its block by a power of two
 * (its fp32 range guard) and here simply does not scale R back.  Reflectors and T factors are scale invariant: ttr_qr_apply on
 * the same workspace is unchanged.  ttr_qr_factor_pushed_expo: the same for the fused push.
 */
int ttr_qr_factor_expo(int dtype, int64_t m, int64_t n, int64_t batch,
                       const void* A, int64_t lda, int64_t strideA,
                       void* R, int64_t ldr, int64_t strideR,
                       void* workspace, int64_t workspace_bytes, int32_t* expo_acc, void* stream);

/*
 * Fused "push right + QR" of the left-to-right sweep (tensor.py:1823-1832 followed by tensor.py:1816 of the next
 * core): factors the (k*I) x n left unfolding of  P[kk, i, c] = sum_r0 Rm[kk, r0] * core[r0, i, c]  without ever
 * writing P.  Rm is the R factor of the previous core (k x Rin, k, Rin <= 64), `core` the next core, contiguous
 * [Rin, I, n].  The TSQR row blocks are chosen as {(kk, i): i = 4b .. 4b+3}, so every wave forms its own 64 x n
 * slice Rm * core[:, i, :] on the matrix cores straight into the accumulator tiles it then factors.
 * ttr_qr_apply_pushed is the matching ttr_qr_apply (Out rows in the natural order kk*I + i).  Needs k*I >= n.
 */
int64_t ttr_qr_pushed_workspace_bytes(int dtype, int64_t I, int64_t n, int64_t batch);
int ttr_qr_factor_pushed(int dtype, int64_t k, int64_t Rin, int64_t I, int64_t n, int64_t batch,
                         const void* Rm, int64_t ldrm, int64_t strideRm,
                         const void* core, int64_t stride_core,
                         void* R, int64_t ldr, int64_t strideR,
                         void* workspace, int64_t workspace_bytes, void* stream);
int ttr_qr_factor_pushed_expo(int dtype, int64_t k, int64_t Rin, int64_t I, int64_t n, int64_t batch,
                              const void* Rm, int64_t ldrm, int64_t strideRm,
                              const void* core, int64_t stride_core,
                              void* R, int64_t ldr, int64_t strideR,
                              void* workspace, int64_t workspace_bytes, int32_t* expo_acc, void* stream);
/*
 * The same fused push + factorisation for the middle core of a TT SUM a + b (tensor.py:445-668): the next core is the
 * block-diagonal blockdiag(a_core [ra][I][ca], b_core [rb][I][cb]) of tensor.py's `__add__`, which is never
 * materialised -- the kernel reads the two blocks where they lie (column tiles inside one block only walk that block's
 * rows: half the multiply-adds of the padded core for equal ranks).  Rm is k x (ra + rb); workspace size and the
 * implicit Q (ttr_qr_apply_pushed) are those of ttr_qr_factor_pushed with n = ca + cb.  This is the "add, then round"
 * fusion of tools.reduce (tools.py:460-512).
 */
int ttr_qr_factor_pushed_sum(int dtype, int64_t k, int64_t I, int64_t batch,
                             const void* Rm, int64_t ldrm, int64_t strideRm,
                             const void* core_a, int64_t ra, int64_t ca, int64_t stride_a,
                             const void* core_b, int64_t rb, int64_t cb, int64_t stride_b,
                             void* R, int64_t ldr, int64_t strideR,
                             void* workspace, int64_t workspace_bytes, void* stream);
int ttr_qr_apply_pushed(int dtype, int64_t k, int64_t I, int64_t n, int64_t batch,
                        void* workspace, int64_t workspace_bytes,
                        const void* C, int64_t ldc, int64_t strideC, int64_t kcols,
                        void* Out, int64_t ldo, int64_t strideO, int skip_zero_rows, void* stream);
/* `skip_zero_rows` != 0: for the items whose factorisation packed its rows (TTR_KNOB_QR_PACK; flags at
 * ttr_qr_pushed_flag_offset) the exactly-zero rows kk >= 32 of Out are NOT written -- a third of the kernel's HBM traffic; the
 * caller then reads Out only through kernels that take the item's flag (`rows32` of ttr_rowgram / ttr_rotgram / ttr_project). */
/*
 * ttr_qr_apply_pushed that ALSO emits the row Gram matrix of its output, G = M M^T of the k x (I*kcols) right unfolding M of
 * Out (round.py:104-109: what the truncation of the next bond computes first), accumulated from the output tiles while
 * they are still in registers -- the right-to-left sweep reads the new carry twice instead of three times.
 * G: [batch][parts][k][k] contiguous split partials (both triangles), parts = ttr_qr_apply_pushed_gram_parts(...); the
 * eigensolver sums them on load (ttr_eigh_trunc's gparts).  parts == 0: this shape is not covered (fp32, k = 64, kcols = 32,
 * I a multiple of 8 are), call ttr_qr_apply_pushed + ttr_rowgram instead.  ldo must equal kcols.
 * Requires TTR_KNOB_QR_PACK == 0 from before the factorisation until after this call (the epilogue walks the unpacked row map
 * of the level-0 blocks); with the knob != 0 the call returns TTR_E_UNSUPPORTED instead of a Gram matrix with unwritten partials.
 */
int64_t ttr_qr_apply_pushed_gram_parts(int dtype, int64_t k, int64_t I, int64_t n, int64_t kcols);
int ttr_qr_apply_pushed_gram(int dtype, int64_t k, int64_t I, int64_t n, int64_t batch,
                             void* workspace, int64_t workspace_bytes,
                             const void* C, int64_t ldc, int64_t strideC, int64_t kcols,
                             void* Out, int64_t ldo, int64_t strideO, void* G, void* stream);

/*
 * Symmetric eigen-decomposition + the rank rule of truncated_svd, batched, on device.
 * Input  G[b]  (n x n symmetric, e.g. a Gram matrix).
 * Output V[b]  (n x n): eigenvectors as COLUMNS, sorted by DEcreasing sigma,
 *        sigma[b] (n): sqrt of the (clamped) eigenvalues, decreasing,
 *        info[b]  (int32): the selected rank r >= 1, or 0 if sigma_max < 1e-13 (round.py:137-145 zero guard).
 * Rank rule (round.py:147-158): drop the longest tail with sum(sigma^2) <= delta2, then
 * r = max(1, min(rmax, n - tail)); with use_delta == 0 (batch mode, round.py:149-150) r = max(1, min(rmax, n)).
 * Replaces: torch.linalg.eigh round.py:115, the clamp/sqrt/argsort of round.py:118-135 and the rank
 * selection round.py:147-158 (and, in the two-pass 'svd' algorithm, torch.linalg.svd round.py:96).
 * Parallel cyclic two-sided Jacobi; out of LDS for n <= ttr_eigh_max_n_lds(dtype), else out of `workspace`.
 * Rotations are skipped when |G_pq| <= sqrt(n)*eps*sqrt(G_pp*G_qq) (relative criterion: high relative
 * accuracy on graded, accurately formed Gram matrices -- pass 2 of the 'svd' algorithm).  abs_floor = 1
 * additionally skips |G_pq| <= sqrt(n)*eps*max|G_ii| (for a plain Gram matrix, whose entries are only
 * accurate to eps*||G||).  abs_floor = 2 selects, for n <= 64, the tridiagonal solver instead (Householder
 * reduction + implicit-shift QL, two waves per matrix, absolute accuracy O(eps*||G||) like LAPACK steqr):
 * ~10x fewer flops, used for the first pass / 'eig'; larger n falls back to Jacobi with abs_floor = 1.
 * abs_floor = TTR_SOLVER_JACOBI_LIVE (pass 2 of the two-pass 'svd' truncation): purely relative rotation test among
 * the LIVE indices; indices with G_ii <= (n eps)^2 max G_ii (the numerical null space of the input) are frozen.  Every
 * live direction gets the accuracy of a backward-stable SVD (LAPACK gesdd class, round.py:96) whatever its sigma.
 * `sweeps` (optional, [batch]) receives the number of sweeps (Jacobi) / QL iterations used.
 * eig_mode = TTR_EIG_MATCH_DIAG: see above (sigma[b] then follows V's column order; info as usual).
 * TTR_EIG_RAW / TTR_EIG_REF clamp negative eigenvalues: G is taken to be positive semi-definite up to rounding (a Gram matrix).  The
 * tridiagonal solver uses that for a 64 x 64 matrix whose diagonal is exactly zero from index 32 on (zero rows / columns: the
 * carry of a bond whose QR packed its rows): it is solved as its leading 32 x 32 block, V[b] = blockdiag(V11, I), sigma[b][32:] = 0.
 * PRECONDITION of RAW / REF therefore: G is a Gram matrix (PSD up to rounding: G_ii = 0 implies a zero row).  A general symmetric
 * matrix -- e.g. [[0, 1], [1, 0]] padded to 64 x 64 -- must be passed with TTR_EIG_MATCH_DIAG, which neither clamps nor shrinks.
 * The input may be given as `gparts` partial matrices (split-K partials of a Gram kernel, `stride_gpart` elements
 * apart): G[b] = sum_p G[b * strideG + p * stride_gpart + ...]; gparts = 1 for a plain matrix.
 */
#define TTR_SOLVER_JACOBI_REL 0
#define TTR_SOLVER_JACOBI_ABS 1
#define TTR_SOLVER_TRIDIAG 2
#define TTR_SOLVER_JACOBI_LIVE 3
int64_t ttr_eigh_workspace_bytes(int dtype, int64_t n, int64_t batch);
int ttr_eigh_trunc(int dtype, int64_t n, int64_t batch,
                   const void* G, int64_t ldg, int64_t strideG, int64_t gparts, int64_t stride_gpart,
                   void* V, int64_t ldv, int64_t strideV,
                   void* sigma, int64_t stride_sigma,
                   int32_t* info,
                   int eig_mode, int use_delta, double delta2, const double* delta2_dev, int64_t rmax,
                   int abs_floor, int32_t* sweeps,
                   const int32_t* skip_items, const void* sigma_in, int64_t stride_sigma_in,
                   void* workspace, int64_t workspace_bytes, void* stream);
/* `skip_items` / `sigma_in` (optional; Jacobi solvers): pass-through items of the two-pass truncation, see
 * ttr_spectrum_flat below.
 * `delta2_dev` (optional, device pointer to ONE double): the bound delta^2 of the rank rule taken from device memory instead
 * of `delta2` -- tensor.py:2039-2051 computes delta from the norm of the last core and reads it back (`.item()`); an
 * eps-mode sweep that keeps it on the device enqueues every bond without a host synchronisation. */

/*
 * Pass 1 of a batch-mode bond with 40 <= n <= 64 rows and a rank cap r <= 32, r < n (ttr_eigh_top_ok): round.py:96, 147-158 with
 * the rank fixed to rmax never looks below sigma_r.  Two waves per matrix: Householder tridiagonalisation + forward formation of Q as
 * in ttr_eigh_trunc's tridiagonal solver, then -- instead of the QL iteration over the whole spectrum -- Sturm-count multisection
 * for lambda_1..r, one twisted factorisation per eigenvalue (MRRR getvec), Z^T Z -> two Newton-Schulz steps for orthonormality,
 * V[:, :r] = Q Z (Z^T Z)^{-1/2} on MFMA.
 * The kernel decides PER ITEM whether that result can be trusted: the kept spectrum must be flat (sigma_r >= thr sigma_1 > 0, the
 * criterion of ttr_spectrum_flat) and free of close pairs (neighbouring kept eigenvalues >= 512 eps lambda_1 apart: inverse
 * iteration on isolated eigenvalues needs no reorthogonalisation beyond the Newton-Schulz polish).  flat[b] = 1: V[b][:, :r],
 * sigma[b][:r] = sqrt(lambda) (descending), the rest of V[b] / sigma[b] zero, info[b] = r.  flat[b] = 0 or 2: the item fell through to
 * the QL phase of the same launch and carries the full decomposition, exactly what ttr_eigh_trunc(eig_mode = TTR_EIG_RAW,
 * abs_floor = TTR_SOLVER_TRIDIAG, rmax = n) returns; 2 when its sigma pass ttr_spectrum_flat's batch-mode test (sigma_r >= thr
 * sigma_1 > 0: declined for a close pair only), so that flat[b] != 0 IS the pass-through flag of the bond's second pass -- no
 * ttr_spectrum_flat launch, no merge of two flag arrays (round 4; two launches per bond on a latency-bound chain).  `flat` is optional.
 * A 64 x 64 matrix whose diagonal is exactly zero from index 32 on (the Gram matrix of a carry with zero rows 32.., a bond whose QR
 * packed its rows) is solved as its leading 32 x 32 block (G must be a Gram matrix: G_ii = 0 => row i = 0): flat[b] = 1 as above,
 * otherwise V[b] = blockdiag(V11, I), sigma[b][32:] = 0.
 */
int ttr_eigh_top_ok(int64_t n, int64_t r);
int ttr_eigh_top(int dtype, int64_t n, int64_t batch,
                 const void* G, int64_t ldg, int64_t strideG, int64_t gparts, int64_t stride_gpart,
                 void* V, int64_t ldv, int64_t strideV,
                 void* sigma, int64_t stride_sigma,
                 int32_t* info, int64_t r, double thr, int32_t* flat, int need_all, void* stream);
/* `need_all` != 0 (ABI 10; the first pass of an eps-mode bond, whose rank rule needs every singular value): the top-r path only
 * takes items of which it computes EVERY eigenpair -- r >= the item's live size, i.e. the zero-tail (32 x 32) problems of a packed
 * bond under a cap r >= 32; all other items get the full QL decomposition of the same launch.  sigma / V are then complete for
 * every item (structural zeros / unit vectors beyond a shrunk block), exactly as ttr_eigh_trunc(TTR_EIG_RAW, TTR_SOLVER_TRIDIAG)
 * returns them, and `flat` is not meaningful (the caller runs ttr_spectrum_flat with use_delta). */

/*
 * Selected eigenpairs of symmetric matrices with 64 < n <= ttr_eigsel_max_n() (1024): the k <= 64 LARGEST eigenvalues and their
 * eigenvectors -- torch.linalg.eigh / svd of round.py:96, 115 on the Gram matrix of a dense TT-SVD bond when only the top of the
 * spectrum is looked at (batch mode with a rank cap far below n: BASELINE config C3, n = 256, rmax = 8).  LAPACK syevx class:
 *   ttr_tridiag       A[b] (n x n symmetric, leading dimension lda, DESTROYED) = Q T Q^T by Householder reflectors, one
 *                     workgroup per matrix; d[b][n], e[b][n] (e[i] couples i and i + 1, e[n-1] = 0), tau[b][n]; reflector k is
 *                     left in row k of A (columns k+1 .., leading 1 stored).  `workspace` (optional, ttr_tridiag_workspace_bytes):
 *                     with it, few (<= 4) big (n >= 512) matrices are reduced by 2 n - 1 launches that spread every step's pass
 *                     over the whole chip instead of one workgroup per matrix
 *   ttr_tri_eigsel    lam[b][k] = the k largest eigenvalues of T (descending; Sturm-count multisection) and Z[b][n][k] their
 *                     unit eigenvectors (twisted factorisation: close eigenvalues leave them only NEARLY orthogonal -- the host
 *                     shim orthonormalises Z with ttr_qr and falls back when a column collapses);
 *                     scratch: ttr_eigsel_scratch_bytes(dtype, n, batch)
 *   ttr_tridiag_back  Z[b] <- Q[b] Z[b] (in place), A / tau as left by ttr_tridiag
 */
int ttr_eigsel_max_n(void);
int64_t ttr_eigsel_scratch_bytes(int dtype, int64_t n, int64_t batch);
int64_t ttr_tridiag_workspace_bytes(int dtype, int64_t n, int64_t batch);
int ttr_tridiag(int dtype, int64_t n, int64_t batch, void* A, int64_t lda, int64_t strideA, void* d, void* e, void* tau,
                void* workspace, int64_t workspace_bytes, void* stream);
int ttr_tri_eigsel(int dtype, int64_t n, int64_t batch, int64_t k, const void* d, const void* e, void* lam, void* Z, void* scratch,
                   int64_t scratch_bytes, void* stream);
int ttr_tridiag_back(int dtype, int64_t n, int64_t batch, int64_t k, const void* A, int64_t lda, int64_t strideA, const void* tau,
                     void* Z, void* stream);

/*
 * Block-Jacobi driver for symmetric eigenproblems above the single-workgroup limit -- torch.linalg.eigh / svd of
 * round.py:96, 115 on the n = I r Gram matrices of a dense TT-SVD (BASELINE configs C1: n = 1024, C3: n = 256).
 * G[item] (n x n, n = 2 b npairs, b <= 32) and the accumulated eigenvector matrix V[item] are updated IN PLACE, one
 * round = ttr_bj_solve + ttr_bj_apply with the round's pairing `pair_tab` (device int32 [npairs][2]: block indices of
 * every pair; over nbk - 1 rounds of a round-robin tournament every two blocks meet once = one sweep):
 *   ttr_bj_solve    W[item * npairs + p] (w x w, w = 2 b) = eigenvectors of the pair's diagonal problem
 *                   [[G_ii, G_ij], [G_ji, G_jj]], diagonal-matched column order (W -> I as the problem -> diagonal);
 *                   `scratch`: ttr_bj_scratch_bytes(...).
 *   ttr_bj_apply    G[P, Q] <- W_p^T G[P, Q] W_q for all pairs of pairs, V[:, Q] <- V[:, Q] W_q; with `offsq` (device
 *                   double [items], optional) the squared off-diagonal entries of the updated G are added to offsq[item].
 *   ttr_bj_control  once per sweep: sets ctrl[0] = 1 ("converged") when no pair problem of the sweep rotated anything
 *                   (relative != 0) or when max_item sqrt(offsq[item]) / gnorm[item] <= tol or stagnates (relative == 0;
 *                   gnorm = ||G[item]||_F of the input, element type of `dtype`); resets ctrl[1] and offsq.
 * ctrl: device int32 [4] = {converged, pair problems that rotated in the current sweep, sweeps performed, unused}, zeroed
 * by the caller; state: device double [items + 1] = offsq followed by the previous sweep's ratio (initialise to -1).
 * Once ctrl[0] is set every later ttr_bj_* launch returns at its first instruction: the caller enqueues the maximum
 * number of sweeps and never reads anything back.
 */
int64_t ttr_bj_scratch_bytes(int dtype, int64_t b, int64_t npairs, int64_t items);
int ttr_bj_solve(int dtype, int64_t b, int64_t npairs, int64_t items,
                 const void* G, int64_t ldg, int64_t strideG, const int32_t* pair_tab,
                 void* W, void* scratch, int32_t* ctrl, void* stream);
int ttr_bj_apply(int dtype, int64_t b, int64_t npairs, int64_t items,
                 void* G, int64_t ldg, int64_t strideG, void* V, int64_t ldv, int64_t strideV,
                 const int32_t* pair_tab, const void* W, const int32_t* ctrl, double* offsq, void* stream);
int ttr_bj_control(int dtype, int64_t items, int32_t* ctrl, double* state, const void* gnorm, int relative, double tol,
                   void* stream);

/*
 * CP-ALS building blocks (tn.Tensor(X, ranks_cp=R), tensor.py:279-394; BASELINE config C4).
 * ttr_krp_contract:  out[p, q, r] = sum_j T[p, j, q, r] * B[j, r]   (T contiguous [P, J, Q, R], B is J x R with
 * leading dimension ldb, out contiguous [P, Q, R]).  One step of a fused MTTKRP: the reference materialises the
 * Khatri-Rao matrix (`torch.einsum("ir,jr->ijr")` chain, tensor.py:328-334) and a permuted copy of the dense
 * unfolding (tensor.py:336) and multiplies them (tensor.py:338); here the tensor is first contracted with one
 * factor by ttr_gemm (X read once) and the remaining modes are folded in with this kernel (Q = 1: trailing mode,
 * P = 1: leading mode), each reading its input exactly once.
 * ttr_hadamard: out[i] = a[i] * b[i] -- the Gram Hadamard product `prod *= grams[m]` (tensor.py:331).
 */
int ttr_krp_contract(int dtype, int64_t P, int64_t J, int64_t Q, int64_t R,
                     const void* T, const void* B, int64_t ldb, void* out, void* stream);
int ttr_hadamard(int dtype, int64_t count, const void* a, const void* b, void* out, void* stream);

/*
 * Slice-wise Kronecker product of two TT cores (the Hadamard product of two tensor trains multiplies the ranks):
 *   out[b, r1*S1 + s1, i, r2*S2 + s2] = a[b, r1, i, r2] * c[b, s1, i, s2],   a [B, R1, I, R2], c [B, S1, I, S2] contiguous.
 * Replaces: `_core_kron` tensor.py:2309-2320 (called from `Tensor.__mul__`, tensor.py:687-773), the producer of the
 * rank-inflated trains the rounding sweeps exist for.
 */
int ttr_core_kron(int dtype, int64_t batch, int64_t R1, int64_t S1, int64_t I, int64_t R2, int64_t S2,
                  const void* a, const void* c, void* out, void* stream);

/*
 * out[b] = sqrt(sum(x[b]^2)) over `count` contiguous elements (accumulated in double, stored in dtype).
 * Replaces: torch.norm(cores[-1]) tensor.py:2039-2051 and torch.norm(M) round.py:80.
 */
int ttr_norm(int dtype, int64_t count, int64_t batch, const void* x, int64_t stride_x, void* out, void* stream);

/*
 * out[b][i][j] = in[b][i][j] * s[b][j]   (mode TTR_SCALE_MUL)  or  / s[b][j]  (TTR_SCALE_DIV).
 * Replaces: `left * svd[1][:rank]` round.py:169-172 and the 1/sigma column scaling round.py:175.
 */
int ttr_scale_cols(int dtype, int64_t rows, int64_t cols, int64_t batch,
                   const void* in, int64_t ldi, int64_t stride_in,
                   const void* s, int64_t stride_s, int mode,
                   void* out, int64_t ldo, int64_t stride_out, void* stream);
/* x[b][:, j] <- 0 for j >= keep[b], in place (keep: device int32 [batch], e.g. ttr_eigh_trunc's info).  The device-side
 * form of `left = vectors[..., :rank]` (round.py:160-161) for a sweep that computes every bond at its rank cap and reads the
 * selected ranks back once, at the end (at most one host synchronisation per round_tt). */
int ttr_mask_cols(int dtype, int64_t rows, int64_t cols, int64_t batch, void* x, int64_t ldx, int64_t stride_x,
                  const int32_t* keep, void* stream);


/*
 * Fused kernels of the right-to-left truncation sweep: the (R <= 64) x n right unfolding M of a core is streamed once,
 * 16 columns per wave and step, every intermediate stays in MFMA accumulator registers.
 *   ttr_rowgram   G = M M^T                                           replaces `M @ M^T` round.py:104-109
 *   ttr_rotgram   G = (V1^T M)(V1^T M)^T  (V1: R x R)                 pass 2 of the two-pass 'svd' truncation (round.py:96):
 *                 the Gram matrix of the rows ROTATED by the pass-1 eigenvectors, formed from the rotated data (small
 *                 rows from small numbers) without ever writing the rotated matrix
 *   ttr_project   right = diag(1/sigma) U^T M (ro x n),  left = U diag(sigma) (R x ro, optional),  U = V1 V2[:, :ro]
 *                 (V1 = NULL: U = V2[:, :ro]); scale_right = 0: right = U^T M, left = U        replaces round.py:163-172
 * G is written as `nparts` split-K partial matrices per item ([batch][nparts][R][R], contiguous; nparts from
 * ttr_sweep_gram_parts); ttr_eigh_trunc sums them on load.
 */
int64_t ttr_sweep_gram_parts(int64_t n, int64_t batch);
int ttr_rowgram(int dtype, int64_t R, int64_t n, int64_t batch, const void* M, int64_t ldm, int64_t strideM,
                void* G, int64_t nparts, const int32_t* rows32, void* stream);
/* `rows32` (ttr_rowgram, ttr_project; optional device int32 [batch]): != 0 -> rows 32.. of this item's M are exactly zero and are
 * not loaded (half the reads of an HBM-bound kernel).  Source: the flags a packed ttr_qr_factor_pushed leaves in its workspace at
 * byte offset ttr_qr_pushed_flag_offset (TTR_KNOB_QR_PACK: the carry ttr_qr_apply_pushed produces from such a factorisation has
 * zero rows kk >= 32). */
int64_t ttr_qr_pushed_flag_offset(int dtype, int64_t I, int64_t n, int64_t batch);
/* The same flags for a carry that no fused push follows -- the LAST core of the left-to-right sweep, tensor.py:2053-2056's first
 * truncation: flag[b] = 1 when rows 32.. of the 64 x cols matrix R[b] hold at most (c eps)^2 of ||R[b]||_F^2 (c =
 * TTR_KNOB_QR_RANK_SKIP; the packing test of ttr_qr_factor_pushed).  Since round 5 the sweep (tntorch_amd/_hipops.py, ttr_round_tt)
 * calls it ON THE CARRY ITSELF, M = R_{N-2} x (last core) as a 64 x (I r) matrix, not on the R factor: what ttr_rowgram /
 * ttr_rotgram / ttr_project then treat as zero (`rows32`) is below c eps ||M||_F whatever the last core's condition number is --
 * but it is DROPPED energy, not a structural zero: rows 32.. of M are non-zero in memory.  The eps-mode rank rule accounts for it
 * through the noise floor (TTR_KNOB_RANK_NOISE_FLOOR, default 1: sigma[32..] count as eps sigma_0 and ttr_spectrum_flat tests the
 * whole spectrum with its robustness margin E = 64 n eps sigma_0^2 >= (c eps)^2 ||M||^2); with the floor switched off the 32
 * trailing values are exact zeros to the rule and a tolerance below c eps (fp64, eps < 2e-15) may select fewer directions than
 * LAPACK's (ABI 9). */
int ttr_carry_rows32(int dtype, int64_t cols, int64_t batch, const void* R, int64_t ldr, int64_t strideR, int32_t* flag,
                     void* stream);
int ttr_rotgram(int dtype, int64_t R, int64_t n, int64_t batch, const void* M, int64_t ldm, int64_t strideM,
                const void* V1, int64_t ldv1, int64_t strideV1, void* G, int64_t nparts, const int32_t* skip,
                const int32_t* rows32, void* stream);
/*
 * When is the second pass needed?  The first Gram matrix G = M M^T carries sigma_i^2 with an ABSOLUTE error of c eps sigma_1^2
 * (c: a small constant of the fp32 accumulation), i.e. sigma_i and the unit norm of row i of `right` to c eps (sigma_1 /
 * sigma_i)^2 / 2: the second pass exists for kept singular values far below sigma_1.  When the `keep` kept ones lie within
 * a factor 1 / thr of each other (the host shim uses thr = 1/8; measured with the kernels' accumulation order emulated
 * on the CPU: right-orthonormality 8e-7 / relative sigma error 4e-7 after one pass against 4e-7 / 2e-7 after two at
 * sigma_keep = sigma_1 / 8), one pass already is in the accuracy class of the two.  ttr_spectrum_flat writes
 * flat[b] = (sigma[b][r - 1] >= thr * sigma[b][0] > 0) from pass 1's sigma (sorted decreasing), r = the rank the item will get:
 *   use_delta = 0 or delta^2 = 0 (batch mode, round.py:149-150, or no eps): r = keep = min(rank cap, n);
 *   use_delta = 1 (delta2, or *delta2_dev when given): r from the tail-energy rule of round.py:147-158 on pass 1's sigma, and the
 *     item only qualifies when that decision is robust against pass 1's absolute error E = 64 n eps sigma_1^2 of a tail energy:
 *     tail(r) <= delta^2 - E and tail(r - 1) > delta^2 + E (rank cap binding, r = keep: only the latter) -- pass 2 would select
 *     the same rank.  (fp32 with eps = 1e-4: E exceeds delta^2, nothing qualifies; fp64 trains do.)
 * ttr_rotgram skips items with skip[b] != 0 (their G is not written) and ttr_eigh_trunc passes them through
 * (`skip_items`: V = I, sigma = sigma_in, rank rule as usual on sigma_in).
 */
int ttr_spectrum_flat(int dtype, int64_t n, int64_t batch, const void* sigma, int64_t stride_sigma, int64_t keep, double thr,
                      int use_delta, double delta2, const double* delta2_dev, int32_t* flat, const int32_t* rows32, void* stream);
/* `rows32` (optional, ABI 10): the flags ttr_rowgram took for this bond -- sigma[32..] of a flagged item are structural zeros (exact
 * in either pass), so in eps mode the rule's cut of them is certain and only the 32 computed values are tested for robustness:
 * a rank-deficient train (t = g + g) rounded through the reference's NON-batch call skips its second Gram pass like a batch. */
int ttr_project(int dtype, int64_t R, int64_t n, int64_t ro, int64_t batch,
                const void* M, int64_t ldm, int64_t strideM,
                const void* V1, int64_t ldv1, int64_t strideV1,
                const void* V2, int64_t ldv2, int64_t strideV2,
                const void* sigma, int64_t stride_sigma, int scale_right,
                void* right, int64_t ldr, int64_t strideR,
                void* left, int64_t ldl, int64_t strideL, const int32_t* rows32, void* stream);

/*
 * The same fused kernels for TALL matrices (rows x n, n <= 64, row-major): the unfoldings a dense right-to-left TT-SVD
 * truncates first -- prod(I_1..I_{N-1}) rows, I_N columns, the whole input tensor in bytes (tensor.py:401-408 via
 * round.py:101-135 with the Gram matrix on the short side).  The contraction runs over the rows, 16 per wave and step.
 *   ttr_colgram     G = M^T M   (V1 = NULL)   or   G = (M V1)^T (M V1): the rotated matrix is never written
 *                   (`workspace`: split-K partials, size from ttr_colgram_workspace_bytes; G is the reduced n x n matrix)
 *   ttr_colproject  left = M U [diag(1/sigma)] (rows x ro),  right = [diag(sigma)] U^T (ro x n),  U = V1 V2[:, :ro]
 *                   (left_ortho = 1: the scalings in brackets, round.py:173-178; 0: round.py:179-182)
 * A 'svd' truncation of such an unfolding therefore reads it three times and writes only ro / n of its size.
 */
int64_t ttr_colgram_workspace_bytes(int dtype, int64_t rows, int64_t n, int64_t batch);
int ttr_colgram(int dtype, int64_t rows, int64_t n, int64_t batch, const void* M, int64_t ldm, int64_t strideM,
                const void* V1, int64_t ldv1, int64_t strideV1, void* G, void* workspace, int64_t workspace_bytes,
                const int32_t* skip /* optional [batch], as ttr_rotgram's */, void* stream);
int ttr_colproject(int dtype, int64_t rows, int64_t n, int64_t ro, int64_t batch,
                   const void* M, int64_t ldm, int64_t strideM,
                   const void* V1, int64_t ldv1, int64_t strideV1,
                   const void* V2, int64_t ldv2, int64_t strideV2,
                   const void* sigma, int64_t stride_sigma, int left_ortho,
                   void* left, int64_t ldl, int64_t strideL,
                   void* right, int64_t ldr, int64_t strideR, void* stream);

/*
 * Exact power-of-two normalisation, one launch: e[b] = binary exponent of ||x[b]|| (0 for a zero / non-finite norm),
 * out[b] = x[b] * 2^-e[b], and, when `expo_acc` is given, expo_acc[b] += e[b].  (`out` may alias `x`; out = NULL:
 * exponents only.)
 * Replaces the LAPACK-internal rescaling the reference relies on (torch.linalg.qr / svd are scale safe,
 * tensor.py:1816, round.py:96): the fp32 sweep keeps every R factor at O(1) and returns the summed exponent to core 0.
 */
int ttr_pow2_normalize(int dtype, int64_t count, int64_t batch, const void* x, int64_t stride_x,
                       void* out, int64_t stride_out, int32_t* e_out, int32_t* expo_acc, void* stream);

/*
 * out[b][i] = x[b][i] * scale[b * stride_scale] * 2^(expo_sign * expo[b])   (scale and/or expo may be NULL;
 * stride_scale = 0 broadcasts one scalar).  Replaces scalar multiplications of cores (tensor.py:687-773 with a
 * scalar operand) and gives the exponents of ttr_pow2_normalize back.
 */
int ttr_scale_batch(int dtype, int64_t count, int64_t batch, const void* x, int64_t stride_x,
                    const void* scale, int64_t stride_scale, const int32_t* expo, int expo_sign,
                    void* out, int64_t stride_out, void* stream);

/*
 * Orthonormal completion of the numerically null directions a truncation keeps.  X[b] holds `r` vectors of length `n`
 * (vector i, element k at X[b * strideX + i * vec_stride + k * elem_stride]); sigma[b] are the singular values,
 * decreasing.  Vectors i < r with sigma_i <= dead_rel * sigma_0 carry rounding noise only (their sigma is below the
 * resolution of the input): they are re-orthogonalised against all previous vectors (modified Gram-Schmidt, twice;
 * a vector that vanishes is replaced by a hashed pseudo-random one) and normalised, so that the factor is orthonormal
 * to working accuracy like the V of a LAPACK SVD (round.py:96), while the product left * right changes by
 * O(dead_rel * sigma_0) only.  Items without such vectors exit immediately (the normal case).
 */
int64_t ttr_orth_fixup_workspace_bytes(int dtype, int64_t r, int64_t n, int64_t batch, int64_t elem_stride);
int ttr_orth_fixup(int dtype, int64_t r, int64_t n, int64_t batch, void* X, int64_t vec_stride, int64_t elem_stride,
                   int64_t strideX, const void* sigma, int64_t stride_sigma, double dead_rel, const int32_t* rank_dev,
                   void* workspace, int64_t workspace_bytes, void* stream);
/* `workspace` (ABI 10; may be NULL): with ttr_orth_fixup_workspace_bytes(...) > 0 bytes of it, a LARGE batch (>=
 * TTR_KNOB_ORTH_SPLIT items, default 2048; vectors as rows, <= 64 of them) runs its first two rounds as three launches each -- the
 * Gram matrix by ttr_rowgram's kernel (split-K partials summed in double), the coefficients, the streamed product X_dead <- W X
 * (which, for <= 32 vectors, also leaves the second round's Gram matrix: that round has two launches) --
 * instead of one workgroup per item, and ONE launch of the single-workgroup kernel finishes the items that need more (remainders
 * that collapsed and were replaced: rare).  Same rounds, same semantics (a vector with non-finite entries counts as the zero
 * vector), ~2x the HBM rate.  Without a workspace (or below the threshold) the single-launch kernel does everything. */
/* `rank_dev` (optional, device int32 [batch]): only the first min(r, rank_dev[b]) vectors of item b are looked at -- a sweep
 * that computes its factors at the rank cap and cuts them to the selected rank later does not complete vectors it drops. */

/*
 * The WHOLE rounding of a batch of tensor trains in ONE call (ABI 10): the left-to-right orthogonalisation loop of
 * tensor.py:1905-1906 (tensor.py:1816-1832 per core) followed by the right-to-left truncation loop of tensor.py:2053-2083
 * (round.py:52-187 per bond) -- the two Python loops of `Tensor.round_tt` (tensor.py:2008-2083).  The entry enqueues, on
 * `stream`, the same kernels in the same order as the per-kernel entries above would be called by a host loop (results are
 * bit-identical to that loop); it allocates nothing, synchronises nothing and reads nothing back.
 *
 *   shapes     host int64 [3 N]: (r0, I, r1) of every input core; cores_in[mu] = contiguous [batch][r0][I][r1] on the device
 *   rcap       host int64 [N - 1] or NULL: rank cap of bond mu = 1 .. N-1 at index mu - 1 (>= 2^31 - 1: none, round.py:83-84)
 *   algorithm  TTR_ALG_SVD (two Gram passes: the accuracy class of gesdd, round.py:96) / TTR_ALG_EIG (round.py:101-135)
 *   eps_mode   0 = batch semantics (round.py:149-150: every bond is cut at its cap, eps is ignored; any batch);
 *              1 = the reference's NON-batch rule for ONE train (batch == 1): delta = eps / max(1, sqrt(N - 1)) ||last core||
 *              (tensor.py:2039-2051) is formed ON THE DEVICE, every bond is computed at its cap, the rank the rule of
 *              round.py:147-158 selects is written to ranks_dev[mu - 1] and the cores are ZERO beyond the selected ranks: the
 *              caller reads ranks_dev once, after the call, and slices (ranks_dev[i] == 0: the zero guard of round.py:137-145)
 *   flat_thr   items whose kept singular values lie within a factor 1 / flat_thr skip the second Gram pass (ttr_spectrum_flat;
 *              0 = never); use_eigh_top != 0: batch-mode first passes through ttr_eigh_top where ttr_eigh_top_ok
 *   cores_out  cores_out[mu] = contiguous [batch][q_mu][I][q_{mu+1}], q_0 = r0 of core 0, q_N = r1 of the last core,
 *              q_mu = max(1, min(rcap[mu - 1], rows of bond mu)) -- the ranks batch mode produces (fixed by the shapes)
 *   zero_flag_dev  optional device-ACCESSIBLE int32 [1] (eps_mode 0): receives the largest rank-rule result of the FIRST truncation
 *              over the batch; 0 = every item hit the zero guard (round.py:137-141) and the caller returns the rank-1 zero train.
 *              Written by a one-workgroup kernel right after that truncation (with a system-scope fence): device memory, or --
 *              what tntorch_amd passes since round 6 -- a pinned host word the caller initialises to a sentinel and polls, so
 *              that it need not wait for the rest of the sweep before it returns.  eps_mode 1 (round 6): when given, an int32
 *              [N - 1] device-accessible array that receives a COPY of ranks_dev as soon as the last bond is decided (N - 1 <= 64),
 *              i.e. before the last projection and the first core are formed -- the same polling pattern for the reference's
 *              non-batch call
 *   workspace  ttr_round_tt_workspace_bytes(...) bytes, caller-owned; the call may be repeated with the same workspace once the
 *              previous one has completed on the stream
 * Envelope: every TT rank <= ttr_qr_max_cols, every core inside the fused push (k, Rin <= 64, k I >= n), every bond a
 * <= 64-row matrix with at least as many columns.  ttr_round_tt_workspace_bytes returns TTR_E_UNSUPPORTED (< 0) outside it:
 * the caller then runs its own loop over the per-kernel entries.
 */
#define TTR_ALG_SVD 0
#define TTR_ALG_EIG 1
int64_t ttr_round_tt_workspace_bytes(int dtype, int64_t N, const int64_t* shapes, const int64_t* rcap, int64_t batch,
                                     int eps_mode);
int ttr_round_tt(int dtype, int64_t N, const int64_t* shapes, int64_t batch, const void* const* cores_in, const int64_t* rcap,
                 int algorithm, int eps_mode, double eps, double flat_thr, int use_eigh_top, void* const* cores_out,
                 int32_t* ranks_dev, int32_t* zero_flag_dev, void* workspace, int64_t workspace_bytes, void* stream);

/* Per-kernel device timing (HIP events on `stream`), used by bench.py for the roofline line. */
#define TTR_PROF_GEMM 0
#define TTR_PROF_QR_FACTOR 1
#define TTR_PROF_QR_APPLY 2
#define TTR_PROF_EIGH 3
#define TTR_PROF_MISC 4
#define TTR_PROF_ROTGRAM 5
#define TTR_PROF_PROJECT 6
#define TTR_PROF_ROWGRAM 7
#define TTR_PROF_NKINDS 8
/* Diagnostics: when set to a device buffer of >= 64 int64, block (0,0) of every level-0 QR factor launch writes
 * its s_memtime stamps at phase boundaries there.  NULL disables. */
int ttr_debug_set_qr_stamps(void* device_buffer);
/* Diagnostics / A-B measurements: select kernel variants at run time (process-wide, not thread safe).
 *   TTR_KNOB_QR_PANEL  1 = the 8-wave QR blocks factor / apply their panel columns in pairs (two reflectors per step,
 *                      v_permlane-swap reductions; default), 0 = one reflector at a time (round-1 kernel). */
#define TTR_KNOB_QR_PANEL 0
/*   TTR_KNOB_BJ_INNER_SWEEPS  cyclic sweeps ttr_bj_solve spends on every pair problem (0 = until the pair problem has
 *                      converged; default 1: the outer iteration converges with inexact inner solves -- W is a product of
 *                      rotations either way -- and a round costs a fraction) */
#define TTR_KNOB_BJ_INNER_SWEEPS 1
/*   TTR_KNOB_GEMM_BIG  1 (default) = fp32 products with both output dimensions >= 128 run on the 128 x 128-tile kernel
 *                      (symmetric products: upper-triangle tiles only); 0 = the 64 x 64-tile kernel everywhere (A/B runs) */
#define TTR_KNOB_GEMM_BIG 2
/*   TTR_KNOB_QR_STAMP_BX / _BY  which level-0 block of ttr_qr_factor* writes the cycle stamps of ttr_debug_set_qr_stamps
 *                      (block index within the matrix / batch item; default (0, 0), which starts with the first wave of
 *                      workgroups -- a block in the middle of the grid shows the steady state) */
#define TTR_KNOB_QR_STAMP_BX 3
#define TTR_KNOB_QR_STAMP_BY 4
/*   TTR_KNOB_QR_F64_NW4  1 = fp64 TSQR trees consist of 256-row (4-wave) blocks only -- two blocks per CU; 0 (default) = the
 *                      512-row (8-wave) blocks of the fp32 path (132 KB of LDS in fp64: one block per CU; measured 17 %
 *                      faster on config C2 all the same).  Bit 1 (value 2 / 3): the same for fp32 (round 4's A/B on the metric
 *                      workload).  Set it BEFORE any workspace is sized (an A/B switch). */
#define TTR_KNOB_QR_F64_NW4 5
/*   TTR_KNOB_QR_RANK_SKIP  c (default 8): the 8-wave QR blocks do not factor a panel whose remaining part is below c eps of the
 *                      block's Frobenius norm (rank-inflated trains: H = I, tau = 0; ttr_qr_apply skips trailing identity
 *                      panels); 0 = every panel is factored (A/B). */
#define TTR_KNOB_QR_RANK_SKIP 6
/*   TTR_KNOB_QR_PACK  3 (default) = level 0 of a fused push (ttr_qr_factor_pushed) whose R factor has rows 32 .. 63 below
 *                      c eps of its norm (c: TTR_KNOB_QR_RANK_SKIP) drops the pushed rows (kk >= 32, i) as zeros and packs two
 *                      mode indices per wave: blocks b < nb / 2 do all the work, the others write a zero R and return, and the
 *                      launch is block-major (all working blocks first); decided per item on the device,
 *                      ttr_qr_apply_pushed follows the per-item flag the factor kernel leaves in the workspace.  1 / 2 = the
 *                      item-major variants (measured without gain: the dispatcher stalls on alternating long / short
 *                      workgroups); 0 = never (required for ttr_qr_apply_pushed_gram, whose epilogue assumes the unpacked map). */
#define TTR_KNOB_QR_PACK 7
/*   (8 = TTR_KNOB_EIGH_SMALL, below)
 *   TTR_KNOB_RANK_NOISE_FLOOR  c (default 1 since round 6; 0 = off): the rank rule of round.py:147-158 (eps mode: ttr_eigh_trunc
 *                      with use_delta, ttr_spectrum_flat) sees every singular value at no less than c eps sigma_0.  Background: the
 *                      zero-tail eigenproblems of a rank-deficient bond (t = g + g) return EXACT zeros for the null directions, which
 *                      the rule would cut at any delta >= 0; LAPACK's gesdd -- the reference -- returns rounding noise of order
 *                      eps sigma_0 there, so `round_tt(rmax=48)` (eps = 1e-14 by default) on a numerically rank-32 fp32 train keeps
 *                      the cap.  c = 1 reproduces the reference's outcome (fp32: the cap; fp64: eps^2 < 1e-28, still cut);
 *                      TTR_STRICT_RANKS=0 in the environment of tntorch_amd sets 0. */
#define TTR_KNOB_RANK_NOISE_FLOOR 9
/*   TTR_KNOB_ORTH_ROUNDS  (default 4) upper bound on the rounds of ttr_orth_fixup's block variant (diagnostics: what a round costs;
 *                      fewer than the data needs leaves the completed directions short of orthonormal).  In census mode
 *                      (ttr_prof_enable(2)) the MISC counters receive flops += rounds executed, bytes += items with dead directions. */
#define TTR_KNOB_ORTH_ROUNDS 10
/*   TTR_KNOB_JACOBI_LIVE_WAVE  1 = ttr_eigh_trunc with TTR_SOLVER_JACOBI_LIVE on n <= 64 runs ONE wave per matrix (no cross-wave
 *                      barrier in the round loop): measured SLOWER (1.25 vs 0.80 ms per launch of 2048 matrices, round 5);
 *                      0 (default) = the four-wave kernel. */
#define TTR_KNOB_JACOBI_LIVE_WAVE 11
/*   TTR_KNOB_ORTH_V2  1 = ttr_orth_fixup's block variant for <= 32 vectors runs the round-5 inner loops (wide LDS operand
 *                      reads with a permuted K order, dead tile rows only, chunk columns split over the four waves); 0 = round 4's (A/B);
 *                      2 (default) = 1 + the three-launch rounds of a large batch with <= 32 vectors skip the second round's Gram launch:
 *                      the first round's product launch leaves the Gram matrix of the vectors as it wrote them (A/B against 1). */
#define TTR_KNOB_ORTH_V2 12
/*   TTR_KNOB_QR_INTERLEAVE  1 (default) = the block-major launches of TTR_KNOB_QR_PACK = 3 (ttr_qr_factor_pushed level 0,
 *                      ttr_qr_apply_pushed level 0) order the workgroups of each half item by item (an item's working blocks follow
 *                      each other), so that resident workgroups differ in the address bits their block index fixes -- HBM channel
 *                      spreading; 0 = all items' block 0, then all items' block 1, ... (round 4; A/B). */
#define TTR_KNOB_QR_INTERLEAVE 13
/*   TTR_KNOB_SWEEP_STAGGER  the fused <= 64-row sweep kernels walk an item's columns starting at step (item index mod steps) and
 *                      wrap around, so that resident workgroups do not read the same column offsets (= the same HBM channels:
 *                      every item's rows are ldm elements apart and aligned alike) at the same time.  1 (default) = ttr_project only
 *                      (independent columns: bit-identical results); 2 = ttr_rowgram / ttr_rotgram as well (their sums are then
 *                      accumulated in an item-dependent order); 0 = in order (A/B). */
#define TTR_KNOB_SWEEP_STAGGER 14
/*   TTR_KNOB_ORTH_SPLIT  batch size (per launch) from which ttr_orth_fixup, given a workspace, runs its rounds as three launches
 *                      (default 2048; 0 = always the single-launch kernel): 13 % on a 4096-train step whose kept directions mostly
 *                      lie below the resolution (sigma ~ 2^-j), nothing measurable on one without dead directions at that size
 *                      (the launches that exit at once cost 0.3 - 2 % at 1024 items: hence the threshold). */
#define TTR_KNOB_ORTH_SPLIT 15
/*   TTR_KNOB_QR_STAGGER  k (default 0 = off; round 6, A/B): workgroups 256 .. 511 of a fused push + factor launch (the second
 *                      resident block of every CU under round-robin dispatch) start k x 1024 cycles late, so that a CU's two blocks
 *                      alternate their HBM phase (the push) and their panel chain instead of running them side by side. */
#define TTR_KNOB_QR_STAGGER 16
/*   TTR_KNOB_QR_PACK_PRE  1 (default; round 6) = the row-packing decision of a fused push + factor launch (TTR_KNOB_QR_PACK) is
 *                      taken by a one-wave-per-item kernel AHEAD of the launch and read by its blocks: an absorbed block returns at
 *                      once instead of staging Rm first (9 k cycles each, 16384 of them at the tail of every level-0 launch of the
 *                      metric), its partner writes the zero R block and taus for it; 0 = every block derives the decision from Rm
 *                      itself (round 5; A/B).  + 2: the level-1 launch of ttr_qr_apply_pushed does NOT idle the waves that hold the
 *                      absorbed leaves' (exactly zero) rows of a packed item (A/B; default: they load, multiply and store nothing). */
#define TTR_KNOB_QR_PACK_PRE 17
/*   TTR_KNOB_EIGH_BIG_OCC  0 (default) / 3 / 2 = waves per SIMD the 64-row instance of ttr_eigh_top is built for in fp32 launches of
 *                      >= 1024 matrices (0: four waves, 128 registers, 106 spilled; 3: 168 registers; 2: 256 registers, none
 *                      spilled); bit-identical results (A/B, round 6). */
#define TTR_KNOB_EIGH_BIG_OCC 18
/*   TTR_KNOB_EIGH_SMALL  1 = ttr_eigh_top and the tridiagonal solver of ttr_eigh_trunc (TTR_EIG_RAW / REF) on 64 x 64
 *                      matrices first run the 32-row instance of their kernel over the zero-tail items (Gram matrices of packed
 *                      bonds) and then the 64-row one over the rest; 0 = one launch, the 64-row instance shrinks such items
 *                      itself (A/B); 2 (default) = 1, and fp32 ttr_eigh_top launches of >= 1024 matrices use a build of the 32-row
 *                      instance capped at 168 VGPRs (three waves per SIMD, 44 spilled registers; 3: 128 VGPRs, four waves) --
 *                      same results bit for bit, 17 % less eigensolver time at B = 4096. */
#define TTR_KNOB_EIGH_SMALL 8
int ttr_debug_set_knob(int knob, int value);
int ttr_prof_enable(int on);   /* 0 = off, 1 = per-kind device times, 2 = times + executed-work census (ttr_prof_collect_work) */
/* Synchronises the recorded events; fills total milliseconds and launch counts per kind; resets. */
int ttr_prof_collect(double* ms, int64_t* launches);
/* Census mode (ttr_prof_enable(2); ABI 10): flops / bytes the instrumented launches EXECUTED since the last collect, per kind --
 * read off the same device-side decisions the kernels took (zero taus of rank-skipped / absorbed QR panels, packing flags,
 * `rows32` and pass-through flags), by tiny kernels enqueued outside the timed scopes.  Instrumented: qr_factor, qr_apply (flops),
 * rowgram, rotgram, project (flops and bytes), gemm (flops and bytes of the dense product).  Synchronises the device; resets. */
int ttr_prof_collect_work(double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* TTROUND_HIP_H */
