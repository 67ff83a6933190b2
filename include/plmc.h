/*
 * plmc.h -- C ABI of the MI355X-native exact-GP hot path behind the projectedlmc model API.
 *
 * The reference (QWERTY6191/projected-lmc) has no FFI: its hot path is reached through
 * gpytorch calls made by its Python classes.  Each entry point below replaces the named
 * reference call site (file:line in /root/reference/projectedlmc/projected_lmc.py unless
 * stated) and the third-party kernels that call site triggers.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All pointers are DEVICE pointers
 *     owned by the caller (workspace included); the library allocates nothing and keeps no
 *     pointer after return.  Every call is asynchronous on `stream` (a hipStream_t passed as
 *     void*).  Return value: 0 = launched, <0 = argument / HIP error (see plmc_last_error()).
 *   - dtype by suffix: _f32 / _f64.  Everything row-major.
 *   - Blocked algorithms use NB = 128 (plmc_block()).  n_pad = plmc_pad(n) = n rounded up to NB.
 *   - "Factor buffer" A, one per latent GP (batch stride strideA elements):
 *        n_pad rows x lda columns, lda >= n_pad + naug_pad (+ n_pad when the inverse factor is
 *        requested), naug_pad = plmc_pad(naug) >= 0, lda a multiple of NB.
 *        columns [0, n_pad)               : UPPER triangle holds Khat = K + noise*I, then its factor U
 *                                           (Khat = U^T U); padded rows/cols hold the identity.
 *        columns [n_pad, n_pad+naug_pad)  : augmented right-hand sides (targets, cross-covariances);
 *                                           potrf turns each column c into U^-T c (forward solve for free).
 *        columns [n_pad+naug_pad, +n_pad) : (with_inverse) W = U^-T, LOWER triangle, produced by the same
 *                                           sweep (explicit zeros above the diagonal inside diagonal
 *                                           blocks; blocks strictly above the diagonal are never touched).
 *     The strictly lower triangle of the square part is never read.
 *   - Vd: per latent plmc_vd_blocks(n_pad, lda) blocks of NB x NB: the first n_pad/NB are the inverses of the diagonal
 *     blocks of U (upper), the rest is workspace of the sweep (the inverse triangle of the current group of block
 *     rows, its transposed copies and the panel buffers of the group's block rows).
 *   - "W, ldw, strideW" arguments below: pointer to the first W column of latent 0 inside the factor
 *     buffer (A + n_pad + naug_pad), ldw = lda, strideW = strideA -- or any buffer of that layout.
 *   - kernel kinds: PLMC_RBF, PLMC_MATERN12, PLMC_MATERN32, PLMC_MATERN52 (stationary ARD kernels on x / ell) and
 *     PLMC_SPLINE (the reference's SplineKernel, projected_lmc.py:26-36: no lengthscale -- pass ell = 1; accepted by
 *     plmc_assemble_*, plmc_assemble_cross_* and plmc_kinv_grad_* only).
 */
#ifndef PLMC_H
#define PLMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PLMC_RBF = 0, PLMC_MATERN12 = 1, PLMC_MATERN32 = 2, PLMC_MATERN52 = 3, PLMC_SPLINE = 4 };

/* Library identification / geometry. */
int         plmc_version(void);
int         plmc_block(void);                 /* NB (128) */
int64_t     plmc_pad(int64_t n);              /* n rounded up to a multiple of NB */
int         plmc_max_dim(void);               /* largest input dimension d accepted by the fused kernels */
const char *plmc_last_error(void);            /* text of the last error on the calling thread */
/* NB x NB blocks per latent the `Vd` argument of plmc_potrf_* must hold, for elements of `elem_bytes` (4 or 8) bytes.  The
 * sizes depend on (n_pad, lda, elem_bytes) only, never on the dev knobs (version 3: for 4-byte elements they always include
 * the plane buffers of the split engine, on or off; version 4: and, when lda has room for the inverse factor
 * (lda >= 2 n_pad), the full-height planes of W that plmc_kinv_grad_vd_* reads, 6 n_pad^2 bytes).
 * plmc_vd_blocks(n_pad, lda) = the 4-byte count (the larger one: safe for both).
 * ABI note: versions 1-3 sized Vd smaller -- a caller built against them must re-query (check plmc_version() == 4). */
int64_t     plmc_vd_blocks_for(int64_t n_pad, int64_t lda, int elem_bytes);
int64_t     plmc_vd_blocks(int64_t n_pad, int64_t lda);
/* ... for a plmc_potrf_ex_f32 call with with_inverse | 4 (the sweep KEEPS the 16-bit planes of every group's solved rows
 * instead of rolling over two buffers: plmc_potrs_aug_kept_f32 needs them); 4-byte elements only.  Such a sweep always works
 * in groups of 8 block rows (it ignores the dev knob PLMC_GRP: one kept buffer per 8 block rows is what this size reserves and
 * what plmc_potrs_aug_kept_f32 walks).  plmc_kinv_grad_vd_f32 accepts either scratch: the sweep records its per-latent stride. */
int64_t     plmc_vd_blocks_keep(int64_t n_pad, int64_t lda);
/* Bytes of the `partials` scratch plmc_kinv_grad_* needs for (n_pad, q): per-tile partial sums, and for 4-byte elements the
 * bf16 planes of W (6 q n_pad^2 bytes).  plmc_grad_scratch_bytes = the 4-byte size. */
int64_t     plmc_grad_scratch_bytes_for(int64_t n_pad, int q, int elem_bytes);
int64_t     plmc_grad_scratch_bytes(int64_t n_pad, int q);
/* ... and what plmc_kinv_grad_vd_* needs (the partial sums only: the planes of W come from the sweep's Vd). */
int64_t     plmc_grad_partials_bytes(int64_t n_pad, int q);

/*
 * Covariance assembly.  Replaces `self.covar_module(x)` (:316, :1090; kernels built by
 * handle_covar_ :151-167) fused with `likelihood(dist)` = +noise*I (:1200):
 *   A[i][j] = oscale * k(|x_i - x_j| / ell) + noise * (i==j)   for i <= j < n  (upper tiles)
 * X: n x d, ell: q x d, oscale: q or NULL, noise: q.
 */
int plmc_assemble_f32(int kind, const float *X, int n, int d, const float *ell, const float *oscale,
                      const float *noise, float *A, int64_t lda, int64_t strideA, int q, void *stream);
int plmc_assemble_f64(int kind, const double *X, int n, int d, const double *ell, const double *oscale,
                      const double *noise, double *A, int64_t lda, int64_t strideA, int q, void *stream);

/*
 * Write right-hand sides into the augmented block: A[i][n_pad + c0 + r] = rhs[latent][r][i]
 * (rhs: q x nrhs x n, e.g. the projected targets of project_data :1014-1021, which are q x n).
 * Rows >= n are zero.  With clear_cols > 0 the other columns of [n_pad, n_pad + clear_cols) are
 * zeroed too (pass naug_pad to reset the whole augmented block).
 */
int plmc_write_rhs_f32(const float *rhs, int nrhs, int n, float *A, int64_t lda, int64_t strideA,
                       int c0, int clear_cols, int q, void *stream);
int plmc_write_rhs_f64(const double *rhs, int nrhs, int n, double *A, int64_t lda, int64_t strideA,
                       int c0, int clear_cols, int q, void *stream);

/*
 * Cross-covariance block (prediction, ProjectedGPModel.__call__ eval branch :1133-1134 -> gpytorch
 * prediction strategy):  Out[i][col0 + j] = oscale * k(x_i, xs_j)  for i < n, j < ns; rows
 * n <= i < n_rows are zero-filled.  To fill the augmented block of a factor buffer pass Out = A,
 * ldo = lda, col0 = n_pad + c0, n_rows = n_pad; any other dense (n_rows x ldo) buffer works too.
 */
int plmc_assemble_cross_f32(int kind, const float *X, int n, const float *Xs, int ns, int d,
                            const float *ell, const float *oscale, float *Out, int64_t ldo,
                            int64_t strideO, int64_t col0, int64_t n_rows, int q, void *stream);
int plmc_assemble_cross_f64(int kind, const double *X, int n, const double *Xs, int ns, int d,
                            const double *ell, const double *oscale, double *Out, int64_t ldo,
                            int64_t strideO, int64_t col0, int64_t n_rows, int q, void *stream);

/*
 * Blocked right-looking Cholesky of the augmented buffer, Khat = U^T U, in place.
 * Replaces torch.linalg.cholesky_ex + the forward triangular solve behind
 * `latent_output.log_prob(proj_target)` (:1201) / ExactMarginalLogLikelihood (experiments.py:233).
 *   logdet[latent] = log det Khat   (double), info[latent] = 0 or 1 + index of first non-PD pivot
 *   (PLMC_INFO_CHAIN_ABORT: the resident chain kernel of the sweep gave up a bounded wait for another workgroup -- an internal
 *   error, never a property of the matrix; the results are then undefined.  Never observed; PLMC_CHAIN=0 selects the
 *   launch-per-step chain).
 * naug = number of live augmented columns (0 allowed).
 * with_inverse != 0: also produce W = U^-T in columns [n_pad + naug_pad, n_pad + naug_pad + n_pad)
 * (the identity rides along as further right-hand sides) -- the first half of the Khat^-1 that
 * `loss.backward()` needs (experiments.py:270).  No initialisation of those
 * columns is required.
 * with_inverse == 2: additionally accumulate Khat^-1 = W^T W group by group while the sweep runs (the second half of
 * that work, as filler beside the latency-bound end of the factorisation): tile (ib < jb) of Khat^-1 is left in the
 * strictly LOWER triangle of the square part at block (jb, ib) -- never read otherwise -- and the diagonal tiles in the
 * last n_pad/NB blocks of Vd.  plmc_grad_tiles_* then turns it into the MLL gradient in one HBM-bound pass.
 */
#define PLMC_INFO_CHAIN_ABORT 0x7ffffff0
int plmc_potrf_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, float *Vd,
                   double *logdet, int *info, int with_inverse, int q, void *stream);
int plmc_potrf_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, double *Vd,
                   double *logdet, int *info, int with_inverse, int q, void *stream);
/* The same with `eig_lo`: q lower bounds of the smallest eigenvalue of the input matrices (device pointer; for a GP
 * covariance K + s2 I the noise variances s2, since K is positive semi-definite).  Arithmetic of the BULK fp32 products
 * (tail / head updates, group panel; environment PLMC_SPLIT, default 2):
 *   2  two fp16 planes per operand, s x = h0 + 2^-11 h1 (22 bits), three plane products on v_mfma_f32_16x16x32_f16 into two
 *      fp32 accumulator levels.  Every operand family is scaled by a power of two that puts a rigorous bound of its
 *      magnitude -- from the largest diagonal entry D and eig_lo: sqrt(D), D, 1/sqrt(eig_lo), sqrt(D/eig_lo) -- at 2^13, so fp16
 *      cannot overflow; the augmented columns (no a-priori bound) stay on the fp32 MFMAs.  Needs eig_lo; without it ->
 *   3  three bf16 planes (x = hi + mid + lo exactly), six plane products, two accumulator levels: any data, no scaling.
 *   0  v_mfma_f32_16x16x4_f32 everywhere.
 * Either split has 0.3-0.45 x the error of the fp32 MFMA chain against fp64 (profiles/r03_split_numerics.txt).  The chain
 * (diagonal blocks, rank-128 updates), the panel columns of the next group and every fp64 product use the MFMA of the
 * element type.  A wrong (too large) eig_lo can overflow fp16: the result is then inf / nan and `info` reports it. */
int plmc_potrf_ex_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, float *Vd, double *logdet,
                      int *info, int with_inverse, int q, const float *eig_lo, void *stream);
int plmc_potrf_ex_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, double *Vd, double *logdet,
                      int *info, int with_inverse, int q, const double *eig_lo, void *stream);

/* Assembly + factorisation in one call (round 4): plmc_assemble_* followed by plmc_potrf_ex_* on the same buffers, with the
 * assembly overlapped with the sweep -- the sweep writes the rows of its first group of block rows on the caller's stream and
 * queues the other rows (and the scan of the diagonal for the fp16 scales, which needs them) on a helper stream beside that
 * group's chain; nothing reads them before.  The augmented columns must be in place BEFORE the call (plmc_write_rhs_*,
 * plmc_assemble_cross_*).  n_pad = plmc_pad(n).  Same kernels on the same data as the two separate calls: bit-identical results
 * (tests/test_gpu_engine.py).  Replaces, like them: `self.covar_module(x)` + `likelihood(dist)` + the Cholesky inside `log_prob`
 * (projected_lmc.py:316,:1090,:1200-1201). */
int plmc_factorize_ex_f32(int kind, const float *X, int n, int d, const float *ell, const float *oscale, const float *noise,
                          float *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, float *Vd, double *logdet, int *info,
                          int with_inverse, int q, const float *eig_lo, void *stream);
int plmc_factorize_ex_f64(int kind, const double *X, int n, int d, const double *ell, const double *oscale, const double *noise,
                          double *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, double *Vd, double *logdet, int *info,
                          int with_inverse, int q, const double *eig_lo, void *stream);

/*
 * Forward substitution of NEW augmented columns against a buffer that plmc_potrf_* already factorised with
 * with_inverse != 0 (and has not been modified since):  columns [n_pad, n_pad + naug) <- U^-T columns.  wcol0 = first
 * inverse-factor column of that factorisation (n_pad + plmc_pad(naug of the factorisation)); naug here may be smaller.
 * What an eval-mode model does on its second and later calls (projected_lmc.py:1133-1134: gpytorch's prediction strategy
 * keeps the factorisation across calls): n^2 naug flops instead of a new sweep.  Vd: the scratch of the factorisation
 * (its group workspace is reused).
 */
int plmc_potrs_aug_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, float *Vd, int q, void *stream);
int plmc_potrs_aug_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, double *Vd, int q, void *stream);
/* The same against a factorisation that kept its planes -- plmc_potrf_ex_f32(..., with_inverse | 4, ...) into a Vd of
 * plmc_vd_blocks_keep(n_pad, lda) blocks per latent: the group panels and the depth-1024 updates of the substitution then run on
 * the split engine too (A operand = the kept planes of the factor's rows; 2-3 x the fp32 MFMA rate of plmc_potrs_aug_f32).
 * eig_lo: NULL or not as in the factorising call (it selects the scheme; the values are not used).  fp64 and PLMC_SPLIT=0:
 * exactly plmc_potrs_aug_*. */
int plmc_potrs_aug_kept_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, float *Vd, int q,
                            const float *eig_lo, void *stream);
int plmc_potrs_aug_kept_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, double *Vd, int q,
                            const double *eig_lo, void *stream);

/*
 * Gather augmented column c of every latent into a contiguous vector z (q x n_pad) and return
 * quad[latent] = sum z^2 in double (the inv_quad term of MVN.log_prob, :1201).
 */
int plmc_extract_col_f32(const float *A, int64_t n_pad, int64_t lda, int64_t strideA, int c,
                         float *z, double *quad, int q, void *stream);
int plmc_extract_col_f64(const double *A, int64_t n_pad, int64_t lda, int64_t strideA, int c,
                         double *z, double *quad, int q, void *stream);

/* alpha = W^T z = Khat^-1 y  (q x n_pad).  d logp / d y = -alpha. */
int plmc_wt_matvec_f32(const float *W, int64_t n_pad, int64_t ldw, int64_t strideW, const float *z,
                       float *alpha, int q, void *stream);
int plmc_wt_matvec_f64(const double *W, int64_t n_pad, int64_t ldw, int64_t strideW, const double *z,
                       double *alpha, int q, void *stream);

/*
 * MLL gradient from the Khat^-1 a sweep with with_inverse = 2 accumulated (same outputs as plmc_kinv_grad_*: grad, and
 * optionally kinv_diag; partials: the same scratch): (alpha alpha^T - Khat^-1) o dKhat/dtheta reduced per tile from X in
 * LDS, one read of the stored tiles.  A, Vd: the buffers of that sweep.
 */
int plmc_grad_tiles_f32(int kind, const float *A, int64_t n_pad, int64_t lda, int64_t strideA, const float *Vd,
                        const float *alpha, const float *X, int n, int d, const float *ell, const float *oscale,
                        double *grad, float *kinv_diag, void *partials, int q, void *stream);
int plmc_grad_tiles_f64(int kind, const double *A, int64_t n_pad, int64_t lda, int64_t strideA, const double *Vd,
                        const double *alpha, const double *X, int n, int d, const double *ell, const double *oscale,
                        double *grad, double *kinv_diag, void *partials, int q, void *stream);

/*
 * Batched "TN" product on the tile engine:  C[b] (=, +=, -=) A[b]^T B[b]  with A: K x M and B: K x N, both stored
 * K-major (row index = contraction index), C: M x N; mode 0 store / 1 add / 2 subtract.  M, N multiples of plmc_block(),
 * K a multiple of 16 (callers pad with zeros), 16-byte aligned pointers / strides.  Replaces the torch.matmul (rocBLAS)
 * calls of the Cholesky adjoint of the inducing-point interpolation (VariationalMultitaskGPModel :672-683, SGPR :302-303).
 */
int plmc_gemm_tn_f32(int mode, int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B,
                     int64_t ldb, int64_t strideB, float *C, int64_t ldc, int64_t strideC, int batch, void *stream);
int plmc_gemm_tn_f64(int mode, int M, int N, int K, const double *A, int64_t lda, int64_t strideA, const double *B,
                     int64_t ldb, int64_t strideB, double *C, int64_t ldc, int64_t strideC, int batch, void *stream);
/* The same for operands with TRIANGULAR structure (the factors W = L^-1, L^T = U, Ls and the triangular adjoints of the
 * Cholesky adjoint): `tri` = OR of the bits below; a tile then only walks the contraction range in which both operands have
 * entries -- the skipped terms are exact zeros, the result is the same number at a half to a sixth of the flops.  Structure is
 * declared per 128-block (entries inside a diagonal block may be anything).
 *   PLMC_TRI_A_LOWER  A[k][i] = 0 for k < 128 (i / 128)        PLMC_TRI_B_LOWER  B[k][j] = 0 for k < 128 (j / 128)
 *   PLMC_TRI_A_UPPER  A[k][i] = 0 for k >= 128 (i / 128 + 1)
 *   PLMC_TRI_C_LOWER  only tiles with i / 128 >= j / 128 are computed; the others are left untouched, or, with PLMC_TRI_C_ZERO
 *                     as well (mode 0 only), written as zeros. */
#define PLMC_TRI_A_LOWER 1
#define PLMC_TRI_B_LOWER 2
#define PLMC_TRI_A_UPPER 4
#define PLMC_TRI_C_LOWER 8
#define PLMC_TRI_C_ZERO 16
int plmc_gemm_tn_tri_f32(int mode, int tri, int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B,
                         int64_t ldb, int64_t strideB, float *C, int64_t ldc, int64_t strideC, int batch, void *stream);
int plmc_gemm_tn_tri_f64(int mode, int tri, int M, int N, int K, const double *A, int64_t lda, int64_t strideA, const double *B,
                         int64_t ldb, int64_t strideB, double *C, int64_t ldc, int64_t strideC, int batch, void *stream);

/*
 * Eval-mode posterior, the reductions behind the augmented sweep (ProjectedGPModel.__call__, projected_lmc.py:1133-1155;
 * gpytorch's DefaultPredictionStrategy behind ExactGPModel.__call__ :1134).
 * plmc_posterior_moments: from the factor buffer after plmc_potrf_*(with_inverse = 0) on [ Khat | y | K*^T ]
 *   (naug = 1 + ns): mean[latent][s] = v(s)^T z and vsq[latent][s] = |v(s)|^2 with z = column n_pad, v(s) = column
 *   n_pad + 1 + s; the latent posterior is (mean, k(x*, x*) - vsq).  mean, vsq: q x ns, contiguous.
 * plmc_mix_posterior: task-space moments of :1144 / :1152, mean[s][t] = sum_i mean_lat[i][s] Ht[i][t],
 *   var[s][t] = sum_i var_lat[i][s] Ht[i][t]^2 + eps.  mean_lat, var_lat: q x ns; Ht: q x p; outputs ns x p.  With latent
 *   sharding a rank passes its local latents (q = their number) and eps = 0, and adds eps after the all-reduce.
 * Both HBM-bound; sums carried in fp64.
 */
int plmc_posterior_moments_f32(const float *A, int64_t n_pad, int64_t lda, int64_t strideA, int ns, float *mean,
                               float *vsq, int q, void *stream);
int plmc_posterior_moments_f64(const double *A, int64_t n_pad, int64_t lda, int64_t strideA, int ns, double *mean,
                               double *vsq, int q, void *stream);
int plmc_mix_posterior_f32(const float *mean_lat, const float *var_lat, const float *Ht, int q, int ns, int p,
                           double eps, float *mean, float *var, void *stream);
int plmc_mix_posterior_f64(const double *mean_lat, const double *var_lat, const double *Ht, int q, int ns, int p,
                           double eps, double *mean, double *var, void *stream);

/*
 * kinv_diag = diag(Khat^-1) = column sums of squares of W (q x n_pad): the leave-one-out variances
 * sigma2_i = 1 / [Khat^-1]_ii of `MultitaskGPModel.compute_loo` (:642-656) on the dense (n p) x (n p) system, where
 * the fused gradient kernel (single ARD kernel per matrix) does not apply.  HBM-bound, reads W once.
 */
int plmc_w_diag_f32(const float *W, int64_t n_pad, int64_t ldw, int64_t strideW, float *kinv_diag, int q,
                    void *stream);
int plmc_w_diag_f64(const double *W, int64_t n_pad, int64_t ldw, int64_t strideW, double *kinv_diag, int q,
                    void *stream);

/*
 * Fused K^-1 = W^T W (MFMA) + analytic MLL gradient reduction: for every upper tile of K^-1 the
 * epilogue forms (alpha alpha^T - K^-1) o dKhat/dtheta from X staged in LDS and reduces it.
 *   grad[latent][0..d-1] = d logp / d ell_k,  [d] = d/d noise,  [d+1] = d/d oscale   (double)
 * Optional outputs (may be NULL): Kinv (n_pad x ldk upper tiles, batch stride strideK),
 * kinv_diag (q x n_pad: diagonal of K^-1, for leave-one-out, compute_loo :1108-1119).
 * partials: scratch of plmc_grad_scratch_bytes_for(n_pad, q, sizeof element) bytes.
 * fp32 entry point, PLMC_SPLIT != 0: the W^T W products run on the 16-bit matrix cores from a split copy of W (planes
 * behind the partials; arithmetic as documented at plmc_potrf_ex_f32).
 */
int plmc_kinv_grad_f32(int kind, const float *W, int64_t n_pad, int64_t ldw, int64_t strideW,
                       const float *alpha, const float *X, int n, int d, const float *ell,
                       const float *oscale, double *grad, float *Kinv, int64_t ldk, int64_t strideK,
                       float *kinv_diag, void *partials, int q, void *stream);
int plmc_kinv_grad_f64(int kind, const double *W, int64_t n_pad, int64_t ldw, int64_t strideW,
                       const double *alpha, const double *X, int n, int d, const double *ell,
                       const double *oscale, double *grad, double *Kinv, int64_t ldk, int64_t strideK,
                       double *kinv_diag, void *partials, int q, void *stream);
/* The same with `eig_lo`: q lower bounds of the smallest eigenvalue of Khat (device pointer; for a GP covariance K + s2 I
 * the noise variances s2).  With them the fp32 entry point may run its W^T W products as the two-plane fp16 split (see
 * plmc_potrf_ex_f32); NULL or the plain entry point: three-plane bf16 split.  The fp64 entry point ignores it. */
int plmc_kinv_grad_ex_f32(int kind, const float *W, int64_t n_pad, int64_t ldw, int64_t strideW,
                          const float *alpha, const float *X, int n, int d, const float *ell,
                          const float *oscale, double *grad, float *Kinv, int64_t ldk, int64_t strideK,
                          float *kinv_diag, void *partials, int q, const float *eig_lo, void *stream);
int plmc_kinv_grad_ex_f64(int kind, const double *W, int64_t n_pad, int64_t ldw, int64_t strideW,
                          const double *alpha, const double *X, int n, int d, const double *ell,
                          const double *oscale, double *grad, double *Kinv, int64_t ldk, int64_t strideK,
                          double *kinv_diag, void *partials, int q, const double *eig_lo, void *stream);
/* The same for a W that is still where its sweep left it: W = the inverse-factor columns of the factor buffer (ldw = that
 * buffer's lda), Vd = the scratch of that plmc_potrf_ex_* call (with_inverse != 0), untouched since.  The sweep's
 * epilogues have already written the 16-bit planes of W into Vd (the rows of a group the moment they are final), so the
 * split pass over W of the entry points above is skipped (0.6 ms of the 19 ms step at the metric shape) and `partials`
 * only needs plmc_grad_partials_bytes(n_pad, q).  Both calls must see the same PLMC_SPLIT and both or neither an eig_lo:
 * the scratch records which scheme wrote it and a mismatch yields NaN gradients, not silently wrong ones.  Vd = NULL, fp64,
 * PLMC_SPLIT=0 or a layout without inverse-factor columns: exactly plmc_kinv_grad_ex_* (then `partials` must have the
 * full plmc_grad_scratch_bytes_for size). */
int plmc_kinv_grad_vd_f32(int kind, const float *W, int64_t n_pad, int64_t ldw, int64_t strideW,
                          const float *alpha, const float *X, int n, int d, const float *ell,
                          const float *oscale, double *grad, float *Kinv, int64_t ldk, int64_t strideK,
                          float *kinv_diag, void *partials, int q, const float *eig_lo, const float *Vd, void *stream);
int plmc_kinv_grad_vd_f64(int kind, const double *W, int64_t n_pad, int64_t ldw, int64_t strideW,
                          const double *alpha, const double *X, int n, int d, const double *ell,
                          const double *oscale, double *grad, double *Kinv, int64_t ldk, int64_t strideK,
                          double *kinv_diag, void *partials, int q, const double *eig_lo, const double *Vd, void *stream);

/*
 * The one exchange of the sharded path, for a host without torch.distributed (the Python layer's default
 * is torch.distributed "nccl" = RCCL, `projectedlmc/parallel.py`, which can be switched to these with PLMC_COMM=rccl):
 * a direct RCCL all-reduce (sum, in place) of the fused [loss share | parameter gradients] buffer after backward
 * (experiments.py:270-272 on every rank) and of the (2, n*, p) partial prediction sums (projected_lmc.py:1144,1152).
 * RCCL is opened at run time (dlopen; no link-time dependency).  One communicator per process, bound to the device that is
 * current at plmc_comm_init.  Bootstrap: rank 0 calls plmc_comm_unique_id (128 bytes) and the host's launcher carries the
 * bytes to the other ranks (environment, file, MPI, a torch.distributed broadcast); then every rank calls plmc_comm_init.
 * plmc_comm_allreduce_sum_* is asynchronous on `stream`.  Returns 0 or a negative code (plmc_last_error()).
 */
int plmc_comm_unique_id(void *id128);
int plmc_comm_init(const void *id128, int rank, int world);
int plmc_comm_world(void);                    /* 0 without a communicator */
int plmc_comm_rank(void);                     /* -1 without a communicator */
int plmc_comm_allreduce_sum_f32(float *buf, int64_t count, void *stream);
int plmc_comm_allreduce_sum_f64(double *buf, int64_t count, void *stream);
int plmc_comm_destroy(void);

/*
 * Exact (dense) LMC / ICM: Kronecker-structured coregionalisation.
 * Replaces `MultitaskGPModel.forward` -> gpytorch LCMKernel / MultitaskKernel (:462-466, :586-589)
 * + MultitaskGaussianLikelihood (experiments.py:184) and their autograd backward:
 *     K_full = sum_i oscale_i k(X, X; ell_i) (x) B_i  +  I_n (x) Sigma      N = n*p rows, data-major
 *     interleaved (flat index = i_point * p + i_task), B: q x p x p, Sigma: p x p (both symmetric).
 * plmc_lmc_assemble_*  writes the upper tiles (identity padding up to plmc_pad(N)) of ONE factor buffer
 *                      (batch 1); it is then factorised by plmc_potrf_* like any other matrix.
 * plmc_lmc_cross_*     writes K_full(X, X*) (without Sigma) into columns [col0, col0 + ns*p).
 * plmc_lmc_kinv_grad_* fused K_full^-1 = W^T W + gradient reduction; grad (fp64) has
 *                      plmc_lmc_grad_len(p,q,d) entries laid out as
 *                        [ dB: q*p*p | d ell: q*d | d oscale: q | d Sigma: p*p ]
 *                      where dB / dSigma are accumulated over the UPPER triangle of K_full only
 *                      (weight 2 off the diagonal): the caller symmetrises them, (G + G^T) / 2.
 *                      partials: scratch of plmc_lmc_grad_scratch_bytes(N_pad, p, q, d) bytes.
 */
int64_t plmc_lmc_grad_len(int p, int q, int d);
int64_t plmc_lmc_grad_scratch_bytes(int64_t N_pad, int p, int q, int d);
int plmc_lmc_assemble_f32(int kind, const float *X, int n, int d, int p, int q, const float *ell,
                          const float *oscale, const float *B, const float *Sigma, float *A,
                          int64_t lda, void *stream);
int plmc_lmc_assemble_f64(int kind, const double *X, int n, int d, int p, int q, const double *ell,
                          const double *oscale, const double *B, const double *Sigma, double *A,
                          int64_t lda, void *stream);
int plmc_lmc_cross_f32(int kind, const float *X, int n, const float *Xs, int ns, int d, int p, int q,
                       const float *ell, const float *oscale, const float *B, float *Out, int64_t ldo,
                       int64_t col0, int64_t n_rows, void *stream);
int plmc_lmc_cross_f64(int kind, const double *X, int n, const double *Xs, int ns, int d, int p, int q,
                       const double *ell, const double *oscale, const double *B, double *Out,
                       int64_t ldo, int64_t col0, int64_t n_rows, void *stream);
int plmc_lmc_kinv_grad_f32(int kind, const float *W, int64_t N_pad, int64_t ldw, const float *alpha,
                           const float *X, int n, int d, int p, int q, const float *ell,
                           const float *oscale, const float *B, double *grad, void *partials,
                           void *stream);
int plmc_lmc_kinv_grad_f64(int kind, const double *W, int64_t N_pad, int64_t ldw, const double *alpha,
                           const double *X, int n, int d, int p, int q, const double *ell,
                           const double *oscale, const double *B, double *grad, void *partials,
                           void *stream);

/*
 * Vector-Jacobian product of a dense (cross-)covariance block K_i[a][b] = oscale_i k(|x1_a - x2_b| / ell_i)
 * given G = d loss / d K (q x n1 x n2, row stride ldg, batch stride strideG).  Row partials in fp64:
 *   gX1 [q][n1][d] = sum_b G dK/dx1_a,   gEll[q][n1][d] = sum_b G dK/d ell  (caller sums over rows),
 *   gOs [q][n1]    = sum_b G dK/d oscale.
 * Variational path: gradients of K_ZZ / K_ZX w.r.t. the learned inducing
 * locations and lengthscales of VariationalMultitaskGPModel (projected_lmc.py:686-766), which the
 * reference obtains from torch autograd through gpytorch's kernel evaluation (experiments.py:270).
 */
int plmc_kernel_vjp_f32(int kind, const float *X1, int n1, const float *X2, int n2, int d,
                        const float *ell, const float *oscale, const float *G, int64_t ldg,
                        int64_t strideG, double *gX1, double *gEll, double *gOs, int q, void *stream);
int plmc_kernel_vjp_f64(int kind, const double *X1, int n1, const double *X2, int n2, int d,
                        const double *ell, const double *oscale, const double *G, int64_t ldg,
                        int64_t strideG, double *gX1, double *gEll, double *gOs, int q, void *stream);

/*
 * Reduced Householder QR of ONE small matrix A (m x n row-major, plmc_qr_max() >= m >= n >= 1) in a single
 * launch: Q (m x n, orthonormal columns), R (n x n upper triangular, the lower part is written as zeros).
 * Replaces `torch.linalg.qr(self.H)` of LMCMixingMatrix.QR in bulk mode (projected_lmc.py:864-875, called at
 * :1015 and :1208) -- on the device ~45 dependent rocSOLVER launches per training step.  LAPACK xGEQR2 / xORG2R
 * sign convention (R_kk = -sign(a_kk) |a_k:m,k|; a column that is already zero below the diagonal keeps its sign),
 * i.e. the factors the reference's torch.linalg.qr returns, to rounding.
 */
int plmc_qr_max(void);
int plmc_qr_small_f32(const float *A, int m, int n, int64_t lda, float *Q, int64_t ldq, float *R, int64_t ldr,
                      void *stream);
int plmc_qr_small_f64(const double *A, int m, int n, int64_t lda, double *Q, int64_t ldq, double *R, int64_t ldr,
                      void *stream);

/*
 * Optional per-kernel profiler (measurement only; the reference's counterpart is the wall-clock
 * `time.time()` around its loops, experiments.py:261,284).  While enabled, kernel launches are
 * bracketed by two hipEvents on their launch stream: `on` = 0 none, 1 every kernel class, any other
 * value = (bit mask of class indices) << 1, so that a timed run can bracket the few heavy kernels
 * only (each bracket costs ~10 us of stream time, which matters on the latency-bound chain).  plmc_prof_collect() waits for the recorded
 * events, then returns per kernel class (index < plmc_prof_kernels(), name plmc_prof_name(i)):
 * total milliseconds, number of launches, and the ALGORITHMIC flops / bytes of those launches;
 * it clears the record.
 * Process-global state of the library (all of it): this profiler record; per device, TWO sets of three helper streams and
 * sixteen ordering events for the look-ahead in plmc_potrf_* (created on first use, never destroyed), each bound to the
 * caller stream that used it last.  Sweeps queued on one stream share a set (stream order protects it); sweeps from two
 * streams get a set each and may overlap on the device (their buffers must differ); a third stream takes over the least
 * recently used set and waits (stream-side, an event) for that set's last sweep.  Calls from several host THREADS are
 * serialised by one library-wide lock while they enqueue (the kernels overlap on the device as their streams allow); each
 * thread keeps its own choice of sweep set.  The Python layer calls from one thread per process.
 * Dev knobs are environment variables read once per process (PLMC_HALF_TILES, PLMC_GRP, PLMC_KINV_ORDER,
 * PLMC_SERIAL, PLMC_BULK_LDS, PLMC_CHAIN, PLMC_CHAIN_NW, PLMC_CHAIN_EDGE); plmc_dev_reload_knobs() re-reads them (tests and bench.py change one and reload).  They change
 * schedules; PLMC_GRP also changes the depth of the updates and with it the rounding.  PLMC_SPLIT (0, 2, 3) selects the
 * arithmetic of the bulk fp32 products (see plmc_potrf_ex_f32).  Buffer sizes do not depend on any knob.
 * plmc_prof_mfma_rate: dense MFMA rate (TFLOP/s) of the current device measured with a bare instruction stream
 * (v_mfma_f32_16x16x4_f32, _f64_16x16x4_f64 or v_mfma_f32_16x16x32_bf16, 4 waves per SIMD, no memory traffic); synchronous; `sink` = caller-owned
 * device scratch of at least 4 * CUs * 256 * sizeof(element) bytes.
 */
int         plmc_prof_enable(int on);         /* returns the previous setting */
int         plmc_prof_kernels(void);
const char *plmc_prof_name(int id);
int         plmc_prof_collect(double *ms, int64_t *launches, double *flops, double *bytes);
int         plmc_prof_mfma_rate(int kind, void *sink, int64_t sink_bytes, double *tflops);   /* kind: 0 f32, 1 f64, 2 bf16 (16x16x32) */
int         plmc_dev_reload_knobs(void);

#ifdef __cplusplus
}
#endif
#endif /* PLMC_H */
