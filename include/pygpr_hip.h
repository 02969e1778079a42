/*
 * libpygpr_hip -- C ABI of the MI355X (gfx950) dense Gaussian-process hot path.
 *
 * The reference (sarath-srinivas/PyGPR, pure Python on torch CPU) has no FFI: its boundary for this
 * path is the Python class surface, and every entry point below replaces the torch/LAPACK call (or the
 * Python expression) cited next to it.  pygpr_amd/_lib.py binds these with ctypes; INTEGRATION.md
 * shows the same stub as a maintainer of the reference would add it.
 *
 * Conventions
 *   - every matrix/vector pointer is a DEVICE pointer owned by the caller (torch-ROCm allocations);
 *     the library allocates nothing but the streams/events inside a pg_handle
 *   - matrices are row-major with a leading dimension in elements; dtype is PG_F64 or PG_F32
 *   - the O(n^3) entry points need n_pad % 256 == 0; pg_kernel_build fills the padding with identity
 *     (symmetric build) or zeros (cross build), so padded factors are block-diag(L, I)
 *   - calls are asynchronous on `stream`; return 0 = enqueued, <0 = bad argument / HIP error
 *     (text via pg_last_error()); numerical failure (non-PD pivot) is reported LAPACK-style in a
 *     device int `info` that the caller reads after synchronising
 *   - `hp` is the PyGPR hyper-parameter vector (fp64, device) in Compose order (covar.py:50-55);
 *     pg_covspec says where each child's parameters sit in it
 */
#ifndef PYGPR_HIP_H
#define PYGPR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_F64 0
#define PG_F32 1
#define PG_KIND_RBF 0      /* Squared_exponential, covar.py:84-206 */
#define PG_KIND_MATERN52 1 /* new (not in the reference), same hp layout */
#define PG_KIND_SQDIST 2   /* pg_kernel_build only: the scaled squared distance itself, sum_k l_k^2 (x_k - x'_k)^2 --
                              Squared_exponential.distance (covar.py:102-127) when l = 1 */
#define PG_MAX_COMP 4
#define PG_MAX_DIM 64

typedef struct pg_ctx* pg_handle;

/* A Compose([...]) of up to PG_MAX_COMP stationary kernels plus white-noise terms (covar.py:28-81).  A longer Compose is
 * evaluated in passes of PG_MAX_COMP children (pg_kernel_build's `accumulate`; the gradient entries of different
 * children are disjoint, so pg_nlml_grad / pg_kernel_grad_build passes simply write different entries). */
typedef struct pg_covspec {
    int ncomp;                  /* stationary components                                     */
    int kind[PG_MAX_COMP];      /* PG_KIND_*                                                 */
    int off[PG_MAX_COMP];       /* index of [sigma, l_1..l_d] of component c inside hp       */
    int nnoise;                 /* White_noise children (covar.py:209-269)                   */
    int noise_off[PG_MAX_COMP]; /* index of each sigma_n inside hp                           */
} pg_covspec;

int pg_version(void);
const char* pg_last_error(void);
/* A handle owns three HIP streams (panel / rows / update; the update stream is CU-masked) and a pool of events, and works from any
 * caller stream, the legacy default one included.  Release the handle with pg_destroy when
 * done; handles still alive at process exit are destroyed by the library itself (a C atexit handler registered by the
 * first pg_create, so it runs before the HIP runtime's own teardown).  pg_destroy on an already released handle is a no-op. */
int pg_create(pg_handle* h);
int pg_destroy(pg_handle h);

/* Covariance assembly.  Replaces Compose/Squared_exponential/White_noise.kernel (covar.py:50-62,
 * 129-167, 227-245).  Xc == NULL: symmetric build on Xr (K[i][j], i,j < nr; diagonal gets
 * sum(sigma_n^2) + jitter as in gpr.py:68 / loss.py:38); lower_only skips tiles above the diagonal.
 * Otherwise a cross build K[i][j] = k(Xr_i, Xc_j) (covar.py:152-161; no noise term, covar.py:243).
 * accumulate != 0: K += (the sum over children of Compose.kernel, covar.py:57-60, continued in a further pass;
 * the padding is then left alone).  d <= PG_MAX_DIM. */
int pg_kernel_build(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* Xr, long ldr,
                    int nr, const void* Xc, long ldc, int nc, int d, int lower_only, int accumulate, double jitter,
                    void* K, long ldk, int rows_pad, int cols_pad, void* stream);

/* The same for nexp experts of one shape in ONE launch (round 5; no accumulate pass): expert e builds K + e * k_stride from the row points
 * Xr + e * xr_stride, the column points Xc + e * xc_stride and the hyper-parameters hp + e * hp_stride (strides in elements; 0 shares an
 * operand).  Xc == NULL: symmetric builds on Xr (what pg_build_potrf_trtri_batched folds in).  A cross build with xr_stride = 0 is the
 * reference's batched K* = cov.kernel(params [nc, nhp], x [nc, n, d], xp [m, d]) (gpr.py:79 on the batched x of gr_bcm.py:19-29), stored
 * test-point-major; per expert the same kernel and numbers as pg_kernel_build. */
int pg_kernel_build_batched(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, long hp_stride, const void* Xr, long ldr,
                            long xr_stride, int nr, const void* Xc, long ldc, long xc_stride, int nc, int d, int lower_only, double jitter,
                            void* K, long ldk, long k_stride, int rows_pad, int cols_pad, int nexp, void* stream);

/* dK[nhp][n][n] (contiguous, unpadded): the stack Covar.kernel_and_grad returns (covar.py:64-81,
 * 169-206, 247-269).  Drop-in surface only -- the NLML path never materialises it. */
int pg_kernel_grad_build(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* X, long ldx,
                         int n, int d, void* dK, void* stream);

/* Lower Cholesky in place, replaces tc.cholesky (gpr.py:69, loss.py:39,64,97).  inv_diag is the factorisation's
 * workspace of pg_potrf_worksize(dtype, n) elements: its first n * 128 elements receive the inverses of the 128x128
 * diagonal blocks ([n/128][128][128]) that drive every later solve (pg_potrs_vec, pg_trtri ... only read that part);
 * the rest (n + 2048 elements) is scratch of the call: the flag words through which the resident kernels of the coupled
 * chain hand over (csrc/chainstep.hip; zeroed by every call); with PG_PANEL_MODE=1 in the environment (experimental recursive
 * panel step) its buffers follow.
 * info: 0, or j + 1 when the leading minor of order j + 1 is not positive definite (LAPACK's convention), or -1 when a bounded
 * wait inside the coupled chain expired: NOT a property of the matrix -- kernels of the handle's panel and rows streams did not run
 * at the same time (seen while developing: under `rocprofv3 --pmc`, which runs one kernel at a time, before pg_create probed for
 * that; and with a schedule variant that put the leaf and the rows on two CU-masked streams, since removed).  A and Minv are then
 * garbage, and the handle has switched itself to the classic chain (pg_coupled_chain() == 0, pg_chain_timeouts() counts): supply
 * the matrix again and repeat the call, or use pg_build_potrf_trtri_checked, which does that inside the call.
 * Pointers: A (and any C operand of pg_gemm_raw / pg_syrk_tn_sub) may be any device-accessible memory; the no-return fp64 atomic
 * epilogue of the updates is only used when hipPointerGetAttributes says plain device memory (hipMalloc / torch-ROCm). */
long pg_potrf_worksize(int dtype, int n);   /* elements of inv_diag */
int pg_potrf(pg_handle h, int dtype, int n, void* A, long lda, void* inv_diag, int* info, void* stream);

/* pg_potrf followed by pg_trtri in one call: Minv = L^-1 as well (tc.cholesky + the triangular half of the
 * cholesky_solve calls of loss.py:97,116). */
int pg_potrf_trtri(pg_handle h, int dtype, int n, void* A, long lda, void* inv_diag, int* info, void* Minv, long ldm,
                   void* stream);

/* pg_kernel_build(lower_only = 1, symmetric, one pg_covspec) folded into pg_potrf (Minv == NULL) / pg_potrf_trtri: K + jitter I
 * -> A, then its factor in place (covar.py:50-62 + gpr.py:67-69 / loss.py:38-39,63-64,96-97 in one call).  The first outer panel's
 * columns are built before the factorisation's look-ahead starts, the rest on the update stream while the first panel's chain
 * runs (that chain has the chip to itself otherwise).  Same A and factor as the two separate calls. */
int pg_build_potrf_trtri(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* X, long ldx, int n, int d,
                         double jitter, void* A, long lda, int n_pad, void* inv_diag, int* info, void* Minv, long ldm,
                         void* stream);
/* The same, BLOCKING: waits for the factorisation on `stream`, copies *info to *info_host, and if a wait of the coupled chain
 * expired (info = -1) repeats the whole call on the classic chain before it returns -- tc.cholesky (gpr.py:69) never fails on a
 * positive-definite matrix, and neither does this.  *info_host < 0 on return only if the repeat failed as well. */
int pg_build_potrf_trtri_checked(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* X, long ldx, int n,
                                 int d, double jitter, void* A, long lda, int n_pad, void* inv_diag, int* info, void* Minv, long ldm,
                                 void* stream, int* info_host);

/* Batched experts: nexp independent problems of the same size in ONE call -- the leading batch dimension of the reference's
 * Exact_GP (gpr.py:65-74 factorises all experts of a GRBCM in one batched tc.cholesky, gr_bcm.py:19-29).  Expert e uses
 * hp + e * hp_stride, X + e * x_stride (a stride of 0 shares the points), A + e * a_stride, inv_diag + e * inv_stride
 * (inv_stride >= pg_potrf_worksize), info[e], Minv + e * m_stride (Minv may be NULL: factor only).  X == NULL: A already holds
 * the matrices (lower triangles).  Every launch of the blocked algorithm covers all experts (grid.y = expert), so the latency-bound
 * 128-column steps of the chain cost one launch triple per batch instead of one per expert; the per-expert numbers are those of
 * pg_build_potrf_trtri on the classic chain, bit for bit. */
int pg_build_potrf_trtri_batched(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, long hp_stride, const void* X, long ldx,
                                 long x_stride, int n, int d, double jitter, void* A, long lda, long a_stride, int n_pad, void* inv_diag,
                                 long inv_stride, int* info, void* Minv, long ldm, long m_stride, int nexp, void* stream);

/* x = K^-1 y from the factor: the cholesky_solve of gpr.py:70-72 / loss.py:45.  y is not modified;
 * work: pg_potrs_vec_worksize(dtype, n) elements (2 n below n = 2048; above, also the 1024-wide diagonal block
 * inverses the blocked sweeps multiply with). */
long pg_potrs_vec_worksize(int dtype, int n);
int pg_potrs_vec(pg_handle h, int dtype, int n, const void* L, long ldl, const void* inv_diag, const void* y, void* x,
                 void* work, void* stream);

/* Matrix right-hand sides: tc.cholesky_solve(B, L) of gpr.py:100,112 / loss.py:116 (pg_potrs: X = K^-1 B) and its triangular half
 * (pg_trsm_lower: X = L^-1 B).  B [n x nrhs] and X [n x nrhs] row-major, nrhs a multiple of 128 (pad with zero columns), X != B.
 * Both run as products of the MFMA GEMM core against L^-1: pass Minv (from pg_trtri / pg_potrf_trtri) or NULL, in which case
 * pg_trtri forms it in `work` first.  work: pg_potrs_worksize(dtype, n, nrhs, Minv != NULL) elements. */
long pg_potrs_worksize(int dtype, int n, int nrhs, int have_minv);
int pg_potrs(pg_handle h, int dtype, int n, int nrhs, const void* L, long ldl, const void* inv_diag, const void* Minv, long ldm, const void* B,
             long ldb, void* X, long ldx, void* work, void* stream);
int pg_trsm_lower(pg_handle h, int dtype, int n, int nrhs, const void* L, long ldl, const void* inv_diag, const void* Minv, long ldm,
                  const void* B, long ldb, void* X, long ldx, void* work, void* stream);

/* Minv = L^-1 (lower; its strictly upper blocks are scratch).  First half of cholesky_solve against a
 * matrix right-hand side (gpr.py:100,112; loss.py:116). */
int pg_trtri(pg_handle h, int dtype, int n, const void* L, long ldl, const void* inv_diag, void* Minv, long ldm,
             void* stream);

/* Kinv(lower triangle) = Minv^T Minv = K^-1: what loss.py:116 obtains column by column with
 * cholesky_solve(dkrn, L). */
int pg_lauum(pg_handle h, int dtype, int n, const void* Minv, long ldm, void* Kinv, long ldk, void* stream);

/* K^-1 (lower triangle) from the factor in one call = pg_trtri followed by pg_lauum (LAPACK's potri; SURVEY 8b lists it
 * under that name).  work: n x n elements (receives L^-1); Kinv may alias L. */
int pg_potri(pg_handle h, int dtype, int n, const void* L, long ldl, const void* inv_diag, void* Kinv, long ldk,
             void* work, void* stream);

/* out[0] = log det K = 2 sum_i log L_ii over the first n (real) rows: the second term of loss.py:47-49 on its own */
int pg_logdet(pg_handle h, int dtype, int n, const void* L, long ldl, double* out, void* stream);

/* y = op(Minv) x for the lower-triangular Minv (trans: 0 = Minv x, 1 = Minv^T x); work: n/256*n elems */
int pg_trmv(pg_handle h, int dtype, int n, const void* Minv, long ldm, int trans, const void* x, void* y,
            void* work, void* stream);

/* alpha_e = Minv_e^T (Minv_e y_e) for nexp batched experts in three launches (the cholesky_solve of gpr.py:70-72 through the explicit
 * inverse factors of pg_build_potrf_trtri_batched); u: n and work: (n/256) n scratch elements per expert, each with its stride. */
int pg_alpha_batched(pg_handle h, int dtype, int n, const void* Minv, long ldm, long m_stride, const void* y, long y_stride, void* u,
                     long u_stride, void* alpha, long alpha_stride, void* work, long work_stride, int nexp, void* stream);

/* out[0] = 1/2 y^T alpha + sum_i log L_ii + n/2 log 2pi   (loss.py:47-49, 107-109); n = real points */
int pg_nlml_value(pg_handle h, int dtype, int n, const void* L, long ldl, const void* y, const void* alpha,
                  double* out, void* stream);

/* The same alpha and NLML as pg_trmv x 2 + pg_nlml_value (alpha = Minv^T (Minv y), loss.py:102-109), arranged to overlap with the
 * caller's next launch: the log-determinant (from the diagonal of Minv: K^-1 may overwrite L next), the two HBM-bound triangular
 * mat-vecs and the value run on the handle's side stream beside the MFMA-bound pg_lauum the caller enqueues next.
 * out[0] receives the NLML (out[1] is scratch: log det K); u, alpha: n elements; work: (n/256) n elements.  Any later entry point
 * on `stream` other than pg_lauum waits for this work before it starts, so the caller needs no extra synchronisation. */
int pg_alpha_nlml_async(pg_handle h, int dtype, int n_real, int n, const void* L, long ldl, const void* Minv, long ldm, const void* y,
                        void* u, void* alpha, void* work, double* out, void* stream);

/* grad[k] = 1/2 sum (Kinv - alpha alpha^T) o dK/dtheta_k  == -1/2 (tr1 - tr2) of loss.py:116-121 */
long pg_nlml_grad_worksize(int n, int nhp); /* doubles */
int pg_nlml_grad(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* X, long ldx, int n,
                 int d, const void* Kinv, long ldk, const void* alpha, double* grad, int nhp, double* work,
                 long lwork, void* stream);

/* The gradient path for nexp batched experts of one size (round 4): what loss.py:92-128 does in ONE batched factor / solve when the
 * model carries a leading expert dimension (x [nc, n, d] built at gr_bcm.py:19-29, params [nc, nhp]).  After
 * pg_build_potrf_trtri_batched:
 *   pg_alpha_nlml_batched : alpha_e = Minv_e^T (Minv_e y_e) and out[e * out_stride] = NLML_e (log det from the diagonal of Minv_e),
 *                           four launches for the whole batch (u: n, work: (n/256) n scratch elements per expert)
 *   pg_lauum_batched      : Kinv_e = Minv_e^T Minv_e (lower tiles), one launch
 *   pg_nlml_grad_batched  : grad[e * grad_stride + k] as pg_nlml_grad, two launches; work: nexp * pg_nlml_grad_worksize(n, nhp) doubles
 * Strides are in elements; x_stride / hp_stride / y_stride = 0 shares points / hyper-parameters / targets between the experts. */
int pg_alpha_nlml_batched(pg_handle h, int dtype, int n_real, int n, const void* Minv, long ldm, long m_stride, const void* y, long y_stride,
                          void* u, long u_stride, void* alpha, long alpha_stride, void* work, long work_stride, double* out, long out_stride,
                          int nexp, void* stream);
int pg_lauum_batched(pg_handle h, int dtype, int n, const void* Minv, long ldm, long m_stride, void* Kinv, long ldk, long k_stride, int nexp,
                     void* stream);
int pg_nlml_grad_batched(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, long hp_stride, const void* X, long ldx, long x_stride,
                         int n, int d, const void* Kinv, long ldk, long k_stride, const void* alpha, long alpha_stride, double* grad,
                         long grad_stride, int nhp, double* work, long lwork, int nexp, void* stream);

/* Predictive mean and variance from Ks[n_pad x m_pad] = k(X, Xp) (train rows, test columns):
 *   mean[j] = sum_i Ks[i][j] alpha[i]                 (gpr.py:80-85)
 *   var[j]  = kss - sum_i (Minv Ks)[i][j]^2           (gpr.py:98-104: diag(K**) - rowsum(K* o (K^-1 K*^T)^T))
 * kss = the constant diagonal of K** (sum sigma^2 + sum sigma_n^2); var == NULL skips the variance.
 * work: (n_pad/64) * m_pad elements. */
int pg_predict_mean_q(pg_handle h, int dtype, int n_pad, int m_pad, const void* Ks, long ldks, const void* Minv,
                      long ldm, const void* alpha, void* mean, void* var, double kss, void* work, void* stream);

/* The same from the cross-covariance stored test-point-major, Kt[m_pad x n_pad] = k(Xp, X) (build it with pg_kernel_build(Xp, X)):
 * the variance product then reads both operands K-contiguously (the faster NT form of the GEMM core) and the mean is a row-wise
 * mat-vec.  Same results to rounding; work: (n_pad/64) * m_pad elements. */
int pg_predict_mean_q_kt(pg_handle h, int dtype, int n_pad, int m_pad, const void* Kt, long ldkt, const void* Minv,
                         long ldm, const void* alpha, void* mean, void* var, double kss, void* work, void* stream);

/* The diagonal prediction of ALL experts of a batched model in three launches (round 5): what the reference does with one batched
 * kernel / bmm / cholesky_solve on x [nc, n, d] (gpr.py:76-106; the committee's default, gr_bcm.py:151-155).  Expert e reads
 * Kt + e * kt_stride, Minv + e * m_stride, alpha + e * alpha_stride and writes mean + e * mean_stride, var + e * var_stride (var == NULL:
 * means only); kss_e = sum sigma_c^2 + sum sigma_n^2 is formed on the device from hp + e * hp_stride and spec.  work: (n_pad/64) * m_pad
 * elements per expert at work_stride.  Per expert the numbers are those of pg_predict_mean_q_kt, bit for bit. */
int pg_predict_mean_q_kt_batched(pg_handle h, int dtype, int n_pad, int m_pad, const void* Kt, long ldkt, long kt_stride, const void* Minv,
                                 long ldm, long m_stride, const void* alpha, long alpha_stride, void* mean, long mean_stride, void* var,
                                 long var_stride, const pg_covspec* spec, const double* hp, long hp_stride, void* work, long work_stride,
                                 int nexp, void* stream);

/* Dense product V = Minv Ks written out (needed for the full predictive covariance, gpr.py:108-120),
 * and C = C - V^T V on m_pad x m_pad (lower_only != 0: tiles on/below the diagonal only). */
int pg_trmm_lower(pg_handle h, int dtype, int n_pad, int m_pad, const void* Minv, long ldm, const void* Ks,
                  long ldks, void* V, long ldv, void* stream);
int pg_syrk_tn_sub(pg_handle h, int dtype, int m_pad, int n_pad, const void* V, long ldv, void* C, long ldc,
                   int lower_only, void* stream);
/* The same two steps from the cross-covariance stored test-point-major, Kt[m_pad x n_pad] = k(Xp, X) (round 4; what Exact_GP.predict and
 * GRBCM.predict use for var="full", gpr.py:108-120, gr_bcm.py:151-155), for the nexp experts of a batched model in ONE launch each
 * (expert e at base + e * stride elements; a stride of 0 shares an operand): Vt_e[m_pad x n_pad] = Kt_e Minv_e^T = (Minv_e Ks_e)^T, both
 * operands read along k, and C_e = C_e - Vt_e Vt_e^T on m_pad x m_pad.  The tiles of one expert's triangular product pack badly onto the
 * chip (very different lengths) and those of one small m x m output do not fill it; a committee's experts together do both.  Same
 * results as pg_trmm_lower / pg_syrk_tn_sub to rounding. */
int pg_trmm_lower_kt_batched(pg_handle h, int dtype, int n_pad, int m_pad, const void* Minv, long ldm, long m_stride, const void* Kt,
                             long ldkt, long kt_stride, void* Vt, long ldvt, long vt_stride, int nexp, void* stream);
int pg_syrk_nt_sub_batched(pg_handle h, int dtype, int m_pad, int n_pad, const void* Vt, long ldvt, long vt_stride, void* C, long ldc,
                           long c_stride, int nexp, int lower_only, void* stream);

/* Per-expert grBCM terms (gr_bcm.py:125-144): out[0..2][j] = beta, beta*prec, beta*prec*mean with
 * beta = 1/2 (log prec_c - log prec_g), or 1 when is_first (gr_bcm.py:132); accumulate != 0 adds.
 * beta_out / prec_out (nullable, length m) receive this expert's row of GRBCM.beta / GRBCM.prec
 * (gr_bcm.py:135-136). */
int pg_grbcm_local_terms(pg_handle h, int dtype, int m, const void* mean_c, const void* var_c, const void* var_g,
                         int is_first, int accumulate, double* out, long ldo, double* beta_out, double* prec_out,
                         void* stream);
/* The same for the nexp experts a rank owns in ONE launch (round 5): expert c reads mean_l + c * mean_stride, var_l + c * var_stride and
 * fills row c of beta_out / prec_out (leading dimension ldb); `first` = index of the committee's first expert among them (beta = 1), -1 if
 * it lives on another rank.  The sums receive the experts' terms in order: bit-identical to nexp calls of pg_grbcm_local_terms. */
int pg_grbcm_local_terms_batched(pg_handle h, int dtype, int m, const void* mean_l, long mean_stride, const void* var_l, long var_stride,
                                 const void* var_g, int nexp, int first, int accumulate, double* out, long ldo, double* beta_out,
                                 double* prec_out, long ldb, void* stream);
/* Finish the committee (gr_bcm.py:133,143,144) from the summed terms and the global expert; beta0 / prec0
 * (nullable) receive the global expert's row of GRBCM.beta / GRBCM.prec. */
int pg_grbcm_finish(pg_handle h, int dtype, int m, const double* sums, long lds, const void* mean_g,
                    const void* var_g, void* mean, void* var, double* beta0, double* prec0, void* stream);

/* Full-covariance committee (gr_bcm.py:99-114,147).  pg_grbcm_weighted_prec: acc[i][j] (+)= 1/2 (beta_i + beta_j) P[i][j]
 * on the lower triangle for one expert's precision P = cov_c^-1 (rows >= m of a fresh acc get a unit diagonal);
 * pg_symmetrize mirrors the lower triangle up; pg_grbcm_finish_full: mean = diag(cov) o (sum beta prec mu). */
int pg_grbcm_weighted_prec(pg_handle h, int dtype, int m, int m_pad, const void* P, long ldp, const double* beta, void* acc,
                           long lda, int accumulate, void* stream);
int pg_symmetrize(pg_handle h, int dtype, int n, void* A, long lda, void* stream);
int pg_grbcm_finish_full(pg_handle h, int dtype, int m, const double* sums, long lds, const void* mean_g, const void* var_g,
                         const void* cov, long ldc, void* mean, void* stream);

/* Expert partitioning (sampler.py:68-119).  D[i][j] = |x_i - c_j|^2 (euclidean_dist, sampler.py:94-100; D may be NULL)
 * and idx[i] = argmin_j, first minimum (the assignment of cluster_samples, sampler.py:80-82,112-116; idx may be NULL). */
int pg_sqdist_argmin(pg_handle h, int dtype, const void* X, long ldx, int n, const void* C, long ldc, int m, int d, void* D,
                     long ldd, int* idx, void* stream);

/* zero the strictly upper triangle (export of krnchd with torch.cholesky's layout) */
int pg_tril(pg_handle h, int dtype, int n, void* A, long lda, void* stream);

/* pg_potrf overlaps its panel chain (the handle's panel and rows streams) with the trailing update (the handle's CU-masked
 * update stream); all work is joined back onto the caller's stream before pg_potrf returns.  on = 0 keeps everything on the
 * caller's stream (default 1); the two schedules agree to rounding. */
int pg_set_lookahead(pg_handle h, int on);

/* width of pg_potrf's outer column panel: 0 (default) = chosen from n (512 / 1024 / 2048 columns), otherwise a
 * multiple of 128.  The result does not depend on it beyond rounding; a tuning and test knob. */
int pg_set_outer_panel(pg_handle h, int columns);

/* From min_n points on (default 16384; PG_REC_MIN in the environment at pg_create; 0 = never) the fused pg_potrf_trtri /
 * pg_build_potrf_trtri splits the matrix at n / 2 and takes the blocks that cross the split as four large products against the
 * leading half's inverse (csrc/linalg.hip: potrf_trtri_rec; the reference's torch.cholesky + cholesky_solve, gpr.py:69,
 * loss.py:97-116, has no such knob).  min_n must be 0 or a multiple of 512 >= 512.  The result does not depend on it beyond
 * rounding; a tuning and test knob (tests run the split at n = 1024). */
int pg_set_recursive_split(pg_handle h, int min_n);

/* GEMM-core profiling for bench.py's roofline leg: events around every MFMA GEMM launch */
int pg_profile(pg_handle h, int on);   /* on=1 resets and starts, on=0 stops */
int pg_profile_read(pg_handle h, double* flops, double* ms, long* launches);

/* The flag-coupled chain (csrc/chainstep.hip) needs kernels of two of the handle's streams to run at the same time.  pg_create
 * probes that once (2 ms at most) and switches the chain off for the handle where kernels run one at a time -- a counter-collecting
 * profiler (rocprofv3 --pmc), serialising debug settings -- so that the factorisation falls back to its classic chain instead of
 * reporting info = -1.  pg_set_coupled_chain(h, 0) switches the chain off AND destroys the rows stream (the handle then owns two
 * streams);
 * pg_set_coupled_chain(h, 1) re-creates the stream and probes again; pg_coupled_chain reads the state. */
int pg_set_coupled_chain(pg_handle h, int on);
int pg_coupled_chain(pg_handle h);
/* Every wait of the coupled chain is bounded by wall time.  microseconds = 0 (default; PG_CS_SPIN_US in the environment at
 * pg_create overrides): the budget is scaled to the call -- 20x the factorisation's classic-chain estimate, at least 50 ms (n = 4096:
 * 53 ms, 8192: 0.18 s, 16384: 1.0 s); > 0: that many microseconds per wait; < 0: every wait expires at once -- the deterministic test
 * hook of the fall-back path.  pg_chain_timeouts: how many expiries this handle has seen.  Each one switches the handle to the classic
 * chain, TEMPORARILY: after pg_set_rearm_after(h, calls) further factorisations (default 8; PG_CS_REARM; 0 = never) the handle probes
 * its queues again and takes the coupled chain back by itself (pg_chain_rearms counts; a time-out that follows a re-arm doubles that
 * distance, up to 4096, so that a GPU shared for good with another process costs one wait budget ever more rarely); pg_set_coupled_chain(h, 1) does so at once,
 * pg_set_coupled_chain(h, -1) switches to the classic chain as a time-out would (stream kept, automatic re-arm applies),
 * pg_set_coupled_chain(h, 0) for good (rows stream released). */
int pg_set_spin_budget(pg_handle h, long microseconds);
int pg_chain_timeouts(pg_handle h);
int pg_set_rearm_after(pg_handle h, int calls);
int pg_chain_rearms(pg_handle h);
/* the budget of one wait for an n x n factorisation on this handle, microseconds (-1: forced expiry); and one bounded wait on a flag
 * nobody sets with exactly that budget -- scratch: 32 bytes of device memory; afterwards ((long long*)scratch)[2] holds the 10 ns ticks
 * waited and [3] whether the flag came (0: expired).  Diagnostics / tests; the handle's state is not changed. */
long pg_wait_budget_us(pg_handle h, int n);
int pg_spin_probe(pg_handle h, int n, void* scratch, void* stream);

/* how many outer panels of the handle's LAST pg_potrf / pg_potrf_trtri ran on the flag-coupled chain (0: classic chain only;
 * the coupled chain needs the look-ahead schedule, i.e. at least three outer panels; any caller stream works, the legacy
 * default stream included) -- tests / diagnostics */
int pg_last_coupled_panels(pg_handle h);
/* Experimental schedule of the coupled factorisation (part of tc.cholesky, gpr.py:69 / loss.py:97; OFF by default, PG_DEFER=1 in the
 * environment switches it on for new handles): from n = 6144 the columns right of about n / 2 take the leading half's updates as deferred
 * K = n/2-deep products beside the trailing half's chain instead of panel by panel.  Same factor to rounding; measured 3 % slower at
 * n = 8192.  pg_last_deferred_panels: how many column panels of the handle's LAST factorisation were deferred. */
int pg_set_deferred_block(pg_handle h, int on);
int pg_last_deferred_panels(pg_handle h);

/* one 128x128 Cholesky leaf (factor + inverse) on its own; ablate != 0 skips phases -- timing diagnostics only */
int pg_leaf_raw(pg_handle h, int dtype, void* A, long lda, void* inv, long ldi, int* info, int ablate, void* stream);

/* one rows kernel of the flag-coupled chain (chainstep.hip) on its own, every flag it would wait for preset: rows below tile
 * (k0, k0) of an n x n matrix, window from column o0, inv = a 128 x 128 lower-triangular block; flags: 8 ints of scratch --
 * timing diagnostics only (tools/probe_rowstep.py) */
int pg_rowstep_raw(pg_handle h, int dtype, int n, void* A, long lda, int o0, int k0, const void* inv, int* flags, int* info, void* stream);

/* raw MFMA GEMM core, exposed for tests and the roofline micro-benchmark:
 * variant 0: C = a A B^T + b C (128x128 tiles; tri != 0 -> lower tiles only), 2: C = a A B + b C,
 * 3: C = a A^T B + b C.  klo/khi as in csrc/gemm.h. */
int pg_gemm_raw(pg_handle h, int dtype, int variant, int M, int N, int K, double alpha, const void* A, long lda,
                const void* B, long ldb, double beta, void* C, long ldc, int tri, int klo, int khi, void* stream);

#ifdef __cplusplus
}
#endif
#endif
