#pragma once
#include "common.h"
#include "pygpr_hip.h"
long pg_potrf_worksize_impl(int n);
long long pg_wait_ticks(const pg_ctx* ctx, int n);   // budget of one wait of the coupled chain for an n x n factorisation, 10 ns ticks
// A covariance build folded into the factorisation (pg_build_potrf_trtri): A is written by pg_potrf_t itself, the first outer
// panel's columns before the look-ahead starts and the rest on the update stream while the first panel's chain runs.
template <typename T> struct BuildReq {
    const pg_covspec* spec;
    const double* hp;
    const T* X;
    long ldx;
    int n_real, d;
    double jitter;
};
// Batched experts: nexp independent problems of the same size in every launch of the call (A + e * eA, the workspace + e * eInv,
// Minv + e * eM, info[e]; a folded build reads X + e * eX and hp + e * ehp).  The per-step launches of the latency-bound chain are
// then paid once per batch instead of once per expert; batches that are still latency-bound (experts of >= 2048 points, <= 24576 rows in
// all) take the flag-coupled chain with an expert dimension (round 4), the others the classic chain.
struct ExpBatch {
    int nexp;
    long eA, eInv, eM, eX, ehp;
};
template <typename T>
int pg_potrf_t(pg_ctx*, hipStream_t, int n, T* A, long lda, T* invD, int* info, T* Minv, long ldm, const BuildReq<T>* build = nullptr,
               const ExpBatch* eb = nullptr, int col_off = 0, int keep_info = 0);   // col_off: added to the column a bad pivot reports (a call on a
                                                                                     // trailing block); keep_info: do not reset *info (it continues a call)
template <typename T> int pg_potrs_vec_t(pg_ctx*, hipStream_t, int n, const T* L, long ldl, const T* invD, const T* y, T* x, T* work);
template <typename T>
int pg_trtri_t(pg_ctx*, hipStream_t, int n, const T* L, long ldl, const T* invD, T* M, long ldm, int hmax = 0, const ExpBatch* eb = nullptr);
long pg_potrs_vec_worksize_impl(int n);
template <typename T> int pg_logdet_t(hipStream_t, int n, const T* L, long ldl, double* out);
template <typename T>
int pg_potrs_t(pg_ctx*, hipStream_t, int n, int nrhs, const T* L, long ldl, const T* invD, const T* Minv, long ldm, const T* B, long ldb,
               T* X, long ldx, T* work, int both);
template <typename T> int pg_lauum_t(pg_ctx*, hipStream_t, int n, const T* M, long ldm, T* Kinv, long ldk, const ExpBatch* eb = nullptr);
template <typename T> int pg_trmv_t(pg_ctx*, hipStream_t, int n, const T* M, long ldm, int trans, const T* x, T* y, T* work);
template <typename T>
int pg_alpha_batched_t(hipStream_t, int n, const T* M, long ldm, long eM, const T* y, long ey, T* u, long eu, T* alpha, long ea, T* work, long ew,
                       int nexp, int n_real = 0, double* out = nullptr, long eo = 0);
template <typename T>
int pg_alpha_nlml_async_t(pg_ctx*, hipStream_t, int n_real, int n, const T* L, long ldl, const T* Minv, long ldm, const T* y, T* u,
                          T* alpha, T* work, double* out);
template <typename T> int pg_nlml_value_t(hipStream_t, int n, const T* L, long ldl, const T* y, const T* alpha, double* out);
template <typename T>
int pg_predict_mean_q_t(pg_ctx*, hipStream_t, int n, int m, const T* Ks, long ldks, const T* M, long ldm, const T* alpha,
                        T* mean, T* q, double kss, T* work);
template <typename T>
int pg_predict_mean_q_kt_t(pg_ctx*, hipStream_t, int n, int m, const T* Kt, long ldkt, const T* M, long ldm, const T* alpha,
                        T* mean, T* q, double kss, T* work);
template <typename T>
int pg_predict_mean_q_kt_batched_t(pg_ctx*, hipStream_t, int n, int m, const T* Kt, long ldkt, long ekt, const T* M, long ldm, long em,
                                   const T* alpha, long ea, T* mean, long emean, T* q, long evar, const pg_covspec& spec, const double* hp,
                                   long ehp, T* work, long ew, int nexp);
template <typename T>
int pg_grbcm_terms_batched_t(hipStream_t, int m, const T* mean_l, long emean, const T* var_l, long evar, const T* var_g, int nexp, int first,
                             int accumulate, double* out, long ldo, double* beta_out, double* prec_out, long ldb);
template <typename T> int pg_trmm_lower_t(pg_ctx*, hipStream_t, int n, int m, const T* M, long ldm, const T* Ks, long ldks, T* V, long ldv);
template <typename T> int pg_syrk_tn_sub_t(pg_ctx*, hipStream_t, int m, int n, const T* V, long ldv, T* C, long ldc, int lower_only);
template <typename T> int pg_trmm_lower_kt_t(pg_ctx*, hipStream_t, int n, int m, const T* M, long ldm, long em, const T* Kt, long ldkt, long ekt, T* Vt,
                                             long ldvt, long evt, int nexp);
template <typename T> int pg_syrk_nt_sub_t(pg_ctx*, hipStream_t, int m, int n, const T* Vt, long ldvt, long evt, T* C, long ldc, long ec, int nexp,
                                           int lower_only);
template <typename T>
int pg_grbcm_terms_t(hipStream_t, int m, const T* mean_c, const T* var_c, const T* var_g, int is_first, int accumulate,
                     double* out, long ldo, double* beta_out, double* prec_out);
template <typename T>
int pg_grbcm_finish_t(hipStream_t, int m, const double* sums, long lds, const T* mean_g, const T* var_g, T* mean, T* var,
                      double* beta0, double* prec0);
template <typename T> int pg_tril_t(hipStream_t, int n, T* A, long lda);
template <typename T>
int pg_weighted_prec_t(hipStream_t, int m, int m_pad, const T* P, long ldp, const double* beta, T* acc, long lda, int accumulate);
template <typename T> int pg_symmetrize_t(hipStream_t, int n, T* A, long lda);
template <typename T>
int pg_grbcm_finish_full_t(hipStream_t, int m, const double* sums, long lds, const T* mean_g, const T* var_g, const T* cov, long ldc,
                           T* mean);
