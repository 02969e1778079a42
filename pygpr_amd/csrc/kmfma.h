// Matrix-pipe bodies of the covariance build and the gradient contraction (kmfma.hip); dispatched from pg_kbuild / pg_nlml_grad_t.
#pragma once
#include "kbuild.h"
template <typename T>
int pg_kbuild_mfma(hipStream_t st, const pg_covspec& spec, const double* hp, const T* Xr, long ldr, int nr, const T* Xc, long ldc, int nc, int d,
                   int symmetric, int mirror, double jitter, T* K, long ldk, int c0, int c1, int S, long strips, int nexp, long eX, long ehp,
                   long eK, long eXr);
template <typename T>
int pg_grad_mfma(hipStream_t st, const pg_covspec& spec, const double* hp, const T* X, long ldx, int n, int d, const T* Kinv, long ldk,
                 const T* alpha, double* part, int nhp, int tiles, const GradBatch& gb, int nexp, int* nblk);
