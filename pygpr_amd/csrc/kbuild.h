#pragma once
#include "common.h"
#include "pygpr_hip.h"

struct GradBatch {          // strides between batched experts of the gradient contraction (elements); all zero for one expert
    long eX, ehp, eK, ea, epart, egrad;
};
template <typename T>
int pg_kbuild(hipStream_t st, const pg_covspec& spec, const double* hp, const T* Xr, long ldr, int nr,
              const T* Xc, long ldc, int nc, int d, int symmetric, int lower_only, int accumulate, double jitter, T* K,
              long ldk, int rows_pad, int cols_pad, int col0 = 0, int col1 = 0,    // [col0, col1): column window (0, 0 = all)
              int nexp = 1, long eX = 0, long ehp = 0, long eK = 0,                // batched experts: strides of X (= Xc), hp and K
              long eXr = -1);                                                      // ... and of the row points of a cross build (-1: eX)
template <typename T>
int pg_nlml_grad_t(hipStream_t st, const pg_covspec& spec, const double* hp, const T* X, long ldx, int n, int d,
                   const T* Kinv, long ldk, const T* alpha, double* grad, int nhp, double* work, long lwork,
                   int nexp = 1, long ehp = 0, long eX = 0, long eK = 0, long ea = 0, long egrad = 0);   // batched experts: strides
long pg_nlml_grad_worksize_impl(int n, int nhp);
template <typename T>
int pg_centres(hipStream_t st, const T* X, long ldx, int n, const T* Cn, long ldc, int m, int d, T* D, long ldd, int* idx);
template <typename T>
int pg_kgrad(hipStream_t st, const pg_covspec& spec, const double* hp, const T* X, long ldx, int n, int d, T* dK);
