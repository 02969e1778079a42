// Host drivers of the blocked algorithms plus the small bandwidth-bound kernels around the MFMA core.
//
// potrf  : two-level right-looking (outer 1024 / inner 256).  Each inner step: two 128x128 LDS leaves
//          (leaf.hip) give L_kk and inv(L_kk); the panel solve is a GEMM against that inverse
//          (B <- B inv(L_kk)^T, in place, one workgroup per 64 rows); the rest of the outer panel is
//          updated with K = 256; the trailing matrix gets one lower-tile SYRK with K = 1024 per outer panel.
// trtri  : Minv = L^-1 by recursive doubling over the diagonal: X21 = -X22 (L21 X11); all pairs of one
//          level run in one batched launch; the product L21 X11 is parked (transposed) in the mirrored
//          upper block, so no workspace is needed.
// lauum  : K^-1 = Minv^T Minv as ONE lower-tile launch with per-tile K ranges (out of place).
#include "gemm.h"
#include "leaf.h"
#include "linalg.h"
#include <cmath>

#define NB 256

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void copy_blocks_kernel(const T* __restrict__ src, T* __restrict__ dst, long ldd) {
    // dst diagonal block b (256x256) <- src[b][256][256]
    const int b = blockIdx.x;
    const T* s = src + (long)b * NB * NB;
    T* d = dst + (long)b * NB * ldd + (long)b * NB;
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < NB * NB; idx += gridDim.y * 256)
        d[(long)(idx >> 8) * ldd + (idx & 255)] = s[idx];
}

template <typename T> __global__ __launch_bounds__(256) void tril_kernel(T* __restrict__ A, long lda, int n) {
    const int tc = blockIdx.x, tr = blockIdx.y;
    if (tc < tr) return;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int i = tr * 64 + (idx >> 6), j = tc * 64 + (idx & 63);
        if (i < n && j < n && j > i) A[(long)i * lda + j] = (T)0;
    }
}

// x_b <- op(D_b) x_b for one 256x256 diagonal-block inverse (in place through LDS)
template <typename T>
__global__ __launch_bounds__(256) void blk_matvec_kernel(const T* __restrict__ D, T* __restrict__ x, int trans) {
    __shared__ double xs[NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    xs[tid] = (double)x[tid];
    __syncthreads();
    if (trans) {   // out[j] = sum_i D[i][j] x[i]   (thread per column, coalesced rows)
        double s = 0.0;
        for (int i = tid; i < NB; ++i) s += (double)D[(long)i * NB + tid] * xs[i];   // D lower: i >= j
        x[tid] = (T)s;
    } else {       // out[i] = sum_j D[i][j] x[j]   (wave per row)
        for (int r = 0; r < 64; ++r) {
            const int i = wave * 64 + r;
            double s = 0.0;
            for (int j = lane; j <= i; j += 64) s += (double)D[(long)i * NB + j] * xs[j];
            s = wave_sum(s);
            if (lane == 0) x[i] = (T)s;
        }
    }
}

// forward: y[i] -= sum_k L[i][c0 + k] z[k] for rows i >= r0 (wave per row, 64 rows per workgroup)
template <typename T>
__global__ __launch_bounds__(256) void trsv_fwd_update_kernel(const T* __restrict__ L, long ldl, int c0, int r0, int n,
                                                              T* __restrict__ y) {
    __shared__ double zs[NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    zs[tid] = (double)y[c0 + tid];
    __syncthreads();
    for (int r = 0; r < 16; ++r) {
        const int i = r0 + blockIdx.x * 64 + wave * 16 + r;
        if (i >= n) break;
        const T* row = L + (long)i * ldl + c0;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += (double)row[lane + 64 * q] * zs[lane + 64 * q];
        s = wave_sum(s);
        if (lane == 0) y[i] = (T)((double)y[i] - s);
    }
}

// backward: z[j] -= sum_i L[r0 + i][j] a[i] for columns j < r0 (thread per column)
template <typename T>
__global__ __launch_bounds__(256) void trsv_bwd_update_kernel(const T* __restrict__ L, long ldl, int r0, T* __restrict__ z) {
    __shared__ double as[NB];
    const int tid = threadIdx.x;
    as[tid] = (double)z[r0 + tid];
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    double s = 0.0;
    for (int i = 0; i < NB; ++i) s += (double)L[(long)(r0 + i) * ldl + j] * as[i];
    z[j] = (T)((double)z[j] - s);
}

// part[rc][j] = sum_{i in row chunk rc} A[i][j] x[i]; tri: skip chunks above the diagonal
template <typename T>
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(const T* __restrict__ A, long lda, const T* __restrict__ x,
                                                             T* __restrict__ part, long ldp, int tri) {
    const int cc = blockIdx.x, rc = blockIdx.y;
    if (tri && rc < cc) return;
    __shared__ double xs[256];
    const int tid = threadIdx.x;
    xs[tid] = (double)x[rc * 256 + tid];
    __syncthreads();
    const int j = cc * 256 + tid;
    const T* a = A + (long)rc * 256 * lda + j;
    double s = 0.0;
#pragma unroll 8
    for (int i = 0; i < 256; ++i) s += (double)a[(long)i * lda] * xs[i];
    part[(long)rc * ldp + j] = (T)s;
}

// out[j] = a + b * sum_{rc >= rc0(j)} part[rc][j]
template <typename T>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ part, long ldp, int nrc, int cols,
                                                        T* __restrict__ out, int tri, double a, double b) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    double s = 0.0;
    for (int rc = tri ? j / 256 : 0; rc < nrc; ++rc) s += (double)part[(long)rc * ldp + j];
    out[j] = (T)(a + b * s);
}

// y[i] = sum_{j < jend(i)} M[i][j] x[j], wave per row; jend = end of row i's 256-block (lower-triangular M)
template <typename T>
__global__ __launch_bounds__(256) void trmv_n_kernel(const T* __restrict__ M, long ldm, int n, const T* __restrict__ x,
                                                     T* __restrict__ y) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = 0; r < 4; ++r) {
        const int i = blockIdx.x * 16 + wave * 4 + r;
        if (i >= n) return;
        const int jend = (i / NB + 1) * NB;
        const T* row = M + (long)i * ldm;
        double s = 0.0;
        for (int j = lane; j < jend; j += 64) s += (double)row[j] * (double)x[j];
        s = wave_sum(s);
        if (lane == 0) y[i] = (T)s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nlml_value_kernel(const T* __restrict__ L, long ldl, const T* __restrict__ y,
                                                         const T* __restrict__ alpha, int n, double* __restrict__ out) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < n; i += 256)
        s += 0.5 * (double)y[i] * (double)alpha[i] + log((double)L[(long)i * ldl + i]);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[0] = red[0] + red[1] + red[2] + red[3] + 0.5 * (double)n * 1.83787706640934548356;  // log 2pi
}

template <typename T>
__global__ __launch_bounds__(256) void grbcm_terms_kernel(const T* __restrict__ mean_c, const T* __restrict__ var_c,
                                                          const T* __restrict__ var_g, int m, int is_first,
                                                          int accumulate, double* __restrict__ out, long ldo,
                                                          double* __restrict__ beta_out, double* __restrict__ prec_out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double pc = 1.0 / (double)var_c[j], pg = 1.0 / (double)var_g[j];
    const double beta = is_first ? 1.0 : 0.5 * (log(pc) - log(pg));
    if (beta_out) beta_out[j] = beta;
    if (prec_out) prec_out[j] = pc;
    const double t0 = beta, t1 = beta * pc, t2 = beta * pc * (double)mean_c[j];
    if (accumulate) { out[j] += t0; out[ldo + j] += t1; out[2 * ldo + j] += t2; }
    else { out[j] = t0; out[ldo + j] = t1; out[2 * ldo + j] = t2; }
}

template <typename T>
__global__ __launch_bounds__(256) void grbcm_finish_kernel(const double* __restrict__ sums, long lds, const T* __restrict__ mean_g,
                                                           const T* __restrict__ var_g, int m, T* __restrict__ mean,
                                                           T* __restrict__ var, double* __restrict__ beta0,
                                                           double* __restrict__ prec0) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double pg = 1.0 / (double)var_g[j];
    const double b0 = 1.0 - sums[j];
    if (beta0) beta0[j] = b0;
    if (prec0) prec0[j] = pg;
    const double v = 1.0 / (sums[lds + j] + b0 * pg);
    var[j] = (T)v;
    mean[j] = (T)(v * (sums[2 * lds + j] + b0 * pg * (double)mean_g[j]));
}

#define LAUNCH_CHECK() PG_CHECK(hipGetLastError())

// ------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------
template <typename T> static GemmP<T> gp0() {
    GemmP<T> p;
    p.A = p.B = nullptr; p.C = nullptr;
    p.lda = p.ldb = p.ldc = 0;
    p.M = p.N = p.K = 0;
    p.alpha = (T)1; p.beta = (T)0;
    p.tri = p.klo = p.khi = 0;
    p.sA = p.sB = p.sC = 0; p.batch = 1;
    p.part = nullptr; p.ldp = 0; p.info = nullptr;
    return p;
}

long pg_potrf_worksize_impl(int n) { return (long)n * NB + 128 * 128; }

// Factor the 256x256 diagonal block at k0 (two LDS leaves + three single-tile GEMMs) and assemble its inverse.
template <typename T>
static int diag_block(pg_ctx* ctx, hipStream_t st, T* A, long lda, int k0, T* inv, T* scratch, int* info) {
    T* Akk = A + (long)k0 * lda + k0;
    T* A21 = Akk + 128 * lda;
    T* A22 = A21 + 128;
    T* inv22 = inv + 128 * NB + 128;
    int rc;
    if ((rc = pg_leaf<T>(st, Akk, lda, inv, NB, info, k0))) return rc;
    GemmP<T> p = gp0<T>();
    p.info = info;
    // A21 <- A21 inv11^T
    p.M = p.N = p.K = 128; p.A = A21; p.lda = lda; p.B = inv; p.ldb = NB; p.C = A21; p.ldc = lda; p.khi = 2;
    if ((rc = pg_gemm<T>(ctx, st, GEMM_NT_128, p))) return rc;
    // A22 -= A21 A21^T
    p.khi = 0; p.B = A21; p.ldb = lda; p.C = A22; p.alpha = (T)-1; p.beta = (T)1;
    if ((rc = pg_gemm<T>(ctx, st, GEMM_NT_128, p))) return rc;
    if ((rc = pg_leaf<T>(st, A22, lda, inv22, NB, info, k0 + 128))) return rc;
    // inv21 = -inv22 (L21 inv11): scratch = (L21 inv11)^T = inv11^T L21^T, then NT against it
    p = gp0<T>(); p.info = info;
    p.M = p.N = p.K = 128; p.A = inv; p.lda = NB; p.B = A21; p.ldb = lda; p.C = scratch; p.ldc = 128; p.klo = 1;
    if ((rc = pg_gemm<T>(ctx, st, GEMM_TT_128, p))) return rc;
    p.klo = 0; p.khi = 1; p.A = inv22; p.lda = NB; p.B = scratch; p.ldb = 128; p.C = inv + 128 * NB; p.ldc = NB;
    p.alpha = (T)-1;
    return pg_gemm<T>(ctx, st, GEMM_NT_128, p);
}

// Two-level right-looking Cholesky: inner steps of NB = 256 columns (diagonal block, panel solve against its
// inverse, update of the REST OF THE OUTER PANEL only), and one lower-tile SYRK with K = NBO per outer panel, so
// the big trailing update re-reads/re-writes C once per NBO columns instead of once per 256.
#define NBO 1024
template <typename T>
int pg_potrf_t(pg_ctx* ctx, hipStream_t st, int n, T* A, long lda, T* invD, int* info) {
    if (n <= 0 || n % NB) { pg_set_error("pg_potrf: n=%d is not a positive multiple of %d", n, NB); return -2; }
    PG_CHECK(hipMemsetAsync(info, 0, sizeof(int), st));
    PG_CHECK(hipMemsetAsync(invD, 0, (size_t)pg_potrf_worksize_impl(n) * sizeof(T), st));
    T* scratch = invD + (long)n * NB;   // 128 x 128
    int rc;
    for (int o0 = 0; o0 < n; o0 += NBO) {
        const int oend = std::min(n, o0 + NBO);
        for (int k0 = o0; k0 < oend; k0 += NB) {
            T* inv = invD + (long)(k0 / NB) * NB * NB;
            if ((rc = diag_block<T>(ctx, st, A, lda, k0, inv, scratch, info))) return rc;
            const int m = n - k0 - NB;
            if (m <= 0) continue;
            T* P = A + (long)(k0 + NB) * lda + k0;          // rows below the diagonal block, 256 columns
            GemmP<T> p = gp0<T>(); p.info = info;
            p.M = m; p.N = NB; p.K = NB; p.A = P; p.lda = lda; p.B = inv; p.ldb = NB; p.C = P; p.ldc = lda; p.khi = 2;
            if ((rc = pg_gemm<T>(ctx, st, GEMM_NT_RP, p))) return rc;
            const int w = oend - k0 - NB;                   // columns of the outer panel still to be factored
            if (w > 0) {
                p = gp0<T>(); p.info = info;
                p.M = m; p.N = w; p.K = NB; p.A = P; p.lda = lda; p.B = P; p.ldb = lda; p.C = P + NB; p.ldc = lda;
                p.alpha = (T)-1; p.beta = (T)1;
                if ((rc = pg_gemm<T>(ctx, st, GEMM_NT_128, p))) return rc;
            }
        }
        const int m = n - oend;
        if (m > 0) {
            T* P = A + (long)oend * lda + o0;               // m x (oend - o0): final L values of this outer panel
            GemmP<T> p = gp0<T>(); p.info = info;
            p.M = p.N = m; p.K = oend - o0; p.A = P; p.lda = lda; p.B = P; p.ldb = lda;
            p.C = A + (long)oend * lda + oend; p.ldc = lda;
            p.alpha = (T)-1; p.beta = (T)1; p.tri = 1;
            if ((rc = pg_gemm<T>(ctx, st, GEMM_NT_128, p))) return rc;
        }
    }
    return 0;
}

template <typename T>
int pg_potrs_vec_t(pg_ctx*, hipStream_t st, int n, const T* L, long ldl, const T* invD, const T* y, T* x) {
    if (n <= 0 || n % NB) { pg_set_error("pg_potrs_vec: n=%d is not a positive multiple of %d", n, NB); return -2; }
    if (x != y) PG_CHECK(hipMemcpyAsync(x, y, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, st));
    const int nb = n / NB;
    for (int b = 0; b < nb; ++b) {   // L z = y
        hipLaunchKernelGGL(blk_matvec_kernel<T>, dim3(1), dim3(256), 0, st, invD + (long)b * NB * NB, x + b * NB, 0);
        const int r0 = (b + 1) * NB;
        if (r0 < n)
            hipLaunchKernelGGL(trsv_fwd_update_kernel<T>, dim3((n - r0) / 64), dim3(256), 0, st, L, ldl, b * NB, r0, n, x);
    }
    for (int b = nb - 1; b >= 0; --b) {   // L^T a = z
        hipLaunchKernelGGL(blk_matvec_kernel<T>, dim3(1), dim3(256), 0, st, invD + (long)b * NB * NB, x + b * NB, 1);
        if (b > 0)
            hipLaunchKernelGGL(trsv_bwd_update_kernel<T>, dim3(b), dim3(256), 0, st, L, ldl, b * NB, x);
    }
    LAUNCH_CHECK();
    return 0;
}

template <typename T>
int pg_trtri_t(pg_ctx* ctx, hipStream_t st, int n, const T* L, long ldl, const T* invD, T* M, long ldm) {
    if (n <= 0 || n % NB) { pg_set_error("pg_trtri: n=%d is not a positive multiple of %d", n, NB); return -2; }
    hipLaunchKernelGGL(copy_blocks_kernel<T>, dim3(n / NB, 16), dim3(256), 0, st, invD, M, ldm);
    LAUNCH_CHECK();
    // invariant: the diagonal is tiled by `nfull` inverted blocks of size h plus one smaller inverted block `rem`
    int rc;
    long h = NB, rem = 0;
    for (;;) {
        const long nfull = (n - rem) / h;
        if (nfull + (rem ? 1 : 0) <= 1) break;
        const long npair = nfull / 2;
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0: the npair full pairs as one batch; pass 1: the trailing (h, rem) pair if there is one
            long r0, h2, batch;
            if (pass == 0) { if (!npair) continue; r0 = 0; h2 = h; batch = npair; }
            else { if (!((nfull & 1) && rem)) continue; r0 = (nfull - 1) * h; h2 = rem; batch = 1; }
            const long stride = 2 * h * (ldm + 1);
            GemmP<T> p = gp0<T>();
            // S = (L21 X11)^T = X11^T L21^T  -> mirrored block M[r0 : r0+h, r0+h : r0+h+h2]
            p.M = (int)h; p.N = (int)h2; p.K = (int)h;
            p.A = M + r0 * ldm + r0; p.lda = ldm;
            p.B = L + (r0 + h) * ldl + r0; p.ldb = ldl;
            p.C = M + r0 * ldm + r0 + h; p.ldc = ldm;
            p.klo = 1; p.batch = (int)batch;
            p.sA = stride; p.sC = stride; p.sB = 2 * h * (ldl + 1);
            if ((rc = pg_gemm<T>(ctx, st, GEMM_TT_128, p))) return rc;
            // X21 = -X22 S^T
            p = gp0<T>();
            p.M = (int)h2; p.N = (int)h; p.K = (int)h2;
            p.A = M + (r0 + h) * ldm + r0 + h; p.lda = ldm;
            p.B = M + r0 * ldm + r0 + h; p.ldb = ldm;
            p.C = M + (r0 + h) * ldm + r0; p.ldc = ldm;
            p.alpha = (T)-1; p.khi = 1; p.batch = (int)batch;
            p.sA = p.sB = p.sC = stride;
            if ((rc = pg_gemm<T>(ctx, st, GEMM_NT_128, p))) return rc;
        }
        if (nfull & 1) rem = rem ? h + rem : h;   // the odd block merges with rem, or becomes the new rem
        h *= 2;
    }
    return 0;
}

template <typename T>
int pg_lauum_t(pg_ctx* ctx, hipStream_t st, int n, const T* M, long ldm, T* Kinv, long ldk) {
    if (n <= 0 || n % NB) { pg_set_error("pg_lauum: n=%d is not a positive multiple of %d", n, NB); return -2; }
    GemmP<T> p = gp0<T>();
    p.M = p.N = p.K = n; p.A = M; p.lda = ldm; p.B = M; p.ldb = ldm; p.C = Kinv; p.ldc = ldk;
    p.tri = 1; p.klo = 1;
    return pg_gemm<T>(ctx, st, GEMM_TN_128, p);
}

template <typename T>
int pg_trmv_t(pg_ctx*, hipStream_t st, int n, const T* M, long ldm, int trans, const T* x, T* y, T* work) {
    if (n <= 0 || n % NB) { pg_set_error("pg_trmv: n=%d is not a positive multiple of %d", n, NB); return -2; }
    if (!trans) {
        hipLaunchKernelGGL(trmv_n_kernel<T>, dim3((n + 15) / 16), dim3(256), 0, st, M, ldm, n, x, y);
    } else {
        hipLaunchKernelGGL(gemv_t_partial_kernel<T>, dim3(n / 256, n / 256), dim3(256), 0, st, M, ldm, x, work, (long)n, 1);
        hipLaunchKernelGGL(colreduce_kernel<T>, dim3(n / 256), dim3(256), 0, st, work, (long)n, n / 256, n, y, 1, 0.0, 1.0);
    }
    LAUNCH_CHECK();
    return 0;
}

template <typename T>
int pg_nlml_value_t(hipStream_t st, int n, const T* L, long ldl, const T* y, const T* alpha, double* out) {
    hipLaunchKernelGGL(nlml_value_kernel<T>, dim3(1), dim3(256), 0, st, L, ldl, y, alpha, n, out);
    LAUNCH_CHECK();
    return 0;
}

template <typename T>
int pg_predict_mean_q_t(pg_ctx* ctx, hipStream_t st, int n, int m, const T* Ks, long ldks, const T* M, long ldm,
                        const T* alpha, T* mean, T* q, double kss, T* work) {
    if (n % NB || m % 256 || n <= 0 || m <= 0) { pg_set_error("pg_predict_mean_q: n_pad=%d m_pad=%d must be multiples of 256", n, m); return -2; }
    hipLaunchKernelGGL(gemv_t_partial_kernel<T>, dim3(m / 256, n / 256), dim3(256), 0, st, Ks, ldks, alpha, work, (long)m, 0);
    hipLaunchKernelGGL(colreduce_kernel<T>, dim3(m / 256), dim3(256), 0, st, work, (long)m, n / 256, m, mean, 0, 0.0, 1.0);
    LAUNCH_CHECK();
    if (q) {
        GemmP<T> p = gp0<T>();
        p.M = n; p.N = m; p.K = n; p.A = M; p.lda = ldm; p.B = Ks; p.ldb = ldks; p.khi = 1;
        p.part = work; p.ldp = m;
        int rc = pg_gemm<T>(ctx, st, GEMM_NN_128_SS, p);
        if (rc) return rc;
        hipLaunchKernelGGL(colreduce_kernel<T>, dim3(m / 256), dim3(256), 0, st, work, (long)m, n / 64, m, q, 0, kss, -1.0);
        LAUNCH_CHECK();
    }
    return 0;
}

template <typename T>
int pg_trmm_lower_t(pg_ctx* ctx, hipStream_t st, int n, int m, const T* M, long ldm, const T* Ks, long ldks, T* V, long ldv) {
    if (n % NB || m % 128) { pg_set_error("pg_trmm_lower: n_pad=%d m_pad=%d not aligned", n, m); return -2; }
    GemmP<T> p = gp0<T>();
    p.M = n; p.N = m; p.K = n; p.A = M; p.lda = ldm; p.B = Ks; p.ldb = ldks; p.C = V; p.ldc = ldv; p.khi = 1;
    return pg_gemm<T>(ctx, st, GEMM_NN_128, p);
}

template <typename T>
int pg_syrk_tn_sub_t(pg_ctx* ctx, hipStream_t st, int m, int n, const T* V, long ldv, T* C, long ldc, int lower_only) {
    if (n % 16 || m % 128) { pg_set_error("pg_syrk_tn_sub: m_pad=%d n_pad=%d not aligned", m, n); return -2; }
    GemmP<T> p = gp0<T>();
    p.M = p.N = m; p.K = n; p.A = V; p.lda = ldv; p.B = V; p.ldb = ldv; p.C = C; p.ldc = ldc;
    p.alpha = (T)-1; p.beta = (T)1; p.tri = lower_only ? 1 : 0;
    return pg_gemm<T>(ctx, st, GEMM_TN_128, p);
}

template <typename T>
int pg_grbcm_terms_t(hipStream_t st, int m, const T* mean_c, const T* var_c, const T* var_g, int is_first, int accumulate,
                     double* out, long ldo, double* beta_out, double* prec_out) {
    hipLaunchKernelGGL(grbcm_terms_kernel<T>, dim3((m + 255) / 256), dim3(256), 0, st, mean_c, var_c, var_g, m, is_first,
                       accumulate, out, ldo, beta_out, prec_out);
    LAUNCH_CHECK();
    return 0;
}
template <typename T>
int pg_grbcm_finish_t(hipStream_t st, int m, const double* sums, long lds, const T* mean_g, const T* var_g, T* mean, T* var,
                      double* beta0, double* prec0) {
    hipLaunchKernelGGL(grbcm_finish_kernel<T>, dim3((m + 255) / 256), dim3(256), 0, st, sums, lds, mean_g, var_g, m, mean, var,
                       beta0, prec0);
    LAUNCH_CHECK();
    return 0;
}
template <typename T> int pg_tril_t(hipStream_t st, int n, T* A, long lda) {
    const int t = (n + 63) / 64;
    hipLaunchKernelGGL(tril_kernel<T>, dim3(t, t), dim3(256), 0, st, A, lda, n);
    LAUNCH_CHECK();
    return 0;
}

#define INST(T)                                                                                                        \
    template int pg_potrf_t<T>(pg_ctx*, hipStream_t, int, T*, long, T*, int*);                                         \
    template int pg_potrs_vec_t<T>(pg_ctx*, hipStream_t, int, const T*, long, const T*, const T*, T*);                 \
    template int pg_trtri_t<T>(pg_ctx*, hipStream_t, int, const T*, long, const T*, T*, long);                         \
    template int pg_lauum_t<T>(pg_ctx*, hipStream_t, int, const T*, long, T*, long);                                   \
    template int pg_trmv_t<T>(pg_ctx*, hipStream_t, int, const T*, long, int, const T*, T*, T*);                       \
    template int pg_nlml_value_t<T>(hipStream_t, int, const T*, long, const T*, const T*, double*);                    \
    template int pg_predict_mean_q_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, const T*, long, const T*, T*, \
                                        T*, double, T*);                                                               \
    template int pg_trmm_lower_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, const T*, long, T*, long);         \
    template int pg_syrk_tn_sub_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, T*, long, int);                     \
    template int pg_grbcm_terms_t<T>(hipStream_t, int, const T*, const T*, const T*, int, int, double*, long, double*, \
                                     double*);                                                                         \
    template int pg_grbcm_finish_t<T>(hipStream_t, int, const double*, long, const T*, const T*, T*, T*, double*,     \
                                      double*);                                                                        \
    template int pg_tril_t<T>(hipStream_t, int, T*, long);
INST(double)
INST(float)
