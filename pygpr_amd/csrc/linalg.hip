// Host drivers of the blocked algorithms plus the small bandwidth-bound kernels around the MFMA core.
//
// potrf  : two-level.  Outer panels of NBO = 1024 columns; inside a panel, left-looking steps of 128 columns:
//            U: this 128-column block -= (earlier columns of the panel) x (their rows of the block)^T   (skinny GEMM)
//            leaf: 128x128 LDS factorisation + inverse of the diagonal block (leaf.hip)
//            T: rows below <- rows below x inv(L_kk)^T, in place (each workgroup owns 128 full rows)
//          then ONE lower-tile SYRK with K = 1024 on the trailing matrix, so C is re-read once per 1024 columns.
// trtri  : Minv = L^-1 by recursive doubling over the diagonal: X21 = -X22 (L21 X11); all pairs of one
//          level run in one batched launch; the product L21 X11 is parked (transposed) in the mirrored
//          upper block, so no workspace is needed.
// lauum  : K^-1 = Minv^T Minv as ONE lower-tile launch with per-tile K ranges (out of place).
// potrs  : blocked substitution with the 128x128 diagonal inverses, one fused kernel per block step.
#include "gemm.h"
#include "leaf.h"
#include "chainstep.h"
#include "kbuild.h"
#include "linalg.h"
#include <cmath>
#include <cstdlib>
#include <vector>

#define NB 128      // diagonal block / leaf size; inv_diag holds [n/128][128][128]
// Outer panel of the Cholesky.  Narrow panels keep the serial chain short (its U products grow with the panel), wide
// ones make the trailing update deep enough to run at the GEMM core's rate; measured on MI355X (fp64, fused with L^-1):
// 512 wins up to n = 9216 (-4 % at 8192), 1024 from 12288 to 16384, 2048 from 20480 (-5 % at 32768).
// pg_set_outer_panel (initialised from PG_NBO) overrides.
static int pg_nbo(const pg_ctx* ctx, int n) {
    if (ctx->nbo) return ctx->nbo;
    return n <= 10240 ? 512 : (n <= 18432 ? 1024 : 2048);
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void copy_blocks_kernel(const T* __restrict__ src, T* __restrict__ dst, long ldd, long esrc, long edst) {
    // dst diagonal block b (128x128) <- src[b][128][128]   (blockIdx.z: batched expert)
    src += blockIdx.z * esrc;
    dst += blockIdx.z * edst;
    const int b = blockIdx.x;
    const T* s = src + (long)b * NB * NB;
    T* d = dst + (long)b * NB * ldd + (long)b * NB;
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < NB * NB; idx += gridDim.y * 256)
        d[(long)(idx >> 7) * ldd + (idx & 127)] = s[idx];
}

// dst[r][0:cols] <- src[r][0:cols] for r < rows (16-byte vectors; cols a multiple of 128): the panel's solved rows going home
template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, long lds_, T* __restrict__ dst, long ldd,
                                                        int rows, int cols) {
    constexpr int VE = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    const int vpr = cols / VE;                       // vectors per row
    const long total = (long)rows * vpr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / vpr;
        const int c = (int)(idx % vpr) * VE;
        *reinterpret_cast<vec_t*>(dst + r * ldd + c) = *reinterpret_cast<const vec_t*>(src + r * lds_ + c);
    }
}

template <typename T> __global__ __launch_bounds__(256) void tril_kernel(T* __restrict__ A, long lda, int n) {
    const int tc = blockIdx.x, tr = blockIdx.y;
    if (tc < tr) return;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int i = tr * 64 + (idx >> 6), j = tc * 64 + (idx & 63);
        if (i < n && j < n && j > i) A[(long)i * lda + j] = (T)0;
    }
}

// Forward step b of L z = y:  z_b = inv(L_bb) w_b (every workgroup recomputes it: 128x128 matvec), workgroup 0
// stores z_b, and workgroup g updates its 64 rows below:  w[i] -= sum_k L[i][b*128 + k] z_b[k].
// All global loads of a phase are issued before the first reduction (the loops are latency-bound otherwise).
template <typename T>
__global__ __launch_bounds__(256) void trsv_fwd_step_kernel(const T* __restrict__ L, long ldl, const T* __restrict__ invb,
                                                            int b, int n, T* __restrict__ w, T* __restrict__ z) {
    __shared__ double ws[NB], zs[NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = b * NB;
    if (tid < NB) ws[tid] = (double)w[c0 + tid];
    // update rows of this workgroup: prefetch their L entries while the diagonal solve runs
    const int r0 = c0 + NB + blockIdx.x * 64 + wave * 16;
    T l0[16], l1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = r0 + r;
        const T* row = L + (long)min(i, n - 1) * ldl + c0;
        l0[r] = row[lane];
        l1[r] = row[lane + 64];
    }
    // diagonal inverse (lower triangular): wave handles 32 rows, lane handles columns lane, lane + 64
    T d0[32], d1[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const T* row = invb + (long)(wave * 32 + r) * NB;
        d0[r] = row[lane];
        d1[r] = row[lane + 64];
    }
    __syncthreads();
    const double w0 = ws[lane], w1 = ws[lane + 64];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int i = wave * 32 + r;
        double s = (lane <= i ? (double)d0[r] * w0 : 0.0) + (lane + 64 <= i ? (double)d1[r] * w1 : 0.0);
        s = wave_sum(s);
        if (lane == 0) zs[i] = s;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < NB) z[c0 + tid] = (T)zs[tid];
    const double z0 = zs[lane], z1 = zs[lane + 64];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = r0 + r;
        double s = (double)l0[r] * z0 + (double)l1[r] * z1;
        s = wave_sum(s);
        if (lane == 0 && i < n) w[i] = (T)((double)w[i] - s);
    }
}

// Backward step b of L^T a = z:  a_b = inv(L_bb)^T z_b, workgroup 0 stores a_b, workgroup g updates its 256
// columns to the left:  z[j] -= sum_i L[b*128 + i][j] a_b[i].
template <typename T>
__global__ __launch_bounds__(256) void trsv_bwd_step_kernel(const T* __restrict__ L, long ldl, const T* __restrict__ invb,
                                                            int b, T* __restrict__ z, T* __restrict__ a) {
    __shared__ double zs[NB], as[NB], part[NB];
    const int tid = threadIdx.x;
    const int c0 = b * NB;
    if (tid < NB) zs[tid] = (double)z[c0 + tid];
    __syncthreads();
    {   // column j = tid & 127; the 128 rows are split in two halves over the 256 threads (inv is lower: i >= j)
        const int j = tid & (NB - 1), half = tid >> 7;
        double s = 0.0;
#pragma unroll 16
        for (int q = 0; q < 64; ++q) {
            const int i = half * 64 + q;
            const double v = (double)invb[(long)i * NB + j];
            s += (i >= j) ? v * zs[i] : 0.0;
        }
        if (half) part[j] = s;
        __syncthreads();
        if (!half) as[j] = s + part[j];
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < NB) a[c0 + tid] = (T)as[tid];
    const int j = blockIdx.x * 256 + tid;
    if (j < c0) {
        double s = 0.0;
#pragma unroll 32
        for (int i = 0; i < NB; ++i) s += (double)L[(long)(c0 + i) * ldl + j] * as[i];
        z[j] = (T)((double)z[j] - s);
    }
}

// part[rc][j] = sum_{i in row chunk rc} A[i][j] x[i]; tri: skip chunks above the diagonal
template <typename T>
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(const T* __restrict__ A, long lda, const T* __restrict__ x,
                                                             T* __restrict__ part, long ldp, int tri, long eA = 0, long ex = 0, long ep = 0) {
    A += blockIdx.z * eA; x += blockIdx.z * ex; part += blockIdx.z * ep;      // batched experts
    const int cc = blockIdx.x, rc = blockIdx.y;
    if (tri && rc < cc) return;
    __shared__ double xs[256];
    const int tid = threadIdx.x;
    xs[tid] = (double)x[rc * 256 + tid];
    __syncthreads();
    const int j = cc * 256 + tid;
    const T* a = A + (long)rc * 256 * lda + j;
    // lower-triangular operand: inside the diagonal chunk only rows of this column's 128-block and below count
    // (the mirrored 128-blocks above it are scratch of the triangular inverse)
    const int i0 = (tri && rc == cc) ? (tid & 128) : 0;
    double s = 0.0;
#pragma unroll 8
    for (int i = i0; i < 256; ++i) s += (double)a[(long)i * lda] * xs[i];
    part[(long)rc * ldp + j] = (T)s;
}

// out[j] = a + b * sum_{rc >= rc0(j)} part[rc][j]   (acc: out[j] += b * sum)
template <typename T>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ part, long ldp, int nrc, int cols,
                                                        T* __restrict__ out, int tri, double a, double b, int acc, long ep = 0, long eo = 0) {
    part += blockIdx.z * ep; out += blockIdx.z * eo;                          // batched experts
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    double s = 0.0;
    for (int rc = tri ? j / 256 : 0; rc < nrc; ++rc) s += (double)part[(long)rc * ldp + j];
    out[j] = (T)((acc ? (double)out[j] : a) + b * s);
}

// y[i] -= sum_{j < cols} A[i][j] x[j], wave per row (cols a multiple of 256), 16-byte loads
template <typename T>
__global__ __launch_bounds__(256) void gemv_n_sub_kernel(const T* __restrict__ A, long lda, int rows, int cols,
                                                         const T* __restrict__ x, T* __restrict__ y) {
    constexpr int VE = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = 0; r < 4; ++r) {
        const int i = blockIdx.x * 16 + wave * 4 + r;
        if (i >= rows) continue;
        const T* row = A + (long)i * lda;
        double s0 = 0.0, s1 = 0.0;
        for (int j = lane * VE; j < cols; j += 2 * 64 * VE) {
            const vec_t a0 = *reinterpret_cast<const vec_t*>(row + j), b0 = *reinterpret_cast<const vec_t*>(x + j);
            const bool two = j + 64 * VE < cols;
            const vec_t a1 = two ? *reinterpret_cast<const vec_t*>(row + j + 64 * VE) : a0;
            const vec_t b1 = two ? *reinterpret_cast<const vec_t*>(x + j + 64 * VE) : b0;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                s0 += (double)a0[e] * (double)b0[e];
                if (two) s1 += (double)a1[e] * (double)b1[e];
            }
        }
        const double s = wave_sum(s0 + s1);
        if (lane == 0) y[i] = (T)((double)y[i] - s);
    }
}

// y[i] = sum_{j < jend(i)} M[i][j] x[j], wave per row; jend = end of row i's 128-block (lower-triangular M).
// 16-byte loads, four independent partial sums per lane, and the long rows (bottom of the matrix) launched first.
template <typename T>
__global__ __launch_bounds__(256) void trmv_n_kernel(const T* __restrict__ M, long ldm, int n, const T* __restrict__ x,
                                                     T* __restrict__ y, long eM = 0, long ex = 0, long ey = 0) {
    M += blockIdx.z * eM; x += blockIdx.z * ex; y += blockIdx.z * ey;         // batched experts
    constexpr int VE = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = gridDim.x - 1 - blockIdx.x;
    for (int r = 0; r < 4; ++r) {
        const int i = b * 16 + wave * 4 + r;
        if (i >= n) continue;
        const int jend = (i / NB + 1) * NB;   // multiple of 128 = 64 lanes x VE (fp64) or 32 lanes x VE (fp32)
        const T* row = M + (long)i * ldm;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = lane * VE;
        for (; j + 3 * 64 * VE < jend; j += 4 * 64 * VE) {
            const vec_t a0 = *reinterpret_cast<const vec_t*>(row + j), b0 = *reinterpret_cast<const vec_t*>(x + j);
            const vec_t a1 = *reinterpret_cast<const vec_t*>(row + j + 64 * VE), b1 = *reinterpret_cast<const vec_t*>(x + j + 64 * VE);
            const vec_t a2 = *reinterpret_cast<const vec_t*>(row + j + 128 * VE), b2 = *reinterpret_cast<const vec_t*>(x + j + 128 * VE);
            const vec_t a3 = *reinterpret_cast<const vec_t*>(row + j + 192 * VE), b3 = *reinterpret_cast<const vec_t*>(x + j + 192 * VE);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                s0 += (double)a0[e] * (double)b0[e];
                s1 += (double)a1[e] * (double)b1[e];
                s2 += (double)a2[e] * (double)b2[e];
                s3 += (double)a3[e] * (double)b3[e];
            }
        }
        for (; j < jend; j += 64 * VE) {
            const vec_t a0 = *reinterpret_cast<const vec_t*>(row + j), b0 = *reinterpret_cast<const vec_t*>(x + j);
#pragma unroll
            for (int e = 0; e < VE; ++e) s0 += (double)a0[e] * (double)b0[e];
        }
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) y[i] = (T)s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void logdet_kernel(const T* __restrict__ L, long ldl, int n, double* __restrict__ out, double scale) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double s = 0.0;                 // scale = 2 on the factor's diagonal, -2 on its inverse's (log det K = 2 sum log L_ii)
    for (int i = tid; i < n; i += 256) s += scale * log((double)L[(long)i * ldl + i]);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[0] = red[0] + red[1] + red[2] + red[3];
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

// out[0] = 1/2 y^T alpha + 1/2 logdet + n/2 log 2pi   (the NLML from a log-determinant computed earlier)
template <typename T>
__global__ __launch_bounds__(256) void nlml_finish_kernel(const T* __restrict__ y, const T* __restrict__ alpha, int n,
                                                          const double* __restrict__ logdet, double* __restrict__ out) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += 0.5 * (double)y[i] * (double)alpha[i];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[0] = red[0] + red[1] + red[2] + red[3] + 0.5 * logdet[0] + 0.5 * (double)n * 1.83787706640934548356;
}

// Per-expert grBCM terms (gr_bcm.py:125-144) for nexp owned experts in one launch (round 5; the one-expert entry point is the same kernel
// with nexp = 1): out[0..2][j] (+)= sum_c beta_c, beta_c prec_c, beta_c prec_c mean_c with beta_c = 1/2 (log prec_c - log prec_g), or 1 for
// the committee's first expert (c == first; gr_bcm.py:132).  Thread j walks the experts in order, so the sums receive the terms in the
// order of one-by-one calls (bit-identical); rows c of beta_out / prec_out (leading dimension ldb) receive expert c's.
template <typename T>
__global__ __launch_bounds__(256) void grbcm_terms_batched_kernel(const T* __restrict__ mean_l, long emean, const T* __restrict__ var_l, long evar,
                                                                  const T* __restrict__ var_g, int m, int nexp, int first, int accumulate,
                                                                  double* __restrict__ out, long ldo, double* __restrict__ beta_out,
                                                                  double* __restrict__ prec_out, long ldb) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double pg = 1.0 / (double)var_g[j];
    const double lpg = log(pg);
    double s0 = accumulate ? out[j] : 0.0, s1 = accumulate ? out[ldo + j] : 0.0, s2 = accumulate ? out[2 * ldo + j] : 0.0;
    for (int c = 0; c < nexp; ++c) {
        const double pc = 1.0 / (double)var_l[c * evar + j];
        const double beta = (c == first) ? 1.0 : 0.5 * (log(pc) - lpg);
        if (beta_out) beta_out[c * ldb + j] = beta;
        if (prec_out) prec_out[c * ldb + j] = pc;
        const double t1 = beta * pc;
        if (c == 0 && !accumulate) { s0 = beta; s1 = t1; s2 = t1 * (double)mean_l[c * emean + j]; }
        else { s0 += beta; s1 += t1; s2 += t1 * (double)mean_l[c * emean + j]; }
    }
    out[j] = s0; out[ldo + j] = s1; out[2 * ldo + j] = s2;
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

// acc[i][j] (+)= 1/2 (beta_i + beta_j) P[i][j] on the lower triangle (gr_bcm.py:107-111); the padding keeps a unit diagonal
template <typename T>
__global__ __launch_bounds__(256) void weighted_prec_kernel(const T* __restrict__ P, long ldp, const double* __restrict__ beta,
                                                            int m, int m_pad, T* __restrict__ acc, long lda, int accumulate) {
    const int tc = blockIdx.x, tr = blockIdx.y;
    if (tc > tr) return;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int i = tr * 64 + (idx >> 6), j = tc * 64 + (idx & 63);
        if (i >= m_pad || j > i) continue;
        T* a = acc + (long)i * lda + j;
        if (i < m) {
            const double v = 0.5 * (beta[i] + beta[j]) * (double)P[(long)i * ldp + j];
            *a = (T)(accumulate ? (double)*a + v : v);
        } else if (!accumulate) {
            *a = (i == j) ? (T)1 : (T)0;
        }
    }
}

// A[i][j] = A[j][i] for j > i
template <typename T> __global__ __launch_bounds__(256) void symmetrize_kernel(T* __restrict__ A, long lda, int n) {
    const int tc = blockIdx.x, tr = blockIdx.y;
    if (tc < tr) return;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int i = tr * 64 + (idx >> 6), j = tc * 64 + (idx & 63);
        if (i < n && j < n && j > i) A[(long)i * lda + j] = A[(long)j * lda + i];
    }
}

// mean[j] = cov[j][j] * (sum_c beta prec mu + beta_0 prec_0 mu_0)   (gr_bcm.py:147)
template <typename T>
__global__ __launch_bounds__(256) void grbcm_finish_full_kernel(const double* __restrict__ sums, long lds, const T* __restrict__ mean_g,
                                                                const T* __restrict__ var_g, const T* __restrict__ cov, long ldc,
                                                                int m, T* __restrict__ mean) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double pg = 1.0 / (double)var_g[j];
    const double b0 = 1.0 - sums[j];
    mean[j] = (T)((double)cov[(long)j * ldc + j] * (sums[2 * lds + j] + b0 * pg * (double)mean_g[j]));
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
    p.tri = p.klo = p.khi = 0; p.krev = 0;
    p.sA = p.sB = p.sC = 0; p.batch = 1;
    p.nexp = 1; p.eA = p.eB = p.eC = 0; p.einfo = 0;
    p.part = nullptr; p.ldp = 0; p.info = nullptr; p.noxcd = 0;
    return p;
}

// inv_diag [n/128][128][128], then the flag words of the coupled chain (3 n/128 + 1 ints, and the in-kernel time log when
// PG_CS_TLOG is set), then -- only in the experimental recursive-panel mode (PG_PANEL_MODE=1) -- the work area of the panel step:
// W = inverse of the current outer panel's triangular factor (at most 2048 x 2048) and Xs = the panel's solved rows before they
// are copied back (n x at most 2048).  The default mode's workspace is n * 128 + 128 n/128 + 2048 elements (round 2 always
// carried W and Xs: +300 MB at n = 16384, 35x the default need of a 2048-point expert).
#define NBO_MAX 2048
static int pg_panel_mode_env() {
    static const int v = getenv("PG_PANEL_MODE") ? atoi(getenv("PG_PANEL_MODE")) : 0;
    return v;
}
static long pg_flag_elems(int n) { return 128L * (n / NB) + 2048; }   // elements of T (>= 4 bytes each)
long pg_potrf_worksize_impl(int n) {
    const long w = std::min<long>(n, NBO_MAX);
    return (long)n * NB + pg_flag_elems(n) + (pg_panel_mode_env() == 1 ? w * w + (long)n * w : 0);
}

template <typename T>
int pg_trtri_t(pg_ctx* ctx, hipStream_t st, int n, const T* L, long ldl, const T* invD, T* M, long ldm, int hmax, const ExpBatch* eb) {
    if (n <= 0 || n % NB) { pg_set_error("pg_trtri: n=%d is not a positive multiple of %d", n, NB); return -2; }
    const int nexp = eb ? eb->nexp : 1;
    const long eL = eb ? eb->eA : 0, eI = eb ? eb->eInv : 0, eMm = eb ? eb->eM : 0;
    hipLaunchKernelGGL(copy_blocks_kernel<T>, dim3(n / NB, 8, nexp), dim3(256), 0, st, invD, M, ldm, eI, eMm);
    LAUNCH_CHECK();
    // invariant: the diagonal is tiled by `nfull` inverted blocks of size h plus one smaller inverted block `rem`
    int rc;
    long h = NB, rem = 0;
    for (;;) {
        const long nfull = (n - rem) / h;
        if (nfull + (rem ? 1 : 0) <= 1) break;
        if (hmax && h >= hmax) break;   // block-diagonal inverse only: blocks of hmax (and a smaller last one)
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
            p.nexp = nexp; p.eA = eMm; p.eB = eL; p.eC = eMm;
            // few pairs of small blocks: 64 x 64 tiles give 4x the workgroups, each a quarter as long (h = 256: 42 -> 15 us)
            // (experts together: the launch carries nexp times the tiles)
            const bool small = batch * (h / 128) * (h2 / 128) * nexp <= 1024 && h2 % 64 == 0;   // swept 128 .. 4200
            if ((rc = pg_gemm<T>(ctx, st, small ? GEMM_TT_64 : GEMM_TT_128, p))) return rc;
            // X21 = -X22 S^T
            p = gp0<T>();
            p.M = (int)h2; p.N = (int)h; p.K = (int)h2;
            p.A = M + (r0 + h) * ldm + r0 + h; p.lda = ldm;
            p.B = M + r0 * ldm + r0 + h; p.ldb = ldm;
            p.C = M + (r0 + h) * ldm + r0; p.ldc = ldm;
            p.alpha = (T)-1; p.khi = 1; p.batch = (int)batch;
            p.sA = p.sB = p.sC = stride;
            p.nexp = nexp; p.eA = p.eB = p.eC = eMm;
            if ((rc = pg_gemm<T>(ctx, st, small ? GEMM_NT_64 : GEMM_NT_128, p))) return rc;
        }
        if (nfull & 1) rem = rem ? h + rem : h;   // the odd block merges with rem, or becomes the new rem
        h *= 2;
    }
    return 0;
}

// Top level of the triangular inverse for the split [0, h1) | [h1, h1 + h2): S = (L21 M11)^T into the mirrored block
// (first), then M21 = -M22 S^T (second).  The first product only needs L21 and M11.
template <typename T>
static int trtri_top(pg_ctx* ctx, hipStream_t st, int h1, int h2, const T* L, long ldl, T* M, long ldm, bool first, bool second) {
    int rc;
    if (first) {
        GemmP<T> p = gp0<T>();
        p.M = h1; p.N = h2; p.K = h1; p.A = M; p.lda = ldm; p.B = L + (long)h1 * ldl; p.ldb = ldl; p.C = M + h1; p.ldc = ldm; p.klo = 1;
        if ((rc = pg_gemm<T>(ctx, st, GEMM_TT_128, p))) return rc;
    }
    if (second) {
        GemmP<T> p = gp0<T>();
        p.M = h2; p.N = h1; p.K = h2; p.A = M + (long)h1 * ldm + h1; p.lda = ldm; p.B = M + h1; p.ldb = ldm;
        p.C = M + (long)h1 * ldm; p.ldc = ldm; p.alpha = (T)-1; p.khi = 1;
        if ((rc = pg_gemm<T>(ctx, st, GEMM_NT_128, p))) return rc;
    }
    return 0;
}

long long pg_wait_ticks(const pg_ctx* ctx, int n) {
    if (ctx->spin_ticks != 0) return ctx->spin_ticks;
    const double est_us = 62.0 * (n / NB) + (double)n * n * n / 3.0 / 35.0e6;
    return (long long)(std::max(50000.0, 20.0 * est_us) * 100.0);
}

static int pool_event(pg_ctx* ctx, int idx, hipEvent_t* ev) {
    if (idx >= ctx->npool) {
        const int want = idx + 16;
        hipEvent_t* p = (hipEvent_t*)realloc(ctx->pool, sizeof(hipEvent_t) * want);
        if (!p) { pg_set_error("out of memory for events"); return -3; }
        ctx->pool = p;
        for (int i = ctx->npool; i < want; ++i) PG_CHECK(hipEventCreateWithFlags(&ctx->pool[i], hipEventDisableTiming));
        ctx->npool = want;
    }
    *ev = ctx->pool[idx];
    return 0;
}

template <typename T>
int pg_potrf_t(pg_ctx* ctx, hipStream_t st, int n, T* A, long lda, T* invD, int* info, T* Minv, long ldm, const BuildReq<T>* build,
               const ExpBatch* eb, int col_off, int keep_info);

// Recursive top level of the fused factor-and-invert call (round 4).  For n >= ctx->rec_min (pg_set_recursive_split / PG_REC_MIN,
// default 16384: at 12288 the one-level schedule is 0.5 ms faster, at 20480 / 24576 this one by 1.2 / 2.1 ms) the matrix is split at
// h1 = n/2 (rounded to the padding unit) and everything that crosses the split is ONE large product of the GEMM core:
//     [L11, M11] = potrf_trtri(A11)                     (this function again; the coupled chain below 8192 rows)
//     L21 = K21 M11^T                                   (the panel "solve" as a product with the inverse we need anyway; K ranges)
//     A22 -= L21 L21^T                                  (ONE update with K = h1 instead of h1 / 1024 updates beside the chain)
//     [L22, M22] = potrf_trtri(A22)
//     M21 = -M22 (L21 M11)                              (the top level of the triangular inverse, as before)
// The same 2 n^3 / 3 flop as pg_potrf + pg_trtri, but the n^3 / 2 that cross the split run as four full-chip products at the core's
// own rate (K = h1 deep, no chain beside them) instead of h1 / 1024 right-looking updates that share the chip with the chain's short
// kernels (52 TFLOP/s in the update-bound phase against 66-69 for these products).  No extra memory: K21 is built (or copied) into
// the M21 block of the inverse's buffer -- free until the last product writes it -- so the panel product is out of place.
// A bad pivot in the leading half leaves info != 0: every later launch returns at once (GemmP::info), the second half keeps it.
static int pg_rec_split(const pg_ctx* ctx, int n) {
    if (ctx->rec_min <= 0 || n < ctx->rec_min || n < 2 * PG_PAD || ctx->bg || ctx->panel_mode) return 0;
    return ((n / 2 + PG_PAD - 1) / PG_PAD) * PG_PAD;
}

template <typename T>
static int potrf_trtri_rec(pg_ctx* ctx, hipStream_t st, int n, T* A, long lda, T* invD, int* info, T* Minv, long ldm,
                           const BuildReq<T>* build, int col_off) {
    const int h1 = pg_rec_split(ctx, n), h2 = n - h1;
    int rc;
    T* A21 = A + (long)h1 * lda;
    T* A22 = A21 + h1;
    T* M21 = Minv + (long)h1 * ldm;
    BuildReq<T> b1;
    if (build) { b1 = *build; b1.n_real = std::min(build->n_real, h1); }
    const bool side = build && ctx->lookahead && !ctx->prof_on && ctx->upd;
    if (side) PG_CHECK(hipEventRecord(ctx->ev[0], st));     // this call's start: what the trailing blocks' build has to wait for
    if ((rc = pg_potrf_t<T>(ctx, st, h1, A, lda, invD, info, Minv, ldm, build ? &b1 : nullptr, nullptr, col_off, 1))) return rc;
    if (build) {
        const int nr2 = std::max(0, build->n_real - h1);
        const T* X2 = build->X + (long)h1 * build->ldx;
        // K21 -> the M21 block (cross build: no noise, no jitter), K22 -> A22 (lower tiles).  On the update stream, behind the leading
        // half's last trailing update: HBM-bound work beside that factorisation's chain-bound tail and the MFMA-bound inverse.
        hipStream_t bs = side ? ctx->upd : st;
        if (side) PG_CHECK(hipStreamWaitEvent(bs, ctx->ev[0], 0));   // the blocks' previous readers ran on st before this call
        if ((rc = pg_kbuild<T>(bs, *build->spec, build->hp, X2, build->ldx, nr2, build->X, build->ldx, std::min(build->n_real, h1), build->d,
                               0, 0, 0, 0.0, M21, ldm, h2, h1)))
            return rc;
        if ((rc = pg_kbuild<T>(bs, *build->spec, build->hp, X2, build->ldx, nr2, X2, build->ldx, nr2, build->d, 1, 1, 0, build->jitter, A22,
                               lda, h2, h2)))
            return rc;
        if (side) {
            PG_CHECK(hipEventRecord(ctx->ev[1], bs));
            PG_CHECK(hipStreamWaitEvent(st, ctx->ev[1], 0));
        }
    } else {
        const long vecs = (long)h2 * h1 / (16 / sizeof(T));
        hipLaunchKernelGGL(copy_rows_kernel<T>, dim3((unsigned)std::min<long>((vecs + 255) / 256, 8192)), dim3(256), 0, st, A21, lda, M21, ldm,
                           h2, h1);
        LAUNCH_CHECK();
    }
    {   // L21 = K21 M11^T: M11 is lower triangular, tile column j needs k < (j + 1) 128
        GemmP<T> p = gp0<T>(); p.info = info;
        p.M = h2; p.N = h1; p.K = h1; p.A = M21; p.lda = ldm; p.B = Minv; p.ldb = ldm; p.C = A21; p.ldc = lda; p.khi = 2;
        if ((rc = pg_gemm<T>(ctx, st, GEMM_NT_128, p))) return rc;
    }
    {   // A22 -= L21 L21^T: 2080 equally long lower tiles at n = 16384 -- 4.06 rounds of the chip's 512 workgroup slots; pg_gemm ends such
        // a launch in quarter tiles (pg_gemm_mixed_kernel: 8.9 -> 8.0 ms)
        GemmP<T> p = gp0<T>(); p.info = info;
        p.M = p.N = h2; p.K = h1; p.A = A21; p.lda = lda; p.B = A21; p.ldb = lda; p.C = A22; p.ldc = lda;
        p.alpha = (T)-1; p.beta = (T)1; p.tri = 1;
        if ((rc = pg_gemm<T>(ctx, st, GEMM_NT_128, p))) return rc;
    }
    if ((rc = pg_potrf_t<T>(ctx, st, h2, A22, lda, invD + (long)(h1 / NB) * NB * NB, info, Minv + (long)h1 * ldm + h1, ldm, nullptr, nullptr,
                            col_off + h1, 1)))
        return rc;
    return trtri_top<T>(ctx, st, h1, h2, A, lda, Minv, ldm, true, true);
}

// Look-ahead: for outer panel o let Chain(o) = its 8 (U, leaf, T) steps, Sa(o) = update of panel o+1's columns by
// panel o (all rows below), Sb(o) = lower-tile SYRK of everything right of panel o+1 by panel o.
//   panel stream  (handle's high-priority stream)       : Chain(0) Sa(0) Chain(1) [wait Sb(0)] Sa(1) Chain(2) [wait Sb(1)] Sa(2) ...
//   update stream (handle's CU-masked stream)            :   [wait Chain(0)] Sb(0)     [wait Chain(1)] Sb(1) ...
// Sa(o) and Chain(o+1) touch only panel o+1's columns and Sb(o) only columns right of it, so they overlap; Sa(o)
// follows Sb(o-1), the previous writer of its tiles, so every tile still receives its updates in panel order.  The update
// stream may not use the last PG_RESERVED_CUS compute units: the chain's small kernels (and the leaf, which needs a
// whole CU's LDS) always find free CUs instead of queueing behind 280 us SYRK tiles.  Everything is joined back onto
// the caller's stream at the end.
template <typename T>
int pg_potrf_t(pg_ctx* ctx, hipStream_t st, int n, T* A, long lda, T* invD, int* info, T* Minv, long ldm, const BuildReq<T>* build,
               const ExpBatch* eb, int col_off, int keep_info) {
    if (n <= 0 || n % PG_PAD) { pg_set_error("pg_potrf: n=%d is not a positive multiple of %d", n, PG_PAD); return -2; }
    const int nexp = eb ? eb->nexp : 1;                       // batched experts: every launch below covers all of them
    const long eA = eb ? eb->eA : 0, eI = eb ? eb->eInv : 0;
    if (nexp < 1) { pg_set_error("pg_potrf: empty batch"); return -2; }
    auto batched = [&](GemmP<T>& p, long sa, long sb, long sc) { p.nexp = nexp; p.eA = sa; p.eB = sb; p.eC = sc; p.einfo = nexp > 1 ? 1 : 0; };
    if (!keep_info) PG_CHECK(hipMemsetAsync(info, 0, sizeof(int) * nexp, st));
    if (Minv && nexp == 1 && pg_rec_split(ctx, n) > 0) return potrf_trtri_rec<T>(ctx, st, n, A, lda, invD, info, Minv, ldm, build, col_off);
    const int NBO = pg_nbo(ctx, n);
    // Outer panel boundaries (uniform; the last one may be short).  Measured on the round-2 build and left out: cutting the
    // first panel in two (256 + NBO - 256 columns) so that the first big update starts after two leaves instead of eight,
    // and half-width panels over the last 4096 / 8192 columns -- both within +-0.2 ms at n = 12288 / 16384: the schedule
    // is throughput-bound on the whole (the chain and the updates share the chip at 62-77 % MFMA use per CU), so shifting
    // work between its two streams does not shorten it.
    // flag-coupled chain (chainstep.hip) for the panels with at most `sync_rows` rows left -- the chain-bound tail: leaves on the
    // panel stream, the rows below them on the handle's rows stream, coupled by flags in device memory instead of launches.
    // Its rows kernels update a block column by the previous panel too (the window of the left-looking product starts there), so
    // the per-panel update Sa disappears from the critical path; panels are at most 384 wide there (PG_CS_PANEL) to keep that product
    // shorter than a leaf.
    // Measured (MI355X, fp64, build + factor): n = 4096 2.28 -> 1.68 ms, 8192 6.30 -> 5.19, 16384 30.75 -> 29.13 with the last 8192
    // rows coupled (2048: 30.41, 4096: 30.03, 6144: 29.62, 12288: 30.14, all: 31.48 -- while the trailing update still fills the
    // chip the resident rows workgroups hold the slots it needs).  With the experimental background inverse of the fused call
    // running (PG_BG_STREAM=1), the rows workgroups starve beside its long tiles: n = 16384 fused 51.5 -> 52.5 ms, so above 8192
    // that configuration keeps the classic chain.
    static const int sync_env = getenv("PG_SYNC_ROWS") ? atoi(getenv("PG_SYNC_ROWS")) : -1;
    const int sync_rows = sync_env >= 0 ? sync_env : ((Minv && ctx->bg && n > 8192) ? 0 : 8192);
    hipStream_t rows_stream = ctx->rows;      // the handle's own (capi.hip)
    // Experts together take the coupled chain too (round 4) -- one leaf workgroup and one grid row of rows workgroups per expert, flag
    // words in each expert's own work buffer -- where the batch is still latency-bound: experts of at least 2048 points and at most
    // 24576 rows in all.  Measured (fit = build + factor + inverse + alpha, D = 16, coupled against classic, same box): 2 x 4096 3.18 / 3.83 ms,
    // 4 x 4096 5.21 / 5.47, 8 x 2048 2.05 / 2.35, 8 x 3072 4.85 / 5.00; beyond that the batch's skinny products fill the chip either way and
    // the resident kernels only take slots from the trailing updates (8 x 4096 9.32 / 9.06, 8 x 9216 78.7 / 75.5; 16 x 2048 and 3 x 8192
    // equal), and below 2048 points three panels are too few (8 x 1024 0.79 / 0.73).  PG_CS_BATCHED=0: never; 2: whenever it can run.
    static const int cs_batched = getenv("PG_CS_BATCHED") ? atoi(getenv("PG_CS_BATCHED")) : 1;
    const bool batch_cp = cs_batched && nexp <= 24 && pg_leaf_has_early() && (cs_batched == 2 || (n >= 2048 && (long)nexp * n <= 24576));
    const bool want_cp = ctx->lookahead && !ctx->prof_on && ctx->coupled && rows_stream && sync_rows > 0 && ctx->panel_mode == 0 &&
                         (nexp == 1 || batch_cp);
    std::vector<int> pb;      // panel o = columns [pb[o], pb[o+1])
    static const int cs_panel = getenv("PG_CS_PANEL") ? atoi(getenv("PG_CS_PANEL")) : 384;   // round 3, same box: 512 -> 384: n = 4096 1.64 -> 1.60 ms, 8192 5.10 -> 5.03, 16384 equal
    // experiment: wider coupled panels while the trailing update still bounds the step (deeper K for Sb), narrow ones in the chain-bound tail
    static const int cs_panel_wide = getenv("PG_CS_PANEL_WIDE") ? atoi(getenv("PG_CS_PANEL_WIDE")) : 0;
    static const int cs_wide_rows = getenv("PG_CS_WIDE_ROWS") ? atoi(getenv("PG_CS_WIDE_ROWS")) : 5120;
    for (int c = 0; c < n;) {
        pb.push_back(c);
        int w = NBO;
        if (want_cp && n - c <= sync_rows) w = std::min(NBO, (cs_panel_wide > 0 && n - c > cs_wide_rows) ? cs_panel_wide : cs_panel);
        c += w;
    }
    pb.push_back(n);
    const int npan = (int)pb.size() - 1;
    const bool la = ctx->lookahead && !ctx->prof_on && npan >= 3;
    hipStream_t ps = la ? ctx->aux : st;   // panel stream
    hipStream_t us = la ? ctx->upd : st;   // update stream
    hipEvent_t ev;
    int rc;
    // folded covariance build: the first panel's columns now, the rest on the update stream beside the first panel's chain
    const bool build_split = build && la && npan >= 2;
    if (build) {
        const int c1 = build_split ? pb[1] : n;
        if ((rc = pg_kbuild<T>(st, *build->spec, build->hp, build->X, build->ldx, build->n_real, build->X, build->ldx, build->n_real,
                               build->d, 1, 1, 0, build->jitter, A, lda, n, n, 0, c1, nexp, eb ? eb->eX : 0, eb ? eb->ehp : 0, eA)))
            return rc;
    }
    if (la) {
        if ((rc = pool_event(ctx, 0, &ev))) return rc;
        PG_CHECK(hipEventRecord(ev, st));
        PG_CHECK(hipStreamWaitEvent(ps, ev, 0));
        PG_CHECK(hipStreamWaitEvent(us, ev, 0));
    }
    if (build_split) {
        if ((rc = pg_kbuild<T>(us, *build->spec, build->hp, build->X, build->ldx, build->n_real, build->X, build->ldx, build->n_real,
                               build->d, 1, 1, 0, build->jitter, A, lda, n, n, pb[1], n, nexp, eb ? eb->eX : 0, eb ? eb->ehp : 0, eA)))
            return rc;
        if ((rc = pool_event(ctx, 7 + 2 * npan, &ev))) return rc;      // ev_build: every column right of the first panel exists
        PG_CHECK(hipEventRecord(ev, us));
    }
    // fused L^-1: split the diagonal at `split` columns; the leading part is inverted in the background once its
    // columns are final (after the chain of panel split/NBO - 1), together with the first top-level product
    // (below n = 5120 the cross-stream split costs more than the overlap returns: 3.66 vs 3.79 ms at n = 4096)
    const int split = (Minv && la && ctx->bg && n / NBO >= 4 && n >= 5120 && nexp == 1) ? ((n + NBO - 1) / NBO / 2) * NBO : 0;
    const long NBW = std::min<long>(n, NBO_MAX);
    T* flagw = invD + (long)n * NB;       // flag words of the coupled chain
    T* Wt = flagw + pg_flag_elems(n);     // (PG_PANEL_MODE=1 only) inverse of the current panel's triangular factor, leading dimension = panel width
    T* Xs = Wt + NBW * NBW;               // (PG_PANEL_MODE=1 only) the panel's solved rows (out of place), leading dimension = panel width
    const bool coupled = la && want_cp;
    int o_s = npan;                             // first coupled panel
    if (coupled)
        for (int o = 0; o < npan; ++o)
            if (n - pb[o] <= sync_rows) { o_s = o; break; }
    ctx->last_coupled = npan - o_s;
    const int nblk = n / NB;
    int* f_diag = reinterpret_cast<int*>(flagw);   // [nblk] workgroups that have published tile (b, b)
    int* f_done = f_diag + nblk;                // [nblk] leaf b has stored L_bb and its inverse
    int* f_brow = f_done + nblk;                // [nblk] workgroups that have published X[block row b+1, block column b]
    int* f_early = f_brow + nblk;               // [nblk] leaf b has stored the first 64 rows of its inverse (two-phase hand-over)
    int* f_browe = f_early + nblk;              // [nblk] workgroups that have published the first 64 columns of X[block row b+1, block column b]
    int* f_tmo = f_browe + nblk;                // sticky time-out word
    if (o_s < npan) ctx->chain_epoch = ctx->chain_epoch % 1000000 + 1;
    // budget of one wait (10 ns ticks): the caller's (pg_set_spin_budget / PG_CS_SPIN_US), or 20x what this factorisation takes on the
    // classic chain by a crude model (62 us per 128 columns + n^3 / 3 flop at 35 TFLOP/s), at least 50 ms -- a leaf's wait may cover a
    // whole trailing update of the part before the coupled region (3 ms at n = 16384), never a multiple of the call
    const CsWait cw = {f_tmo, ctx->tmo_dev, pg_wait_ticks(ctx, n), ctx->chain_epoch};
    const CsBatch cbat = {nexp, eA, eI, (long)(eI * (long)sizeof(T) / (long)sizeof(int))};   // flag stride in ints
    // two-phase hand-over (chainstep.hip): only the third leaf form raises the early flag
    static const bool two_phase_env = !(getenv("PG_CS_TWO_PHASE") && atoi(getenv("PG_CS_TWO_PHASE")) == 0);
    const bool two_phase = two_phase_env && pg_leaf_has_early();
    if (o_s < npan) {
        // (ps: behind the fork event; experts together: the same words in every expert's work buffer)
        PG_CHECK(hipMemset2DAsync(f_diag, (size_t)(nexp > 1 ? eI : pg_potrf_worksize_impl(n)) * sizeof(T), 0, (size_t)((5 * nblk + 1 + 3) / 4) * 16, (size_t)nexp, ps));
    }
    // Coupled panels while the trailing update still bounds the step (more than `sa_rows` rows right of the panel): the next panel's
    // columns take this panel's update as ONE product on the update stream ("Sa", then a flag for the next leaf) instead of through the
    // rows kernels' two-panel window.  There the rows workgroups live on the reserved CUs only (77 KB of LDS do not fit beside the 64 x 64
    // blocks of the trailing update), and their window products, squeezed onto 32 CUs, took as long as the update itself
    // (n = 8192, steps 8-10: 34 | 158 | 228 us, the leaf of the next panel not even finding a CU for 190 us).
    static const int sa_rows = getenv("PG_CS_SA_ROWS") ? atoi(getenv("PG_CS_SA_ROWS")) : 0;   // measured: slower at every setting (n = 8192: off 4.77, 6144: 4.83, 4608: 4.90, 3072: 4.98, 2048: 5.06 ms): the panel period there is the trailing update's own time
    auto sa_after = [&](int o) { return sa_rows > 0 && o >= o_s && o + 1 < npan && (n - pb[o + 1]) > sa_rows; };
    std::vector<char> nf_split(npan + 1, 0);   // Sb(o) was launched as NEAR + FAR
    // Deferred trailing block (round 5; pg_set_deferred_block / PG_DEFER=1, OFF by default: measured slower).
    // A right-looking factorisation spends its throughput work first and ends in a chain-bound tail beside an idle chip (n = 8192: the
    // last 19 of 64 steps).  With the switch on, the panels left of column cd = pb[od] (about n / 2) update only the columns left of
    // cd2 = pb[oe]; the block right of cd2 receives everything those panels owe it LATER, panel by panel, as products K = cd deep
    // ("BU(p)", one trapezoid launch per column panel p >= oe, its K range dealt to several workgroups per tile) on the update stream
    // between the NEAR / FAR launches of the second half.  oe >= od + 1: the two-panel window of a rows kernel that works on a deferred
    // panel then starts right of cd (nothing is applied twice).  Measured at n = 8192 (same box, potrf alone): 4.64 ms without, 4.78
    // with -- the first half shrinks (3.3 -> 2.7 ms in the kernel trace), but a column panel of the deferred block is a few hundred
    // tiles that run at 41-45 TFLOP/s beside the resident rows kernels (64 x 64 tiles; 128 x 128 with the K range dealt out: 5.36 ms),
    // no better than the K = 384 updates they replace, and the second half waits for them (2.6 ms against 1.7).
    static const int defer_min = getenv("PG_DEFER_MIN") ? atoi(getenv("PG_DEFER_MIN")) : 6144;
    static const double defer_frac = getenv("PG_DEFER_FRAC") ? atof(getenv("PG_DEFER_FRAC")) : 0.5;
    static const int defer_gap = getenv("PG_DEFER_GAP") ? std::max(1, atoi(getenv("PG_DEFER_GAP"))) : 1;
    static const int defer_lead = getenv("PG_DEFER_LEAD") ? std::max(1, atoi(getenv("PG_DEFER_LEAD"))) : 3;
    static const int nf_env0 = getenv("PG_CS_NEARFAR") ? atoi(getenv("PG_CS_NEARFAR")) : 1;
    int od = -1, oe = -1, next_bu = 0;
    if (ctx->defer && coupled && o_s == 0 && nexp == 1 && n >= defer_min && nf_env0 && sa_rows == 0) {
        int o = 0;
        while (o < npan && pb[o] < (int)(defer_frac * n)) ++o;
        if (o >= 3 && o + defer_gap + 2 < npan) { od = o; oe = o + defer_gap; next_bu = oe; }
    }
    const bool defer = od > 0;
    const int cd = defer ? pb[od] : 0, cd2 = defer ? pb[oe] : 0;
    ctx->last_deferred = defer ? npan - oe : 0;
    auto launch_bu = [&](int p) -> int {   // A[pb[p]:, panel p] -= L[pb[p]:, 0:cd] L[panel p, 0:cd]^T, lower tiles
        GemmP<T> q = gp0<T>(); q.info = info;
        q.M = n - pb[p]; q.N = pb[p + 1] - pb[p]; q.K = cd;
        q.A = A + (long)pb[p] * lda; q.lda = lda; q.B = q.A; q.ldb = lda; q.C = A + (long)pb[p] * lda + pb[p]; q.ldc = lda;
        q.alpha = (T)-1; q.beta = (T)1; q.tri = 1;
        static const long bu_thresh = getenv("PG_BU_TILE_THRESH") ? atol(getenv("PG_BU_TILE_THRESH")) : 1024;
        static const long bu_ksplit = getenv("PG_BU_KSPLIT") ? atol(getenv("PG_BU_KSPLIT")) : 1200;   // workgroups a launch should have
        const long tiles = (long)(q.M / 128) * (q.N / 128);
        const int variant = tiles < bu_thresh ? GEMM_NT_64 : GEMM_NT_128;
        // A panel of the deferred block is a few hundred tiles, each K = cd deep: one workgroup per tile walks 4224 columns in 210 us
        // whatever the launch's size.  The K range is dealt to `ks` workgroups per tile (the launch's batch dimension, all adding into
        // the same C through the no-return fp64 atomics of the beta = 1 epilogue).
        int ks = 1;
        if (sizeof(T) == 8 && bu_ksplit > 0 && !ctx->no_atomic_c) {
            const long bt = variant == GEMM_NT_64 ? 64 : 128, tn_ = q.N / bt, tm_ = q.M / bt;
            const long wgs = tn_ * (tn_ + 1) / 2 + (tm_ - tn_) * tn_;
            for (int k = 2; k <= 8; ++k)
                if (cd % (k * 32) == 0 && wgs * (k - 1) < bu_ksplit) ks = k;
        }
        if (ks > 1) { q.K = cd / ks; q.batch = ks; q.sA = q.sB = cd / ks; q.sC = 0; }
        int r = pg_gemm<T>(ctx, us, variant, q);
        if (r) return r;
        hipEvent_t e;
        if ((r = pool_event(ctx, 10 + 4 * npan + p, &e))) return r;   // ev_bu[p]
        PG_CHECK(hipEventRecord(e, us));
        return 0;
    };
    for (int o = 0; o < npan; ++o) {
        const int o0 = pb[o], oend = pb[o + 1];
        const bool cp = o >= o_s;
        const bool sa_this = cp && sa_after(o), sa_prev = cp && o > o_s && sa_after(o - 1);
        if (cp) {
            hipStream_t rs = rows_stream;
            if (o == o_s) {
                // everything so far that touched these columns ran on the panel stream or was waited for there
                if ((rc = pg_flagset(ps, f_diag + o0 / NB, PG_CS_NCRIT, nexp, cbat.eF))) return rc;
                if ((rc = pool_event(ctx, 4 + 2 * npan, &ev))) return rc;
                PG_CHECK(hipEventRecord(ev, ps));
                PG_CHECK(hipStreamWaitEvent(rs, ev, 0));
            }
            for (int k0 = o0; k0 < oend; k0 += NB) {
                const int kb = k0 / NB;
                T* inv = invD + (long)kb * NB * NB;
                if ((rc = pg_leaf_sync<T>(ps, A + (long)k0 * lda + k0, lda, inv, info, k0 + col_off, f_diag + kb, PG_CS_NCRIT, f_done + kb, cw,
                                          two_phase ? f_early + kb : nullptr, &cbat)))
                    return rc;
                if (n - k0 - NB <= 0) break;
                const int c = k0 + NB;                       // the block column this step brings up to date
                const int oc = c < oend ? o : o + 1;         // its panel
                // the classic part applied the panel before the first coupled one; so does Sa(o - 1) in the update-bound part
                const int wstart = (oc == o_s || (oc == o && sa_prev)) ? pb[oc] : pb[oc - 1];
                const bool last_sa = c == oend && sa_this;   // the next panel's first column is Sa(o)'s: this step only solves
                if (c == oend && oc >= 2 && !last_sa) {      // first touch of panel oc: Sb(oc - 2) wrote these columns last
                    // (its NEAR launch when that panel was a coupled one: nf_split[oc - 2])
                    if ((rc = pool_event(ctx, nf_split[oc - 2] ? 9 + 3 * npan + (oc - 2) : 2 + 2 * (oc - 2) + 1, &ev))) return rc;
                    PG_CHECK(hipStreamWaitEvent(rs, ev, 0));
                    if (defer && oc >= oe) {                 // a deferred panel: what the first half owes it arrives as BU(oc)
                        if ((rc = pool_event(ctx, 10 + 4 * npan + oc, &ev))) return rc;
                        PG_CHECK(hipStreamWaitEvent(rs, ev, 0));
                    }
                }
                if (c == oend && oc == 1 && build_split && !last_sa) {   // first touch of a column the folded build wrote on the update stream
                    if ((rc = pool_event(ctx, 7 + 2 * npan, &ev))) return rc;
                    PG_CHECK(hipStreamWaitEvent(rs, ev, 0));
                }
                if (k0 == o0 && sa_prev) {                   // this panel's columns were last written by Sa(o - 1) on the update stream
                    if ((rc = pool_event(ctx, 8 + 2 * npan + (o - 1), &ev))) return rc;
                    PG_CHECK(hipStreamWaitEvent(rs, ev, 0));
                }
                if ((rc = pg_rowstep<T>(rs, A, lda, n, wstart, k0, last_sa ? 0 : 1, inv, f_done + kb, f_brow + kb, f_diag + kb + 1, cw, info,
                                        (two_phase && !last_sa) ? f_early + kb : nullptr, f_browe + kb, 1, &cbat)))
                    return rc;
            }
        }
        const int mode = (oend < n && pg_panel_mode_env() == 1 && nexp == 1) ? ctx->panel_mode : 0;
        const bool v2 = mode == 1;
        const int tri_end = v2 ? oend : n;     // last row the panel stream's 128-column steps touch
        for (int k0 = o0; k0 < oend && !cp; k0 += NB) {
            T* Akk = A + (long)k0 * lda + k0;
            T* inv = invD + (long)(k0 / NB) * NB * NB;
            if (k0 > o0) {   // U: bring this column block up to date with the panel's earlier columns
                GemmP<T> p = gp0<T>(); p.info = info;
                p.M = tri_end - k0; p.N = NB; p.K = k0 - o0;
                p.A = A + (long)k0 * lda + o0; p.lda = lda; p.B = p.A; p.ldb = lda; p.C = Akk; p.ldc = lda;
                p.alpha = (T)-1; p.beta = (T)1;
                batched(p, eA, eA, eA);
                // fewer than two 64-row workgroups per CU: half the row tile keeps two waves on every SIMD (gemm.h); in the
                // chain-bound tail the launch's latency is what counts: 32 x 32 tiles with a 64-deep K tile
                static const int u32rows = getenv("PG_U32_ROWS") ? atoi(getenv("PG_U32_ROWS")) : 8192;
                int uv = (p.M <= u32rows && p.K % 64 == 0) ? GEMM_NT_32x32 : (p.M <= 12288 ? GEMM_NT_32x64 : GEMM_NT_64);
                // experts together: every launch carries nexp times the tiles, and the panel stream shares the chip with the batch's
                // trailing update -- throughput per tile counts there, not the launch's latency
                static const int buv = getenv("PG_BATCH_UV") ? atoi(getenv("PG_BATCH_UV")) : 2;   // 8 x 4096: 0: 9.54, 1: 9.25, 2: 9.04 ms
                if (nexp > 1 && buv == 1) uv = GEMM_NT_32x64;
                if (nexp > 1 && buv == 2 && (long)(p.M / 64) * 2 * nexp >= 256) uv = GEMM_NT_64;
                if ((rc = pg_gemm<T>(ctx, ps, uv, p))) return rc;
            }
            if ((rc = pg_leaf<T>(ps, Akk, lda, inv, NB, info, k0 + col_off, 0, nexp, eA, eI))) return rc;
            const int m = tri_end - k0 - NB;
            if (m > 0) {     // T: rows below <- rows below * inv(L_kk)^T (in place: one workgroup owns 64 full rows)
                GemmP<T> p = gp0<T>(); p.info = info;
                p.M = m; p.N = NB; p.K = NB; p.A = Akk + (long)NB * lda; p.lda = lda; p.B = inv; p.ldb = NB;
                p.C = Akk + (long)NB * lda; p.ldc = lda; p.khi = 2;
                batched(p, eA, eI, eA);
                if ((rc = pg_gemm<T>(ctx, ps, m <= 12288 ? GEMM_NT_32x128 : GEMM_NT_64x128, p))) return rc;
            }
        }
        if (v2) {
            const int pw = oend - o0, mb = n - oend;
            if ((rc = pg_trtri_t<T>(ctx, ps, pw, A + (long)o0 * lda + o0, lda, invD + (long)(o0 / NB) * NB * NB, Wt, (long)pw))) return rc;
            GemmP<T> p = gp0<T>(); p.info = info;
            p.M = mb; p.N = pw; p.K = pw; p.A = A + (long)oend * lda + o0; p.lda = lda; p.B = Wt; p.ldb = pw; p.C = Xs; p.ldc = pw;
            p.khi = 2;       // W is lower triangular (and what lies above its diagonal 128-blocks is scratch of the doubling)
            const long tiles = (long)(mb / 128) * (pw / 128);
            if ((rc = pg_gemm<T>(ctx, ps, tiles < 1024 ? GEMM_NT_64 : GEMM_NT_128, p))) return rc;
            const long vecs = (long)mb * pw / (16 / sizeof(T));
            hipLaunchKernelGGL(copy_rows_kernel<T>, dim3((unsigned)std::min<long>((vecs + 255) / 256, 4096)), dim3(256), 0, ps, Xs, (long)pw,
                               A + (long)oend * lda + o0, lda, mb, pw);
            LAUNCH_CHECK();
        }
        if (oend >= n) break;
        hipStream_t cs = cp ? rows_stream : ps; // the stream whose completion means Chain(o) is done
        if (la) {
            if ((rc = pool_event(ctx, 2 + 2 * o, &ev))) return rc;       // ev_chain[o]
            PG_CHECK(hipEventRecord(ev, cs));
            PG_CHECK(hipStreamWaitEvent(us, ev, 0));
            if (split && oend == split) {   // columns [0, split) of L are final in every row
                PG_CHECK(hipStreamWaitEvent(ctx->bg, ev, 0));
                if ((rc = pg_trtri_t<T>(ctx, ctx->bg, split, A, lda, invD, Minv, ldm))) return rc;
                if ((rc = trtri_top<T>(ctx, ctx->bg, split, n - split, A, lda, Minv, ldm, true, false))) return rc;
            }
        }
        const int o2 = (o + 2 <= npan) ? pb[o + 2] : n;   // first column right of panel o+1
        if (sa_this) {   // Sa(o) of a coupled panel: on the update stream (behind ev_chain[o]), then the flag the next panel's first leaf waits for
            GemmP<T> p = gp0<T>(); p.info = info;
            p.M = n - oend; p.N = o2 - oend; p.K = oend - o0;
            p.A = A + (long)oend * lda + o0; p.lda = lda; p.B = p.A; p.ldb = lda; p.C = A + (long)oend * lda + oend; p.ldc = lda;
            p.alpha = (T)-1; p.beta = (T)1;
            const long tiles = (long)(p.M / 128) * (p.N / 128);
            // PG_CS_SA_STREAM=1: on the ROWS stream behind the panel's last rows kernel (no event towards the next panel's rows kernels;
            // the update stream keeps nothing but the trailing updates, so Sa(o + 1) does not queue behind Sb(o))
            static const int sa_on_rows = getenv("PG_CS_SA_STREAM") ? atoi(getenv("PG_CS_SA_STREAM")) : 0;
            hipStream_t ss = sa_on_rows ? rows_stream : us;
            if (sa_on_rows && o >= 1) {   // these columns were last written by Sb(o - 1)
                if ((rc = pool_event(ctx, 2 + 2 * (o - 1) + 1, &ev))) return rc;
                PG_CHECK(hipStreamWaitEvent(ss, ev, 0));
            }
            if (sa_on_rows && o == 0 && build_split) {
                if ((rc = pool_event(ctx, 7 + 2 * npan, &ev))) return rc;
                PG_CHECK(hipStreamWaitEvent(ss, ev, 0));
            }
            if ((rc = pg_gemm<T>(ctx, ss, tiles < 1024 ? GEMM_NT_64 : GEMM_NT_128, p))) return rc;
            if ((rc = pg_flagset(ss, f_diag + oend / NB, PG_CS_NCRIT))) return rc;
            if ((rc = pool_event(ctx, 8 + 2 * npan + o, &ev))) return rc;
            PG_CHECK(hipEventRecord(ev, ss));
        }
        if (!cp) {   // Sa(o): panel o+1's columns -= panel o   (coupled panels: part of the rows kernels' left-looking product)
            GemmP<T> p = gp0<T>(); p.info = info;
            p.M = n - oend; p.N = o2 - oend; p.K = oend - o0;
            p.A = A + (long)oend * lda + o0; p.lda = lda; p.B = p.A; p.ldb = lda; p.C = A + (long)oend * lda + oend; p.ldc = lda;
            p.alpha = (T)-1; p.beta = (T)1;
            batched(p, eA, eA, eA);
            // few 128x128 tiles with a 1024-deep K loop leave most CUs idle: use 64x64 tiles then (4x the workgroups)
            const long tiles = (long)(p.M / 128) * (p.N / 128) * nexp;
            if (la && o > 0) {   // these columns were last written by Sb(o-1)
                if ((rc = pool_event(ctx, 2 + 2 * (o - 1) + 1, &ev))) return rc;
                PG_CHECK(hipStreamWaitEvent(cs, ev, 0));
            }
            if (o == 0 && build_split) {   // ... or, for the first panel, by the folded build on the update stream
                if ((rc = pool_event(ctx, 7 + 2 * npan, &ev))) return rc;
                PG_CHECK(hipStreamWaitEvent(cs, ev, 0));
            }
            // (64 x 64 below 1024 large tiles; re-swept with the eight-wave 128 x 128 blocks: 256 is 0.4-0.8 ms slower at 16384)
            if ((rc = pg_gemm<T>(ctx, la ? cs : us, tiles < 1024 ? GEMM_NT_64 : GEMM_NT_128, p))) return rc;
        }
        const int m2 = n - o2;
        // Coupled panels (round 4): Sb(o) in two launches, NEAR = panel o+2's columns (all rows below), then FAR = everything right of
        // them.  The last rows kernel of Chain(o + 1) touches panel o+2's first column and so had to wait for ALL of Sb(o), while
        // Sb(o + 1) waits for that chain: the update stream idled 58-60 us per panel (profiles/r03_potrf_kernel_trace_n8192.txt).
        // Now that rows kernel waits for NEAR(o) only, Chain(o + 1) ends beside FAR(o), and NEAR(o + 1) queues right behind it.
        static const int nf_env = getenv("PG_CS_NEARFAR") ? atoi(getenv("PG_CS_NEARFAR")) : 1;
        const int o3 = (o + 3 <= npan) ? pb[o + 3] : n;
        const bool near_far = cp && nf_env && la && m2 > 0 && o3 < n;
        if (near_far) {
            const bool dfr = defer && o < od;              // a panel of the first half: nothing right of cd2
            GemmP<T> p = gp0<T>(); p.info = info;
            if (!(dfr && o + 2 >= oe)) {
                nf_split[o] = 1;
                p.M = n - o2; p.N = o3 - o2; p.K = oend - o0;
                p.A = A + (long)o2 * lda + o0; p.lda = lda; p.B = p.A; p.ldb = lda; p.C = A + (long)o2 * lda + o2; p.ldc = lda;
                p.alpha = (T)-1; p.beta = (T)1;
                batched(p, eA, eA, eA);
                const long tiles = (long)(p.M / 128) * (p.N / 128) * nexp;
                if ((rc = pg_gemm<T>(ctx, us, tiles < 1024 ? GEMM_NT_64 : GEMM_NT_128, p))) return rc;
                if ((rc = pool_event(ctx, 9 + 3 * npan + o, &ev))) return rc;   // ev_near[o]
                PG_CHECK(hipEventRecord(ev, us));
            }
            T* P = A + (long)o3 * lda + o0;
            p = gp0<T>(); p.info = info;
            p.M = p.N = n - o3; p.K = oend - o0; p.A = P; p.lda = lda; p.B = P; p.ldb = lda;
            p.C = A + (long)o3 * lda + o3; p.ldc = lda;
            p.alpha = (T)-1; p.beta = (T)1; p.tri = 1;
            if (dfr) p.N = std::max(0, cd2 - o3);          // the trapezoid left of cd2 (all rows)
            batched(p, eA, eA, eA);
            const long tn_ = p.N / 128, tm_ = p.M / 128;
            const long ftiles = (tn_ * (tn_ + 1) / 2 + (tm_ - tn_) * tn_) * nexp;
            static const long sb_thresh2 = getenv("PG_SB_TILE_THRESH") ? atol(getenv("PG_SB_TILE_THRESH")) : 2048;
            if (p.N > 0 && (rc = pg_gemm<T>(ctx, us, ftiles < sb_thresh2 ? GEMM_NT_64 : GEMM_NT_128, p))) return rc;
        } else if (m2 > 0) {  // Sb(o)
            T* P = A + (long)o2 * lda + o0;
            GemmP<T> p = gp0<T>(); p.info = info;
            p.M = p.N = m2; p.K = oend - o0; p.A = P; p.lda = lda; p.B = P; p.ldb = lda;
            p.C = A + (long)o2 * lda + o2; p.ldc = lda;
            p.alpha = (T)-1; p.beta = (T)1; p.tri = 1;
            batched(p, eA, eA, eA);
            const long tiles = (long)(m2 / 128) * (m2 / 128 + 1) / 2 * nexp;
            // (threshold swept 512 / 1024 / 2048: 2048 is 2 % faster at n = 8192 and neutral at 16384)
            // (re-swept with the eight-wave blocks, 2048 / 1024 / 512 / 256: 2048 stays best at 8192, neutral at 16384)
            static const long sb_thresh = getenv("PG_SB_TILE_THRESH") ? atol(getenv("PG_SB_TILE_THRESH")) : 2048;
            if ((rc = pg_gemm<T>(ctx, us, tiles < sb_thresh ? GEMM_NT_64 : GEMM_NT_128, p))) return rc;
        }
        if (la) {
            if ((rc = pool_event(ctx, 2 + 2 * o + 1, &ev))) return rc;   // ev_sb[o]
            PG_CHECK(hipEventRecord(ev, us));
        }
        if (defer && o >= od - 1) {   // the first half is final (ev_chain[od - 1], waited for above): the deferred block's column panels
            for (int i = 0; i < (o == od - 1 ? defer_lead : 1) && next_bu < npan; ++i)
                if ((rc = launch_bu(next_bu++))) return rc;
        }
    }
    if (la) {   // join both streams back onto the caller's stream
        if ((rc = pool_event(ctx, 1, &ev))) return rc;
        PG_CHECK(hipEventRecord(ev, ps));
        PG_CHECK(hipStreamWaitEvent(st, ev, 0));
        if ((rc = pool_event(ctx, 2 + 2 * npan, &ev))) return rc;
        PG_CHECK(hipEventRecord(ev, us));
        PG_CHECK(hipStreamWaitEvent(st, ev, 0));
        if (o_s < npan) {   // the last leaf only waited for the workgroups that own its tile
            if ((rc = pool_event(ctx, 5 + 2 * npan, &ev))) return rc;
            if (rows_stream != st) {
                PG_CHECK(hipEventRecord(ev, rows_stream));
                PG_CHECK(hipStreamWaitEvent(st, ev, 0));
            }
        }
        if (split) {
            if ((rc = pool_event(ctx, 3 + 2 * npan, &ev))) return rc;
            PG_CHECK(hipEventRecord(ev, ctx->bg));
            PG_CHECK(hipStreamWaitEvent(st, ev, 0));
        }
    }
    if (Minv) {
        if (!split) return pg_trtri_t<T>(ctx, st, n, A, lda, invD, Minv, ldm, 0, eb);
        // trailing part of the diagonal, then the second top-level product
        const long off = split;
        if ((rc = pg_trtri_t<T>(ctx, st, n - split, A + off * lda + off, lda, invD + (off / NB) * NB * NB, Minv + off * ldm + off, ldm)))
            return rc;
        return trtri_top<T>(ctx, st, split, n - split, A, lda, Minv, ldm, false, true);
    }
    return 0;
}

// L L^T x = y for one right-hand side.  Small n: one fused step per 128 columns.  From n = 2048: blocks of HB = 1024
// columns against their inverses -- the recursive doubling of pg_trtri stopped at HB, stored as the diagonal blocks of a
// virtual matrix with leading dimension HB (block b at b HB (HB + 1): the blocks do not overlap) -- so a sweep is
// 2-4 launches per 1024 columns instead of one per 128 (n = 8192: 3.7 -> 0.7 ms).
#define HB 1024
long pg_potrs_vec_worksize_impl(int n) { return n < 2 * HB ? 2L * n : (long)n * (2 + (HB + 1) + 4); }

template <typename T>
int pg_potrs_vec_t(pg_ctx* ctx, hipStream_t st, int n, const T* L, long ldl, const T* invD, const T* y, T* x, T* work) {
    if (n <= 0 || n % PG_PAD) { pg_set_error("pg_potrs_vec: n=%d is not a positive multiple of %d", n, PG_PAD); return -2; }
    T* w = work;          // running right-hand side of the forward sweep
    T* z = work + n;      // forward result / running right-hand side of the backward sweep
    PG_CHECK(hipMemcpyAsync(w, y, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, st));
    if (n < 2 * HB) {
        const int nb = n / NB;
        for (int b = 0; b < nb; ++b)      // L z = y
            hipLaunchKernelGGL(trsv_fwd_step_kernel<T>, dim3(std::max(1, (n - (b + 1) * NB) / 64)), dim3(256), 0, st, L, ldl,
                               invD + (long)b * NB * NB, b, n, w, z);
        for (int b = nb - 1; b >= 0; --b) // L^T a = z
            hipLaunchKernelGGL(trsv_bwd_step_kernel<T>, dim3(std::max(1, (b * NB + 255) / 256)), dim3(256), 0, st, L, ldl,
                               invD + (long)b * NB * NB, b, z, x);
        LAUNCH_CHECK();
        return 0;
    }
    T* W = work + 2L * n;                 // block inverses, leading dimension HB
    T* part = W + (long)n * (HB + 1);     // [4][n] partial sums of the transposed products
    int rc;
    if ((rc = pg_trtri_t<T>(ctx, st, n, L, ldl, invD, W, HB, HB))) return rc;
    for (int o = 0; o < n; o += HB) {     // L z = y
        const int s = std::min(HB, n - o), below = n - o - s;
        hipLaunchKernelGGL(trmv_n_kernel<T>, dim3((s + 15) / 16), dim3(256), 0, st, W + (long)o * (HB + 1), (long)HB, s, w + o, z + o);
        if (below > 0)
            hipLaunchKernelGGL(gemv_n_sub_kernel<T>, dim3((below + 15) / 16), dim3(256), 0, st, L + (long)(o + s) * ldl + o, ldl,
                               below, s, z + o, w + o + s);
    }
    const int last = ((n - 1) / HB) * HB;
    for (int o = last; o >= 0; o -= HB) { // L^T x = z
        const int s = std::min(HB, n - o);
        hipLaunchKernelGGL(gemv_t_partial_kernel<T>, dim3(s / 256, s / 256), dim3(256), 0, st, W + (long)o * (HB + 1), (long)HB,
                           z + o, part, (long)n, 1);
        hipLaunchKernelGGL(colreduce_kernel<T>, dim3(s / 256), dim3(256), 0, st, part, (long)n, s / 256, s, x + o, 1, 0.0, 1.0, 0);
        if (o > 0) {                      // z[0:o] -= L[o:o+s, 0:o]^T x[o:o+s]
            hipLaunchKernelGGL(gemv_t_partial_kernel<T>, dim3(o / 256, s / 256), dim3(256), 0, st, L + (long)o * ldl, ldl, x + o,
                               part, (long)n, 0);
            hipLaunchKernelGGL(colreduce_kernel<T>, dim3(o / 256), dim3(256), 0, st, part, (long)n, s / 256, o, z, 0, 0.0, -1.0, 1);
        }
    }
    LAUNCH_CHECK();
    return 0;
}

// Matrix right-hand sides: V = L^-1 B (solve_t == 0: the triangular half) or X = K^-1 B = L^-T (L^-1 B).  L^-1 is formed in `work`
// (n x n) by pg_trtri unless the caller hands one in; both halves are then products of the GEMM core with K ranges: L^-1 is lower
// triangular, so row tile i of L^-1 B stops at column i + 128 and row tile i of L^-T V starts at row i.
template <typename T>
int pg_potrs_t(pg_ctx* ctx, hipStream_t st, int n, int nrhs, const T* L, long ldl, const T* invD, const T* Minv, long ldm, const T* B,
               long ldb, T* X, long ldx, T* work, int both) {
    if (n <= 0 || n % PG_PAD || nrhs <= 0 || nrhs % 128) {
        pg_set_error("pg_potrs: n=%d must be a multiple of %d and nrhs=%d of 128", n, PG_PAD, nrhs);
        return -2;
    }
    int rc;
    const T* W = Minv;
    long ldw = ldm;
    T* V = work;                                   // [n][nrhs] when both halves run
    if (!W) {
        if ((rc = pg_trtri_t<T>(ctx, st, n, L, ldl, invD, work, (long)n))) return rc;
        W = work; ldw = n; V = work + (long)n * n;
    }
    GemmP<T> p = gp0<T>();
    p.M = n; p.N = nrhs; p.K = n; p.A = W; p.lda = ldw; p.B = B; p.ldb = ldb; p.khi = 1;
    p.C = both ? V : X; p.ldc = both ? nrhs : ldx;
    if ((rc = pg_gemm<T>(ctx, st, GEMM_NN_128, p))) return rc;
    if (!both) return 0;
    p = gp0<T>();
    p.M = n; p.N = nrhs; p.K = n; p.A = W; p.lda = ldw; p.B = V; p.ldb = nrhs; p.C = X; p.ldc = ldx; p.klo = 1;
    return pg_gemm<T>(ctx, st, GEMM_TN_128, p);
}

template <typename T>
int pg_lauum_t(pg_ctx* ctx, hipStream_t st, int n, const T* M, long ldm, T* Kinv, long ldk, const ExpBatch* eb) {
    if (n <= 0 || n % PG_PAD) { pg_set_error("pg_lauum: n=%d is not a positive multiple of %d", n, PG_PAD); return -2; }
    GemmP<T> p = gp0<T>();
    p.M = p.N = p.K = n; p.A = M; p.lda = ldm; p.B = M; p.ldb = ldm; p.C = Kinv; p.ldc = ldk;
    p.tri = 1; p.klo = 1;
    static const int krev_env = getenv("PG_LAUUM_KREV") ? atoi(getenv("PG_LAUUM_KREV")) : 1;
    p.krev = krev_env;
    if (eb) { p.nexp = eb->nexp; p.eA = p.eB = eb->eM; p.eC = eb->eA; }      // experts together: Minv + e eM -> Kinv + e eA
    return pg_gemm<T>(ctx, st, GEMM_TN_128, p);
}

template <typename T>
int pg_trmv_t(pg_ctx*, hipStream_t st, int n, const T* M, long ldm, int trans, const T* x, T* y, T* work) {
    if (n <= 0 || n % PG_PAD) { pg_set_error("pg_trmv: n=%d is not a positive multiple of %d", n, PG_PAD); return -2; }
    if (!trans) {
        hipLaunchKernelGGL(trmv_n_kernel<T>, dim3((n + 15) / 16), dim3(256), 0, st, M, ldm, n, x, y);
    } else {
        hipLaunchKernelGGL(gemv_t_partial_kernel<T>, dim3(n / 256, n / 256), dim3(256), 0, st, M, ldm, x, work, (long)n, 1);
        hipLaunchKernelGGL(colreduce_kernel<T>, dim3(n / 256), dim3(256), 0, st, work, (long)n, n / 256, n, y, 1, 0.0, 1.0, 0);
    }
    LAUNCH_CHECK();
    return 0;
}

// out[e * eo] = 1/2 y_e^T alpha_e - sum_i log Minv_e[i][i] + n/2 log 2pi: every expert's NLML from its inverse factor's diagonal
// (log det K = -2 sum log (L^-1)_ii) and its weights, one workgroup per expert
template <typename T>
__global__ __launch_bounds__(256) void nlml_batched_kernel(const T* __restrict__ M, long ldm, long eM, const T* __restrict__ y, long ey,
                                                           const T* __restrict__ alpha, long ea, int n, double* __restrict__ out, long eo) {
    __shared__ double red[4];
    const int e = blockIdx.x, tid = threadIdx.x;
    M += e * eM; y += e * ey; alpha += e * ea;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += 0.5 * (double)y[i] * (double)alpha[i] - log((double)M[(long)i * ldm + i]);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[e * eo] = red[0] + red[1] + red[2] + red[3] + 0.5 * (double)n * 1.83787706640934548356;
}

// alpha_e = Minv_e^T (Minv_e y_e) for nexp experts in three launches (u_e, work_e: scratch of n and (n/256) n elements per expert);
// with `out` a fourth launch leaves every expert's NLML at out[e * eo] (n_real points each)
template <typename T>
int pg_alpha_batched_t(hipStream_t st, int n, const T* M, long ldm, long eM, const T* y, long ey, T* u, long eu, T* alpha, long ea, T* work,
                       long ew, int nexp, int n_real, double* out, long eo) {
    if (n <= 0 || n % PG_PAD || nexp < 1) { pg_set_error("pg_alpha_batched: n=%d (multiple of %d), nexp=%d", n, PG_PAD, nexp); return -2; }
    hipLaunchKernelGGL(trmv_n_kernel<T>, dim3((n + 15) / 16, 1, nexp), dim3(256), 0, st, M, ldm, n, y, u, eM, ey, eu);
    hipLaunchKernelGGL(gemv_t_partial_kernel<T>, dim3(n / 256, n / 256, nexp), dim3(256), 0, st, M, ldm, (const T*)u, work, (long)n, 1, eM, eu, ew);
    hipLaunchKernelGGL(colreduce_kernel<T>, dim3(n / 256, 1, nexp), dim3(256), 0, st, (const T*)work, (long)n, n / 256, n, alpha, 1, 0.0, 1.0, 0, ew, ea);
    if (out) hipLaunchKernelGGL(nlml_batched_kernel<T>, dim3(nexp), dim3(256), 0, st, M, ldm, eM, y, ey, (const T*)alpha, ea, n_real, out, eo);
    LAUNCH_CHECK();
    return 0;
}

template <typename T> int pg_logdet_t(hipStream_t st, int n, const T* L, long ldl, double* out) {
    hipLaunchKernelGGL(logdet_kernel<T>, dim3(1), dim3(256), 0, st, L, ldl, n, out, 2.0);
    LAUNCH_CHECK();
    return 0;
}

template <typename T>
int pg_nlml_value_t(hipStream_t st, int n, const T* L, long ldl, const T* y, const T* alpha, double* out) {
    hipLaunchKernelGGL(nlml_value_kernel<T>, dim3(1), dim3(256), 0, st, L, ldl, y, alpha, n, out);
    LAUNCH_CHECK();
    return 0;
}

// alpha = Minv^T (Minv y) and the NLML, on the handle's side stream: the two triangular mat-vecs are HBM-bound (2 x 8 n^2 / 2 B)
// and overlap with the MFMA-bound L^-T L^-1 the caller enqueues next.  The log-determinant rides along, from the diagonal of
// L^-1 (K^-1 is about to overwrite the factor).  ctx->side_pending makes the next entry point that may read alpha / out wait.
template <typename T>
int pg_alpha_nlml_async_t(pg_ctx* ctx, hipStream_t st, int n_real, int n, const T* L, long ldl, const T* Minv, long ldm, const T* y,
                          T* u, T* alpha, T* work, double* out) {
    if (n <= 0 || n % PG_PAD) { pg_set_error("pg_alpha_nlml_async: n=%d is not a positive multiple of %d", n, PG_PAD); return -2; }
    hipStream_t side = (ctx->lookahead && !ctx->prof_on) ? ctx->aux : st;
    if (side != st) {
        PG_CHECK(hipEventRecord(ctx->ev[4], st));
        PG_CHECK(hipStreamWaitEvent(side, ctx->ev[4], 0));
    }
    // the log-determinant from the inverse's diagonal (1 / L_ii): the factor's buffer may be overwritten by the caller's next
    // call (K^-1 goes there), the inverse's is only read
    (void)L; (void)ldl;
    hipLaunchKernelGGL(logdet_kernel<T>, dim3(1), dim3(256), 0, side, Minv, ldm, n_real, out + 1, -2.0);
    LAUNCH_CHECK();
    int rc;
    if ((rc = pg_trmv_t<T>(ctx, side, n, Minv, ldm, 0, y, u, work))) return rc;
    if ((rc = pg_trmv_t<T>(ctx, side, n, Minv, ldm, 1, u, alpha, work))) return rc;
    hipLaunchKernelGGL(nlml_finish_kernel<T>, dim3(1), dim3(256), 0, side, y, alpha, n_real, out + 1, out);
    LAUNCH_CHECK();
    if (side != st) {
        PG_CHECK(hipEventRecord(ctx->ev[5], side));
        ctx->side_pending = 1;
        ctx->side_owner = st;
    }
    return 0;
}

// y[j] = sum_i A[j][i] x[i] for a row-major A [m x n]: one wave per row, 16-byte loads (n a multiple of 256)
template <typename T>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const T* __restrict__ A, long lda, int n, const T* __restrict__ x,
                                                        T* __restrict__ y, long eA = 0, long ex = 0, long ey = 0) {
    A += blockIdx.y * eA; x += blockIdx.y * ex; y += blockIdx.y * ey;         // batched experts
    constexpr int VE = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const T* a = A + (long)row * lda;
    T s = (T)0;
    for (int k = lane * VE; k < n; k += 64 * VE) {
        const vec_t av = *reinterpret_cast<const vec_t*>(a + k);
        const vec_t xv = *reinterpret_cast<const vec_t*>(x + k);
#pragma unroll
        for (int e = 0; e < VE; ++e) s += av[e] * xv[e];
    }
    s = wave_sum(s);
    if (lane == 0) y[row] = s;
}

// The same with the cross-covariance stored test-point-major: Kt[m_pad x n_pad] = k(Xp, X).  The variance product then has a
// K-contiguous B operand (C = Minv Kt^T, the NT form: 71 vs 66 TFLOP/s for NN on uniform products) and the mean is a row-wise
// mat-vec.
template <typename T>
int pg_predict_mean_q_kt_t(pg_ctx* ctx, hipStream_t st, int n, int m, const T* Kt, long ldkt, const T* M, long ldm,
                           const T* alpha, T* mean, T* q, double kss, T* work) {
    if (n % PG_PAD || m % 256 || n <= 0 || m <= 0) { pg_set_error("pg_predict_mean_q_kt: n_pad=%d m_pad=%d must be multiples of 256", n, m); return -2; }
    hipLaunchKernelGGL(gemv_rows_kernel<T>, dim3(m / 4), dim3(256), 0, st, Kt, ldkt, n, alpha, mean);
    LAUNCH_CHECK();
    if (q) {
        GemmP<T> p = gp0<T>();
        p.M = n; p.N = m; p.K = n; p.A = M; p.lda = ldm; p.B = Kt; p.ldb = ldkt; p.khi = 1;
        p.part = work; p.ldp = m;
        int rc = pg_gemm<T>(ctx, st, GEMM_NT_128_SS, p);
        if (rc) return rc;
        hipLaunchKernelGGL(colreduce_kernel<T>, dim3(m / 256), dim3(256), 0, st, work, (long)m, n / 64, m, q, 0, kss, -1.0, 0);
        LAUNCH_CHECK();
    }
    return 0;
}

// var_e[j] = kss_e - sum_rc part_e[rc][j] with kss_e = sum sigma_c^2 + sum sigma_n^2 of expert e's hyper-parameters: the constant diagonal
// of K** (gpr.py:98: White_noise sees xp = None), formed here so that the batched prediction needs no host arithmetic per expert.
// (products and sums kept apart -- no FMA contraction -- so that the value is the one the host computes for the one-expert call)
template <typename T>
__global__ __launch_bounds__(256) void predict_var_reduce_kernel(const T* __restrict__ part, long ldp, int nrc, int cols, T* __restrict__ out,
                                                                 pg_covspec spec, const double* __restrict__ hp, long ehp, long ep, long eo) {
    part += blockIdx.z * ep; out += blockIdx.z * eo; hp += blockIdx.z * ehp;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < spec.ncomp; ++c) s1 = __dadd_rn(s1, __dmul_rn(hp[spec.off[c]], hp[spec.off[c]]));
    for (int c = 0; c < spec.nnoise; ++c) s2 = __dadd_rn(s2, __dmul_rn(hp[spec.noise_off[c]], hp[spec.noise_off[c]]));
    const double kss = __dadd_rn(s1, s2);
    double s = 0.0;
    for (int rc = 0; rc < nrc; ++rc) s += (double)part[(long)rc * ldp + j];
    out[j] = (T)(kss - s);
}

// The diagonal prediction of ALL experts of a batched model in three launches (round 5): the reference predicts a batched model with one
// batched kernel / bmm / cholesky_solve (gpr.py:76-106 on x [nc, n, d]).  Expert e: Kt + e ekt (its test-point-major cross-covariance),
// Minv + e em, alpha + e ea -> mean + e emean, var + e evar; work: (n/64) m elements per expert at stride ew.  The variance product is
// the one-expert launch with the expert as the core's second batch level: per expert the same tiles in the same k order, bit for bit.
template <typename T>
int pg_predict_mean_q_kt_batched_t(pg_ctx* ctx, hipStream_t st, int n, int m, const T* Kt, long ldkt, long ekt, const T* M, long ldm, long em,
                                   const T* alpha, long ea, T* mean, long emean, T* q, long evar, const pg_covspec& spec, const double* hp,
                                   long ehp, T* work, long ew, int nexp) {
    if (n % PG_PAD || m % 256 || n <= 0 || m <= 0) { pg_set_error("pg_predict_mean_q_kt_batched: n_pad=%d m_pad=%d must be multiples of 256", n, m); return -2; }
    if (nexp < 1 || nexp > 65535) { pg_set_error("pg_predict_mean_q_kt_batched: 1 <= nexp <= 65535"); return -2; }
    hipLaunchKernelGGL(gemv_rows_kernel<T>, dim3(m / 4, nexp), dim3(256), 0, st, Kt, ldkt, n, alpha, mean, ekt, ea, emean);
    LAUNCH_CHECK();
    if (q) {
        GemmP<T> p = gp0<T>();
        p.M = n; p.N = m; p.K = n; p.A = M; p.lda = ldm; p.B = Kt; p.ldb = ldkt; p.khi = 1;
        p.part = work; p.ldp = m;
        p.nexp = nexp; p.eA = em; p.eB = ekt; p.eC = ew;
        int rc = pg_gemm<T>(ctx, st, GEMM_NT_128_SS, p);
        if (rc) return rc;
        hipLaunchKernelGGL(predict_var_reduce_kernel<T>, dim3(m / 256, 1, nexp), dim3(256), 0, st, (const T*)work, (long)m, n / 64, m, q, spec, hp, ehp, ew, evar);
        LAUNCH_CHECK();
    }
    return 0;
}

template <typename T>
int pg_predict_mean_q_t(pg_ctx* ctx, hipStream_t st, int n, int m, const T* Ks, long ldks, const T* M, long ldm,
                        const T* alpha, T* mean, T* q, double kss, T* work) {
    if (n % PG_PAD || m % 256 || n <= 0 || m <= 0) { pg_set_error("pg_predict_mean_q: n_pad=%d m_pad=%d must be multiples of 256", n, m); return -2; }
    hipLaunchKernelGGL(gemv_t_partial_kernel<T>, dim3(m / 256, n / 256), dim3(256), 0, st, Ks, ldks, alpha, work, (long)m, 0);
    hipLaunchKernelGGL(colreduce_kernel<T>, dim3(m / 256), dim3(256), 0, st, work, (long)m, n / 256, m, mean, 0, 0.0, 1.0, 0);
    LAUNCH_CHECK();
    if (q) {
        GemmP<T> p = gp0<T>();
        p.M = n; p.N = m; p.K = n; p.A = M; p.lda = ldm; p.B = Ks; p.ldb = ldks; p.khi = 1;
        p.part = work; p.ldp = m;
        int rc = pg_gemm<T>(ctx, st, GEMM_NN_128_SS, p);
        if (rc) return rc;
        hipLaunchKernelGGL(colreduce_kernel<T>, dim3(m / 256), dim3(256), 0, st, work, (long)m, n / 64, m, q, 0, kss, -1.0, 0);
        LAUNCH_CHECK();
    }
    return 0;
}

template <typename T>
int pg_trmm_lower_t(pg_ctx* ctx, hipStream_t st, int n, int m, const T* M, long ldm, const T* Ks, long ldks, T* V, long ldv) {
    if (n % PG_PAD || m % 128) { pg_set_error("pg_trmm_lower: n_pad=%d m_pad=%d not aligned", n, m); return -2; }
    GemmP<T> p = gp0<T>();
    p.M = n; p.N = m; p.K = n; p.A = M; p.lda = ldm; p.B = Ks; p.ldb = ldks; p.C = V; p.ldc = ldv; p.khi = 1;
    return pg_gemm<T>(ctx, st, GEMM_NN_128, p);
}

// The same two products with the cross-covariance held test-point-major (round 4): Vt[m x n] = Kt Minv^T reads both operands along k
// (the NT form; K range ends with the tile COLUMN), and C_e -= Vt_e Vt_e^T is the NT rank-n update -- for nexp experts in one launch,
// so that the few tiles of an m x m output (136 lower 128 x 128 tiles at m = 2048) fill the chip together.
template <typename T>
int pg_trmm_lower_kt_t(pg_ctx* ctx, hipStream_t st, int n, int m, const T* M, long ldm, long em, const T* Kt, long ldkt, long ekt, T* Vt,
                       long ldvt, long evt, int nexp) {
    if (n % PG_PAD || m % 128) { pg_set_error("pg_trmm_lower_kt: n_pad=%d m_pad=%d not aligned", n, m); return -2; }
    GemmP<T> p = gp0<T>();
    p.M = m; p.N = n; p.K = n; p.A = Kt; p.lda = ldkt; p.B = M; p.ldb = ldm; p.C = Vt; p.ldc = ldvt; p.khi = 2;
    // several experts in one launch: every expert's tiles go longest first, so the short tiles that end one expert run beside the long
    // ones that start the next (one expert alone: 1152 tiles of very different length on 512 slots pack to 0.84 of the ideal)
    p.nexp = nexp; p.eA = ekt; p.eB = em; p.eC = evt;
    return pg_gemm<T>(ctx, st, GEMM_NT_128, p);
}

template <typename T>
int pg_syrk_nt_sub_t(pg_ctx* ctx, hipStream_t st, int m, int n, const T* Vt, long ldvt, long evt, T* C, long ldc, long ec, int nexp,
                     int lower_only) {
    if (n % 16 || m % 128) { pg_set_error("pg_syrk_nt_sub: m_pad=%d n_pad=%d not aligned", m, n); return -2; }
    GemmP<T> p = gp0<T>();
    p.M = p.N = m; p.K = n; p.A = Vt; p.lda = ldvt; p.B = Vt; p.ldb = ldvt; p.C = C; p.ldc = ldc;
    p.alpha = (T)-1; p.beta = (T)1; p.tri = lower_only ? 1 : 0;
    p.nexp = nexp; p.eA = evt; p.eB = evt; p.eC = ec;
    static const int t64 = getenv("PG_SYRK_NT64") ? atoi(getenv("PG_SYRK_NT64")) : 2;   // 2: 128 x 128 tiles below this many slots' worth, 0 / 1: never / always quarter tiles
    const long tm = m / 128, tiles = (lower_only ? tm * (tm + 1) / 2 : tm * tm) * nexp;
    const long slots = ctx && ctx->ncu > 0 ? 2L * ctx->ncu : 512;
    // one expert: quarter tiles while the 128 x 128 tiles do not fill one round of slots (beyond that the mixed launch of pg_gemm ends
    // whole rounds with quarter tiles by itself); several experts (no mixed launch): up to four rounds' worth
    const bool quarter = t64 == 1 || (t64 == 2 && (nexp == 1 ? 4 * tiles < 3 * slots : tiles < 4 * slots));
    return pg_gemm<T>(ctx, st, quarter ? GEMM_NT_64 : GEMM_NT_128, p);
}

template <typename T>
int pg_syrk_tn_sub_t(pg_ctx* ctx, hipStream_t st, int m, int n, const T* V, long ldv, T* C, long ldc, int lower_only) {
    if (n % 16 || m % 128) { pg_set_error("pg_syrk_tn_sub: m_pad=%d n_pad=%d not aligned", m, n); return -2; }
    GemmP<T> p = gp0<T>();
    p.M = p.N = m; p.K = n; p.A = V; p.lda = ldv; p.B = V; p.ldb = ldv; p.C = C; p.ldc = ldc;
    p.alpha = (T)-1; p.beta = (T)1; p.tri = lower_only ? 1 : 0;
    // a few thousand test points: the 128 x 128 tiles of the m x m output do not fill the chip once (m = 2048: 136 lower tiles on 512
    // workgroup slots, 28.7 TFLOP/s) -- quarter tiles then (528 of them)
    static const int t64 = getenv("PG_SYRK_TN64") ? atoi(getenv("PG_SYRK_TN64")) : 1;
    const long tm = m / 128, tiles = lower_only ? tm * (tm + 1) / 2 : tm * tm;
    const bool quarter = t64 && ctx && ctx->ncu > 0 && tiles < 2L * ctx->ncu;
    return pg_gemm<T>(ctx, st, quarter ? GEMM_TN_64 : GEMM_TN_128, p);
}

template <typename T>
int pg_grbcm_terms_t(hipStream_t st, int m, const T* mean_c, const T* var_c, const T* var_g, int is_first, int accumulate,
                     double* out, long ldo, double* beta_out, double* prec_out) {
    hipLaunchKernelGGL(grbcm_terms_batched_kernel<T>, dim3((m + 255) / 256), dim3(256), 0, st, mean_c, 0L, var_c, 0L, var_g, m, 1, is_first ? 0 : -1,
                       accumulate, out, ldo, beta_out, prec_out, 0L);
    LAUNCH_CHECK();
    return 0;
}
template <typename T>
int pg_grbcm_terms_batched_t(hipStream_t st, int m, const T* mean_l, long emean, const T* var_l, long evar, const T* var_g, int nexp, int first,
                             int accumulate, double* out, long ldo, double* beta_out, double* prec_out, long ldb) {
    hipLaunchKernelGGL(grbcm_terms_batched_kernel<T>, dim3((m + 255) / 256), dim3(256), 0, st, mean_l, emean, var_l, evar, var_g, m, nexp, first,
                       accumulate, out, ldo, beta_out, prec_out, ldb);
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
template <typename T>
int pg_weighted_prec_t(hipStream_t st, int m, int m_pad, const T* P, long ldp, const double* beta, T* acc, long lda, int accumulate) {
    const int t = (m_pad + 63) / 64;
    hipLaunchKernelGGL(weighted_prec_kernel<T>, dim3(t, t), dim3(256), 0, st, P, ldp, beta, m, m_pad, acc, lda, accumulate);
    LAUNCH_CHECK();
    return 0;
}
template <typename T> int pg_symmetrize_t(hipStream_t st, int n, T* A, long lda) {
    const int t = (n + 63) / 64;
    hipLaunchKernelGGL(symmetrize_kernel<T>, dim3(t, t), dim3(256), 0, st, A, lda, n);
    LAUNCH_CHECK();
    return 0;
}
template <typename T>
int pg_grbcm_finish_full_t(hipStream_t st, int m, const double* sums, long lds, const T* mean_g, const T* var_g, const T* cov,
                           long ldc, T* mean) {
    hipLaunchKernelGGL(grbcm_finish_full_kernel<T>, dim3((m + 255) / 256), dim3(256), 0, st, sums, lds, mean_g, var_g, cov, ldc, m, mean);
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
    template int pg_potrf_t<T>(pg_ctx*, hipStream_t, int, T*, long, T*, int*, T*, long, const BuildReq<T>*, const ExpBatch*, int, int); \
    template int pg_potrs_vec_t<T>(pg_ctx*, hipStream_t, int, const T*, long, const T*, const T*, T*, T*);             \
    template int pg_trtri_t<T>(pg_ctx*, hipStream_t, int, const T*, long, const T*, T*, long, int, const ExpBatch*);       \
    template int pg_lauum_t<T>(pg_ctx*, hipStream_t, int, const T*, long, T*, long, const ExpBatch*);                  \
    template int pg_potrs_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, const T*, const T*, long, const T*, long, T*, long, T*, int); \
    template int pg_logdet_t<T>(hipStream_t, int, const T*, long, double*);                                   \
    template int pg_trmv_t<T>(pg_ctx*, hipStream_t, int, const T*, long, int, const T*, T*, T*);                       \
    template int pg_alpha_batched_t<T>(hipStream_t, int, const T*, long, long, const T*, long, T*, long, T*, long, T*, long, int, int, double*, long); \
    template int pg_alpha_nlml_async_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, const T*, long, const T*, T*, T*, T*, double*); \
    template int pg_nlml_value_t<T>(hipStream_t, int, const T*, long, const T*, const T*, double*);                    \
    template int pg_predict_mean_q_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, const T*, long, const T*, T*, \
                                        T*, double, T*);                                                               \
    template int pg_predict_mean_q_kt_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, const T*, long, const T*, T*, \
                                           T*, double, T*);                                                            \
    template int pg_predict_mean_q_kt_batched_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, long, const T*, long, long, const T*, long, T*, long, \
                                                   T*, long, const pg_covspec&, const double*, long, T*, long, int);   \
    template int pg_grbcm_terms_batched_t<T>(hipStream_t, int, const T*, long, const T*, long, const T*, int, int, int, double*, long, double*, \
                                             double*, long);                                                           \
    template int pg_trmm_lower_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, const T*, long, T*, long);         \
    template int pg_syrk_tn_sub_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, T*, long, int);                     \
    template int pg_trmm_lower_kt_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, long, const T*, long, long, T*, long, long, int); \
    template int pg_syrk_nt_sub_t<T>(pg_ctx*, hipStream_t, int, int, const T*, long, long, T*, long, long, int, int);  \
    template int pg_grbcm_terms_t<T>(hipStream_t, int, const T*, const T*, const T*, int, int, double*, long, double*, \
                                     double*);                                                                         \
    template int pg_grbcm_finish_t<T>(hipStream_t, int, const double*, long, const T*, const T*, T*, T*, double*,     \
                                      double*);                                                                        \
    template int pg_weighted_prec_t<T>(hipStream_t, int, int, const T*, long, const double*, T*, long, int);            \
    template int pg_symmetrize_t<T>(hipStream_t, int, T*, long);                                                       \
    template int pg_grbcm_finish_full_t<T>(hipStream_t, int, const double*, long, const T*, const T*, const T*, long,  \
                                           T*);                                                                        \
    template int pg_tril_t<T>(hipStream_t, int, T*, long);
INST(double)
INST(float)
