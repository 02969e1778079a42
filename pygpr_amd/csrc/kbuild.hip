// Covariance assembly (HBM-bound, O(n^2 d)) and the fused NLML-gradient contraction.
//
// pg_kbuild: K[i][j] = sum_c k_c(r_i, c_j) (+ (sum sigma_n^2 + jitter) on the diagonal of a symmetric
// build); k_c is the ARD squared exponential of PyGPR/covar.py:129-167 (hp = [sigma, l_1..l_d], l are
// INVERSE length scales, no 1/2 in the exponent) or Matern-5/2 with the same hp layout.  Distances are
// direct sums of squared differences (exactly symmetric, never negative) instead of the reference's
// GEMM expansion (covar.py:102-127).  One 64x64 output tile per 256-thread workgroup; both point tiles
// are staged in LDS k-major ([d][64]); each thread owns a 4x4 micro-tile whose columns are two 16-byte
// vectors, so every store instruction writes 256 contiguous bytes per row.  Rows/columns >= the real
// point count are padding: identity for a symmetric build (keeps the padded Cholesky trivial), zero
// for a cross build.
//
// pg_nlml_grad: g_k = 1/2 sum_ij (K^-1 - a a^T)_ij dK_ij/dtheta_k over the lower triangle, with dK
// recomputed from the point tiles on the fly: dK/dsigma = 2K/sigma, dK/dl_k = -2 l_k D_k^2 K
// (covar.py:169-206), dK/dsigma_n = 2 sigma_n I (covar.py:247-269).  The reference materialises
// dK[nhp,n,n] and solves against it (loss.py:116-121); this is the same number by the K^-1 route.
#include "kbuild.h"
#include "kfun.h"
#include "kmfma.h"
#include <cstdlib>

#define KT 64
#define TLD 65     // odd leading dimension of the LDS transpose tile: column writes are at worst 2-way conflicted

template <typename T> struct VecOf;
template <> struct VecOf<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int N = 2; };
template <> struct VecOf<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };

template <typename T> __device__ __forceinline__ T comp_value(int kind, T sig2, T sqd) {
    if (kind == PG_KIND_RBF) return sig2 * pg_exp(-sqd);
    if (kind == PG_KIND_SQDIST) return sqd;    // Squared_exponential.distance (covar.py:102-127): the scaled squared distance itself
    const T s5 = (T)2.23606797749978969641;
    const T r = sqrt(sqd);
    return sig2 * ((T)1 + s5 * r + (T)(5.0 / 3.0) * sqd) * pg_exp(-s5 * r);
}

template <typename T>
__device__ __forceinline__ void stage_points(T* dst, const T* __restrict__ X, long ldx, int npts, int p0, int d, int tid,
                                             const double* __restrict__ scale = nullptr, double mul = 1.0) {
    // dst[k][64] <- X[p0 + p][k] (* scale[k]: coordinates pre-multiplied by the inverse length scales); points beyond npts read as zero
    for (int idx = tid; idx < KT * d; idx += 256) {
        const int p = idx / d, k = idx % d;
        const int gp = p0 + p;
        T v = (gp < npts) ? X[(long)gp * ldx + k] : (T)0;
        if (scale) v = (T)((double)v * scale[k] * mul);      // (mul = 2: exact)
        dst[k * KT + p] = v;
    }
}

// The 4 x 4 micro-tile of one thread: rows ty*4 + r, columns two (fp64) or one (fp32) 16-byte vectors at v*(16*VE) + tx*VE.
//   PRESC  : one stationary component whose inverse length scales were folded into the staged coordinates (sq += df * df)
//   CHECKED: the tile touches the diagonal, the padding or an accumulate pass -- per-element fix-ups; interior tiles skip them
// The kernel is bound by fp64 VALU issue, not by HBM, at d = 8 (about 250 VALU cycles per 64 elements against the 120 their
// stores take at 5.3 TB/s): both switches only remove instructions.
//   FAST   : (fp64, PRESC, the component is the squared exponential) the reference's own form of the squared distance,
//            |x|^2 + |x'|^2 - 2 x.x' (covar.py:102-127, there a matmul): one FMA per coordinate instead of a subtraction and an
//            FMA.  The rows are staged times two, -|x|^2 per point waits in LDS (nrm_r, nrm_c), so the accumulator STARTS at
//            -|x|^2 - |x'|^2 and ends as the exponential's argument; pg_exp_tab with sigma^2 folded into its table.
template <typename T, bool PRESC, bool CHECKED, bool MIRROR, bool FAST = false>
__device__ __forceinline__ void kb_body(const pg_covspec& spec, const T* xr, const T* xc, const T* l2, const T* sg2, T* tt, int d,
                                        int tr, int tc, int nr, int nc, int symmetric, int accumulate, T* __restrict__ K, long ldk,
                                        int tid, const T* nrm_r = nullptr, const T* nrm_c = nullptr, const double* tab = nullptr) {
    constexpr int VE = VecOf<T>::N, NVC = 4 / VE;   // vectors per row of the micro-tile
    typedef typename VecOf<T>::type vec_t;
    const int tx = tid & 15, ty = tid >> 4;
    T out[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[r][c] = (T)0;
    if (FAST) {
        T arg[4][4];
        {
            T na[4], nb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) na[r] = nrm_r[ty * 4 + r];
#pragma unroll
            for (int v = 0; v < NVC; ++v)
#pragma unroll
                for (int e = 0; e < VE; ++e) nb[v * VE + e] = nrm_c[v * (16 * VE) + tx * VE + e];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) arg[r][c] = na[r] + nb[c];
        }
#pragma unroll 2
        for (int k = 0; k < d; ++k) {
            T a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = xr[k * KT + ty * 4 + r];
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                const vec_t bv = *reinterpret_cast<const vec_t*>(xc + k * KT + v * (16 * VE) + tx * VE);
#pragma unroll
                for (int e = 0; e < VE; ++e) b[v * VE + e] = bv[e];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) arg[r][c] += a[r] * b[c];
        }
        // The expansion's absolute error is eps |x l|^2, not relative to the distance: near-duplicate points can come out with a slightly
        // POSITIVE argument (K_ij > sigma^2) and, in a symmetric build, the diagonal off sigma^2 by that error.  One v_min per element
        // keeps K_ij <= sigma^2; tiles on the diagonal take the exact 0 where a point meets itself (a NaN argument stays NaN).
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                T a = arg[r][c] > (T)0 ? (T)0 : arg[r][c];
                if (CHECKED) {
                    if (symmetric && tr == tc && ty * 4 + r == (c / VE) * (16 * VE) + tx * VE + (c % VE) && a == a) a = (T)0;
                }
                out[r][c] = (T)pg_exp_tab((double)a, tab);
            }
    }
    for (int cp = 0; cp < (FAST ? 0 : spec.ncomp); ++cp) {
        const T* lc = l2 + cp * d;
        T sq[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) sq[r][c] = (T)0;
#pragma unroll 2
        for (int k = 0; k < d; ++k) {      // (two coordinates' LDS reads in flight per trip)
            T a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = xr[k * KT + ty * 4 + r];
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                const vec_t bv = *reinterpret_cast<const vec_t*>(xc + k * KT + v * (16 * VE) + tx * VE);
#pragma unroll
                for (int e = 0; e < VE; ++e) b[v * VE + e] = bv[e];
            }
            if (PRESC) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const T df = a[r] - b[c];
                        sq[r][c] += df * df;
                    }
            } else {
                const T w = lc[k];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const T df = a[r] - b[c];
                        sq[r][c] += w * df * df;
                    }
            }
        }
        // the kind is decided ONCE for the sixteen elements: with the dispatch inside the element loop every covariance value sat
        // behind its own branch and the sixteen exponentials ran one after the other, each a chain of dependent operations
        const int kind = spec.kind[cp];
        const T s2 = sg2[cp];
        if (kind == PG_KIND_RBF) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) out[r][c] += s2 * pg_exp(-sq[r][c]);
        } else if (kind == PG_KIND_SQDIST) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) out[r][c] += sq[r][c];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) out[r][c] += comp_value<T>(PG_KIND_MATERN52, s2, sq[r][c]);
        }
    }
    const T dg = sg2[PG_MAX_COMP];
    const bool mirror = MIRROR && tc < tr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gi = tr * KT + ty * 4 + r;
#pragma unroll
        for (int v = 0; v < NVC; ++v) {
            vec_t vec;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                T val = out[r][v * VE + e];
                if (CHECKED) {
                    const int gj = tc * KT + v * (16 * VE) + tx * VE + e;
                    if (gi >= nr || gj >= nc) val = (symmetric && gi == gj && !accumulate) ? (T)1 : (T)0;
                    else if (symmetric && gi == gj) val += dg;
                }
                vec[e] = val;
                if (MIRROR) { if (mirror) tt[(v * (16 * VE) + tx * VE + e) * TLD + ty * 4 + r] = val; }
            }
            vec_t* dst = reinterpret_cast<vec_t*>(K + (long)gi * ldk + tc * KT + v * (16 * VE) + tx * VE);
            if (CHECKED) { if (accumulate) vec += *dst; }   // a further pass of a Compose with more than PG_MAX_COMP children
            *dst = vec;
        }
    }
    if (MIRROR && mirror) {   // K[tc-tile rows][tr-tile cols] = transpose, read back row-wise so the stores stay 256-byte runs
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int li = ty * 4 + r;
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                vec_t vec;
#pragma unroll
                for (int e = 0; e < VE; ++e) vec[e] = tt[li * TLD + v * (16 * VE) + tx * VE + e];
                vec_t* dst = reinterpret_cast<vec_t*>(K + (long)(tc * KT + li) * ldk + tr * KT + v * (16 * VE) + tx * VE);
                if (CHECKED) { if (accumulate) vec += *dst; }
                *dst = vec;
            }
        }
    }
}

// MIRROR: a symmetric build that also writes the transposed tiles above the diagonal.  NPF: point coordinates a thread stages per
// tile (64 d / 256, rounded up to 2, 4 or 16).
// A workgroup walks a STRIP of up to S tiles of one tile row (round 2: one tile per workgroup): the row's points, the hyper-parameters
// and the launch are paid once per strip, and the next tile's column points are fetched (global -> registers -> the other LDS
// buffer) while the current tile is computed.  A tile's time was 7 us of staging latency, synchronisation and drain around 1 us of
// arithmetic with four workgroups per CU to hide it (rocprof: 2.55 TB/s on the lower-only build); the strip hides it behind work.
template <typename T, bool MIRROR, int NPF, bool FASTK = false>
__global__ __launch_bounds__(256) void pg_kbuild_kernel(pg_covspec spec, const double* __restrict__ hp,
                                                        const T* __restrict__ Xr, long ldr, int nr,
                                                        const T* __restrict__ Xc, long ldc, int nc, int d,
                                                        int symmetric, int accumulate, double jitter,
                                                        T* __restrict__ K, long ldk, int ctile0, int ctile1, int presc, int S,
                                                        long eX, long ehp, long eK, long eXr) {
    // batched experts: blockIdx.y = expert, each with its own points, hyper-parameters and matrix (cross builds: the row points --
    // the test points of a prediction -- have their own stride, 0 when every expert predicts at the same points)
    Xr += blockIdx.y * eXr; Xc += blockIdx.y * eX; hp += blockIdx.y * ehp; K += blockIdx.y * eK;
    int tr, tcs, ntile;
    kb_strip_of(blockIdx.x, symmetric, ctile0, ctile1, S, tr, tcs, ntile);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* xr = reinterpret_cast<T*>(smem_raw);
    T* xc = xr + KT * d;            // two buffers [d][64]
    T* l2 = xc + 2 * KT * d;        // [ncomp][d] squared inverse length scales
    T* sg2 = l2 + PG_MAX_COMP * d;  // [PG_MAX_COMP] sigma^2, then the diagonal term
    T* tt = sg2 + PG_MAX_COMP + 2;  // MIRROR: [64][TLD] transposed tile
    T* nrm = tt + (MIRROR ? KT * TLD : 0);                      // fast path: -|x|^2 of the row points [64], of the column points [2][64]
    double* tab = reinterpret_cast<double*>(nrm + 3 * KT);      // fast path: sigma^2 2^(j/32) [32] (fp64 builds only)
    constexpr bool fast = FASTK;                                // host: one squared-exponential component, fp64, PG_KB_FAST (presc = 2)
    const int tid = threadIdx.x;
    const double* scale = presc ? hp + spec.off[0] + 1 : nullptr;
    // this thread's share of a point tile: elements idx = tid + 256 u < 64 d  ->  point idx / d, coordinate idx % d
    int pp[NPF], kk[NPF];
    double sc[NPF];
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = tid + 256 * u;
        pp[u] = idx < KT * d ? idx / d : -1;
        kk[u] = idx < KT * d ? idx % d : 0;
        sc[u] = (scale && pp[u] >= 0) ? scale[kk[u]] : 1.0;
    }
    // The prefetch must stay a prefetch: the loads are UNCONDITIONAL (clamped address) and nothing touches their result before
    // store_cols.  Written as `valid ? X[..] * scale : 0` each load sat in its own exec-masked block with `s_waitcnt vmcnt(0)` right
    // behind it -- two serial load latencies per tile, each also waiting for the previous tile's stores to drain.
    T pf[NPF];
    bool pok[NPF];
    auto load_cols = [&](int tc) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int gp = tc * KT + pp[u];
            pok[u] = pp[u] >= 0 && gp < nc;
            pf[u] = Xc[(long)min(max(gp, 0), nc - 1) * ldc + kk[u]];
        }
    };
    auto store_cols = [&](T* dst) {
#pragma unroll
        for (int u = 0; u < NPF; ++u)
            if (pp[u] >= 0) dst[kk[u] * KT + pp[u]] = pok[u] ? (T)((double)pf[u] * sc[u]) : (T)0;
    };
    load_cols(tcs);
    stage_points(xr, Xr, ldr, nr, tr * KT, d, tid, scale, fast ? 2.0 : 1.0);
    for (int idx = tid; idx < spec.ncomp * d; idx += 256) {
        const int c = idx / d, k = idx % d;
        const double l = hp[spec.off[c] + 1 + k];
        l2[idx] = (T)(l * l);
    }
    if (tid < spec.ncomp) {
        const double sg = hp[spec.off[tid]];
        sg2[tid] = (T)(sg * sg);
    }
    if (tid == 64) {
        double dg = jitter;
        for (int i = 0; i < spec.nnoise; ++i) { const double s = hp[spec.noise_off[i]]; dg += s * s; }
        sg2[PG_MAX_COMP] = (T)dg;
    }
    store_cols(xc);
    if (fast && tid < 32) {
        const double sg = hp[spec.off[0]];
        tab[tid] = sg * sg * pg_exp2_32[tid];
    }
    __syncthreads();
    // -|x|^2 per staged point, one thread per point (rows: staged times two, hence the quarter)
    auto col_norms = [&](const T* pts, T* dst) {
        if (tid < KT) {
            T sacc = (T)0;
            for (int k = 0; k < d; ++k) { const T v = pts[k * KT + tid]; sacc += v * v; }
            dst[tid] = -sacc;
        }
    };
    if (fast) {
        if (tid >= KT && tid < 2 * KT) {
            T sacc = (T)0;
            for (int k = 0; k < d; ++k) { const T v = xr[k * KT + tid - KT]; sacc += v * v; }
            nrm[tid - KT] = (T)-0.25 * sacc;
        }
        col_norms(xc, nrm + KT);
        __syncthreads();
    }
    // workgroup-uniform: every tile of the strip lies strictly below the diagonal (or the build is a cross build) and inside the
    // real points -- the strip then runs the body without per-element fix-ups.  ONE body per workgroup: with both bodies inlined
    // in the tile loop the kernel needed 160 VGPRs (three workgroups per CU instead of four).
    const int tcl = tcs + ntile - 1;
    const bool interior = !accumulate && (!symmetric || tcl < tr) && (tr + 1) * KT <= nr && (tcl + 1) * KT <= nc;
    auto walk = [&](auto body) {
        for (int t = 0; t < ntile; ++t) {
            const int tc = tcs + t;
            const T* cur = xc + (t & 1) * KT * d;
            if (t + 1 < ntile) load_cols(tc + 1);           // in flight while this tile is computed
            body(cur, tc);
            if (t + 1 < ntile) {
                store_cols(xc + ((t + 1) & 1) * KT * d);
                __syncthreads();   // publishes the next tile's points; also orders this tile's reads of `tt` before the next one's writes
                if (fast) {
                    col_norms(xc + ((t + 1) & 1) * KT * d, nrm + KT + ((t + 1) & 1) * KT);
                    __syncthreads();
                }
            }
        }
    };
#define KB_ARGS(cur, tc) spec, xr, cur, l2, sg2, tt, d, tr, tc, nr, nc, symmetric, accumulate, K, ldk, tid
    if constexpr (FASTK) {
#define KB_FARGS(cur, tc) KB_ARGS(cur, tc), nrm, nrm + KT + (int)((cur - xc) / (KT * d)) * KT, tab
        if (interior) walk([&](const T* cur, int tc) { kb_body<T, true, false, MIRROR, true>(KB_FARGS(cur, tc)); });
        else walk([&](const T* cur, int tc) { kb_body<T, true, true, MIRROR, true>(KB_FARGS(cur, tc)); });
#undef KB_FARGS
    } else {
        if (presc) {
            if (interior) walk([&](const T* cur, int tc) { kb_body<T, true, false, MIRROR>(KB_ARGS(cur, tc)); });
            else walk([&](const T* cur, int tc) { kb_body<T, true, true, MIRROR>(KB_ARGS(cur, tc)); });
        } else {
            if (interior) walk([&](const T* cur, int tc) { kb_body<T, false, false, MIRROR>(KB_ARGS(cur, tc)); });
            else walk([&](const T* cur, int tc) { kb_body<T, false, true, MIRROR>(KB_ARGS(cur, tc)); });
        }
    }
#undef KB_ARGS
}

template <typename T>
int pg_kbuild(hipStream_t st, const pg_covspec& spec, const double* hp, const T* Xr, long ldr, int nr,
              const T* Xc, long ldc, int nc, int d, int symmetric, int lower_only, int accumulate, double jitter, T* K,
              long ldk, int rows_pad, int cols_pad, int col0, int col1, int nexp, long eX, long ehp, long eK, long eXr) {
    if (eXr < 0 || symmetric) eXr = eX;      // symmetric builds: one point set per expert
    if (rows_pad % KT || cols_pad % KT || d < 1 || d > PG_MAX_DIM) {
        pg_set_error("pg_kbuild: bad shape rows_pad=%d cols_pad=%d d=%d", rows_pad, cols_pad, d);
        return -2;
    }
    const bool mirror = symmetric && !lower_only;
    const size_t lds = (size_t)(3 * KT * d + PG_MAX_COMP * d + PG_MAX_COMP + 2 + (mirror ? KT * TLD : 0) + 3 * KT) * sizeof(T) + 32 * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {   // large d passes the 64 KB a kernel gets without opting in (134 KB for the mirrored fp64 build at d = 64)
        const size_t lds_max = (size_t)(3 * KT * PG_MAX_DIM + PG_MAX_COMP * PG_MAX_DIM + PG_MAX_COMP + 2 + KT * TLD + 3 * KT) * sizeof(T) + 32 * sizeof(double);
        const void* fns[6] = {reinterpret_cast<const void*>(pg_kbuild_kernel<T, true, 2>), reinterpret_cast<const void*>(pg_kbuild_kernel<T, true, 4>),
                              reinterpret_cast<const void*>(pg_kbuild_kernel<T, true, 16>), reinterpret_cast<const void*>(pg_kbuild_kernel<T, false, 2>),
                              reinterpret_cast<const void*>(pg_kbuild_kernel<T, false, 4>), reinterpret_cast<const void*>(pg_kbuild_kernel<T, false, 16>)};
        for (const void* f : fns) PG_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        attr_done = true;
    }
    // columns [col0, col1) only (col1 <= 0: all): a lower-only symmetric build in two column windows lets the factorisation
    // start on the first panel while the rest is still being written (pg_potrf_t, BuildReq)
    if (col1 <= 0) { col0 = 0; col1 = cols_pad; }
    if (col0 % KT || col1 % KT || col0 < 0 || col1 > cols_pad || col0 >= col1) {
        pg_set_error("pg_kbuild: bad column window [%d, %d)", col0, col1);
        return -2;
    }
    if (mirror && (col0 != 0 || col1 != cols_pad)) {
        pg_set_error("pg_kbuild: a mirrored build cannot be windowed");
        return -2;
    }
    const int c0 = col0 / KT, c1 = col1 / KT, W = c1 - c0, TR = rows_pad / KT;
    // strips of up to S tiles of one tile row per workgroup (PG_KB_STRIP; 1 = one tile per workgroup, round 2's granularity)
    // Re-swept with the fast body and the true prefetch (N = 16384, d = 8, lower-only / mirrored, ms): S = 1: 0.326 / 0.561, 2: 0.265 / 0.477,
    // 3: 0.236 / 0.411, 4: 0.224 / 0.406, 5: 0.214 / 0.394, 6: 0.213 / 0.397, 8: 0.233 / 0.424, 12: 0.224 / 0.425, 16: 0.239 / 0.429,
    // 32: 0.259 / 0.459 -- strips whose rows are NOT a multiple of 4 KB long do best (concurrent workgroups spread over the channels)
    static const int strip_env = getenv("PG_KB_STRIP") ? atoi(getenv("PG_KB_STRIP")) : 6;
    const int S = std::max(1, std::min(strip_env, 64));
    const int SW = (W + S - 1) / S;
    // symmetric: the triangle of the window's own tile rows plus the rectangle below it; cross build: every tile of the window
    const long strips = symmetric ? kb_strips_before(W, S) + (long)(TR - c1) * SW : (long)TR * SW;
    if (strips <= 0) return 0;
    // one stationary component (the common Compose([SE, WN])): its inverse length scales go into the staged coordinates.  In fp64
    // (x l) - (x' l) rounds differently from l^2 (x - x')^2 in the last bit; PG_KB_PRESC=0 keeps the unscaled form
    static const int presc_env = getenv("PG_KB_PRESC") ? atoi(getenv("PG_KB_PRESC")) : 1;
    // ... and when that component is the squared exponential, fp64 builds take the fast body (kb_body, FAST): presc = 2
    static const int fast_env = getenv("PG_KB_FAST") ? atoi(getenv("PG_KB_FAST")) : 1;
    const int presc = (presc_env && spec.ncomp == 1) ? ((fast_env && sizeof(T) == 8 && spec.kind[0] == PG_KIND_RBF && !accumulate) ? 2 : 1) : 0;
    // One stationary component, d <= 16, no accumulate pass: the distance on the matrix pipe (kmfma.hip).  PG_KB_MFMA = 0: never;
    // 1 (default): wherever the VALU bodies have no fast form -- d > 8, Matern-5/2, fp32; 2: also for the fp64 squared exponential at
    // d <= 8, which the fast body of round 3 serves at 0.63-0.66 of the HBM peak.
    const int mfma_env = getenv("PG_KB_MFMA") ? atoi(getenv("PG_KB_MFMA")) : 1;      // (read per call: tests compare the bodies in one process)
    // (mirrored, lower-only and cross builds of one (kind, dtype, d) all take the same body: their values agree bit for bit)
    if (mfma_env && spec.ncomp == 1 && !accumulate && d <= 16 && (spec.kind[0] == PG_KIND_RBF || spec.kind[0] == PG_KIND_MATERN52) &&
        (mfma_env >= 2 || presc != 2 || d > 8))
        return pg_kbuild_mfma<T>(st, spec, hp, Xr, ldr, nr, Xc, ldc, nc, d, symmetric, mirror ? 1 : 0, jitter, K, ldk, c0, c1, S, strips, nexp, eX,
                                 ehp, eK, eXr);
    const int npf = d <= 8 ? 2 : (d <= 16 ? 4 : 16);
#define KB_LAUNCH(M, P, F)                                                                                                           \
    hipLaunchKernelGGL((pg_kbuild_kernel<T, M, P, F>), dim3((unsigned)strips, (unsigned)nexp), dim3(256), lds, st, spec, hp, Xr, ldr, nr, Xc, \
                       ldc, nc, d, symmetric, accumulate, jitter, K, ldk, c0, c1, presc, S, eX, ehp, eK, eXr)
    bool launched = false;
    if constexpr (sizeof(T) == 8) {
        if (presc == 2) {
            static bool fattr = false;
            if (!fattr) {
                const size_t lds_max = (size_t)(3 * KT * PG_MAX_DIM + PG_MAX_COMP * PG_MAX_DIM + PG_MAX_COMP + 2 + KT * TLD + 3 * KT) * sizeof(T) + 32 * sizeof(double);
                const void* fns[6] = {reinterpret_cast<const void*>(pg_kbuild_kernel<T, true, 2, true>), reinterpret_cast<const void*>(pg_kbuild_kernel<T, true, 4, true>),
                                      reinterpret_cast<const void*>(pg_kbuild_kernel<T, true, 16, true>), reinterpret_cast<const void*>(pg_kbuild_kernel<T, false, 2, true>),
                                      reinterpret_cast<const void*>(pg_kbuild_kernel<T, false, 4, true>), reinterpret_cast<const void*>(pg_kbuild_kernel<T, false, 16, true>)};
                for (const void* f : fns) PG_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
                fattr = true;
            }
            if (mirror) { if (npf == 2) KB_LAUNCH(true, 2, true); else if (npf == 4) KB_LAUNCH(true, 4, true); else KB_LAUNCH(true, 16, true); }
            else { if (npf == 2) KB_LAUNCH(false, 2, true); else if (npf == 4) KB_LAUNCH(false, 4, true); else KB_LAUNCH(false, 16, true); }
            launched = true;
        }
    }
    if (!launched) {
        if (mirror) { if (npf == 2) KB_LAUNCH(true, 2, false); else if (npf == 4) KB_LAUNCH(true, 4, false); else KB_LAUNCH(true, 16, false); }
        else { if (npf == 2) KB_LAUNCH(false, 2, false); else if (npf == 4) KB_LAUNCH(false, 4, false); else KB_LAUNCH(false, 16, false); }
    }
#undef KB_LAUNCH
    PG_CHECK(hipGetLastError());
    return 0;
}
template int pg_kbuild<double>(hipStream_t, const pg_covspec&, const double*, const double*, long, int,
                               const double*, long, int, int, int, int, int, double, double*, long, int, int, int, int, int, long, long, long, long);
template int pg_kbuild<float>(hipStream_t, const pg_covspec&, const double*, const float*, long, int,
                              const float*, long, int, int, int, int, int, double, float*, long, int, int, int, int, int, long, long, long, long);

// ------------------------------------------------------------------------------------------------
// dK stack of the public Covar.kernel_and_grad (covar.py:64-81,169-206,247-269): dK[p][i][j] for
// every hyper-parameter p.  Only the drop-in surface needs it; the NLML path never materialises it.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pg_kgrad_kernel(pg_covspec spec, const double* __restrict__ hp,
                                                       const T* __restrict__ X, long ldx, int n, int d,
                                                       T* __restrict__ dK, long slab) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= n || j >= n) return;
    const long e = (long)i * n + j;
    const T* xi = X + (long)i * ldx;
    const T* xj = X + (long)j * ldx;
    for (int cp = 0; cp < spec.ncomp; ++cp) {
        const int o = spec.off[cp];
        const double sg = hp[o];
        T sq = (T)0;
        for (int k = 0; k < d; ++k) {
            const double l = hp[o + 1 + k];
            const T df = xi[k] - xj[k];
            sq += (T)(l * l) * df * df;
        }
        T kv, base, coef;
        if (spec.kind[cp] == PG_KIND_RBF) {
            kv = (T)(sg * sg) * pg_exp(-sq);
            base = kv;
            coef = (T)-2;
        } else {
            const T s5 = (T)2.23606797749978969641;
            const T rr = sqrt(sq), ex = pg_exp(-s5 * rr);
            kv = (T)(sg * sg) * ((T)1 + s5 * rr + (T)(5.0 / 3.0) * sq) * ex;
            base = (T)(sg * sg) * ((T)1 + s5 * rr) * ex;
            coef = (T)(-5.0 / 3.0);
        }
        dK[(long)o * slab + e] = kv * (T)(2.0 / sg);
        for (int k = 0; k < d; ++k) {
            const T df = xi[k] - xj[k];
            dK[(long)(o + 1 + k) * slab + e] = coef * (T)hp[o + 1 + k] * df * df * base;
        }
    }
    for (int q = 0; q < spec.nnoise; ++q)
        dK[(long)spec.noise_off[q] * slab + e] = (i == j) ? (T)(2.0 * hp[spec.noise_off[q]]) : (T)0;
}

template <typename T>
int pg_kgrad(hipStream_t st, const pg_covspec& spec, const double* hp, const T* X, long ldx, int n, int d, T* dK) {
    if (n <= 0 || d < 1 || d > PG_MAX_DIM) { pg_set_error("pg_kernel_grad_build: bad shape n=%d d=%d", n, d); return -2; }
    hipLaunchKernelGGL(pg_kgrad_kernel<T>, dim3((n + 63) / 64, (n + 3) / 4), dim3(256), 0, st, spec, hp, X, ldx, n, d, dK,
                       (long)n * n);
    PG_CHECK(hipGetLastError());
    return 0;
}
template int pg_kgrad<double>(hipStream_t, const pg_covspec&, const double*, const double*, long, int, int, double*);
template int pg_kgrad<float>(hipStream_t, const pg_covspec&, const double*, const float*, long, int, int, float*);

// ------------------------------------------------------------------------------------------------
// expert partitioning (sampler.py:68-119): squared distances to a small set of centres and the nearest centre.
// One thread per point, centres staged in LDS; direct differences (the reference expands into a GEMM, sampler.py:94-100).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pg_centres_kernel(const T* __restrict__ X, long ldx, int n, const T* __restrict__ Cn,
                                                         long ldc, int m, int d, T* __restrict__ D, long ldd,
                                                         int* __restrict__ idx, int chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* cs = reinterpret_cast<T*>(smem_raw);          // [chunk][d]
    const int i = blockIdx.x * 256 + threadIdx.x;
    T best = (T)0;
    int arg = 0;
    for (int j0 = 0; j0 < m; j0 += chunk) {
        const int mc = min(chunk, m - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < mc * d; e += 256) cs[e] = Cn[(long)(j0 + e / d) * ldc + e % d];
        __syncthreads();
        if (i < n) {
            for (int j = 0; j < mc; ++j) {
                T s = (T)0;
                for (int k = 0; k < d; ++k) {
                    const T df = X[(long)i * ldx + k] - cs[j * d + k];
                    s += df * df;
                }
                if (D) D[(long)i * ldd + j0 + j] = s;
                if ((j0 + j == 0) || s < best) { best = s; arg = j0 + j; }
            }
        }
    }
    if (idx && i < n) idx[i] = arg;
}

template <typename T>
int pg_centres(hipStream_t st, const T* X, long ldx, int n, const T* Cn, long ldc, int m, int d, T* D, long ldd, int* idx) {
    if (n <= 0 || m <= 0 || d < 1 || d > PG_MAX_DIM) { pg_set_error("pg_sqdist: bad shape n=%d m=%d d=%d", n, m, d); return -2; }
    // centres are staged in chunks that fit the 64 KB of LDS a kernel gets without opting in, whatever d is
    const int chunk = std::max(1, std::min(std::min(m, 1024), (int)(48 * 1024 / (d * sizeof(T)))));
    const size_t lds = (size_t)chunk * d * sizeof(T);
    hipLaunchKernelGGL(pg_centres_kernel<T>, dim3((n + 255) / 256), dim3(256), lds, st, X, ldx, n, Cn, ldc, m, d, D, ldd, idx,
                       chunk);
    PG_CHECK(hipGetLastError());
    return 0;
}
template int pg_centres<double>(hipStream_t, const double*, long, int, const double*, long, int, int, double*, long, int*);
template int pg_centres<float>(hipStream_t, const float*, long, int, const float*, long, int, int, float*, long, int*);

// ------------------------------------------------------------------------------------------------
// fused gradient contraction
// ------------------------------------------------------------------------------------------------
// One workgroup walks up to GCH column tiles of its tile row with the accumulators in registers: the row's point tile, the
// cross-wave reduction and the partial-sum row are paid once per strip instead of once per 64 x 64 tile.  Measured at
// N = 16384, D = 8 (contraction + reduce): one tile per workgroup 873 + 120 us, strips of 4 826 + 33 us, strips of 16
// 1061 + 11 us (imbalance); keeping the column point and squared differences in registers (138 VGPRs) 1179 us.
#define GCH 4
template <typename T, int DMAX>
__global__ __launch_bounds__(256) void pg_grad_kernel(pg_covspec spec, const double* __restrict__ hp,
                                                      const T* __restrict__ X, long ldx, int n, int d,
                                                      const T* __restrict__ Kinv, long ldk,
                                                      const T* __restrict__ alpha, double* __restrict__ part,
                                                      int nhp, GradBatch gb) {
    // batched experts: blockIdx.z = expert, each with its own points, hyper-parameters, K^-1, weights and partial sums
    X += blockIdx.z * gb.eX; hp += blockIdx.z * gb.ehp; Kinv += blockIdx.z * gb.eK; alpha += blockIdx.z * gb.ea; part += blockIdx.z * gb.epart;
    const int tr = blockIdx.y;
    const int c0 = blockIdx.x * GCH, c1 = min(c0 + GCH, tr + 1);   // column tiles [c0, c1), none above the diagonal
    const int blk = tr * gridDim.x + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (c0 > tr) {   // strip entirely above the diagonal: its partial row must still be zero
        for (int idx = tid; idx < nhp; idx += 256) part[(long)blk * nhp + idx] = 0.0;
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* red = reinterpret_cast<double*>(smem_raw);             // [4 waves][DMAX + 2]
    T* xr = reinterpret_cast<T*>(red + 4 * (DMAX + 2));             // [DMAX][64], zero for k >= d
    T* xc = xr + KT * DMAX;                                         // two buffers [DMAX][64]
    T* l2 = xc + 2 * KT * DMAX;                                     // [ncomp][DMAX], zero for k >= d
    for (int idx = tid; idx < KT * DMAX; idx += 256) {
        const int p = idx / DMAX, k = idx % DMAX;
        const int gr = tr * KT + p;
        xr[k * KT + p] = (k < d && gr < n) ? X[(long)gr * ldx + k] : (T)0;
    }
    for (int idx = tid; idx < spec.ncomp * DMAX; idx += 256) {
        const int c = idx / DMAX, k = idx % DMAX;
        const double l = (k < d) ? hp[spec.off[c] + 1 + k] : 0.0;
        l2[idx] = (T)(l * l);
    }
    for (int idx = tid; idx < nhp; idx += 256) part[(long)blk * nhp + idx] = 0.0;

    // element e of this thread: row = 4 e + wave (one wave reads one 64-wide row: 512 contiguous bytes),
    // column = lane.  W = weight * (Kinv - a a^T): 2 below the diagonal, 1 on it, 0 above / in padding.
    double tr_w = 0.0;
    int it = 0;   // running tile counter: selects the point-tile buffer
    for (int cp = 0; cp < max(spec.ncomp, 1); ++cp) {
        const bool have = cp < spec.ncomp;      // ncomp == 0 (pure white noise): only the trace of W is needed
        const int o = have ? spec.off[cp] : 0;
        const double sg = have ? hp[o] : 0.0;
        const T sig2 = (T)(sg * sg);
        const T* lc = l2 + cp * DMAX;
        const int kind = have ? spec.kind[cp] : PG_KIND_RBF;
        double acc[DMAX + 1];
#pragma unroll
        for (int k = 0; k <= DMAX; ++k) acc[k] = 0.0;
        for (int tc = c0; tc < c1; ++tc, ++it) {
            T* xb = xc + (it & 1) * KT * DMAX;
            for (int idx = tid; idx < KT * DMAX; idx += 256) {
                const int p = idx / DMAX, k = idx % DMAX;
                const int gc = tc * KT + p;
                xb[k * KT + p] = (k < d && gc < n) ? X[(long)gc * ldx + k] : (T)0;
            }
            __syncthreads();   // also orders this buffer's previous readers (two tiles ago) before the writes above
            const int gj = tc * KT + lane;
            const double aj = (gj < n) ? (double)alpha[gj] : 0.0;
#pragma unroll 2
            for (int e = 0; e < 16; ++e) {
                const int row = 4 * e + wave;
                const int gi = tr * KT + row;
                double w = 0.0;
                if (gi < n && gj <= gi) {
                    w = (double)Kinv[(long)gi * ldk + gj] - (double)alpha[gi] * aj;
                    if (gj < gi) w *= 2.0; else if (cp == 0) tr_w += w;
                }
                if (!have) continue;
                T sq = (T)0;
#pragma unroll
                for (int k = 0; k < DMAX; ++k) {
                    const T df = xr[k * KT + row] - xb[k * KT + lane];
                    sq += lc[k] * df * df;
                }
                double kv, base;   // dK/dl_k = base * l_k * D_k^2 (sign and constants applied in the reduce)
                if (kind == PG_KIND_RBF) {
                    kv = (double)(sig2 * pg_exp(-sq));
                    base = kv;
                } else {
                    const T s5 = (T)2.23606797749978969641;
                    const T rr = sqrt(sq), ex = pg_exp(-s5 * rr);
                    kv = (double)(sig2 * ((T)1 + s5 * rr + (T)(5.0 / 3.0) * sq) * ex);
                    base = (double)(sig2 * ((T)1 + s5 * rr) * ex);
                }
                acc[0] += w * kv;
                const double wb = w * base;
#pragma unroll
                for (int k = 0; k < DMAX; ++k) {
                    const double df = (double)(xr[k * KT + row] - xb[k * KT + lane]);
                    acc[1 + k] += wb * df * df;
                }
            }
        }
        if (have) {
#pragma unroll
            for (int k = 0; k <= DMAX; ++k) {
                const double s = wave_sum(acc[k]);
                if (lane == 0) red[wave * (DMAX + 2) + k] = s;
            }
            __syncthreads();
            if (tid <= d)
                part[(long)blk * nhp + o + tid] =
                    red[tid] + red[(DMAX + 2) + tid] + red[2 * (DMAX + 2) + tid] + red[3 * (DMAX + 2) + tid];
            __syncthreads();
        }
    }
    {
        const double s = wave_sum(tr_w);
        if (lane == 0) red[wave * (DMAX + 2)] = s;
        __syncthreads();
        if (tid < spec.nnoise)
            part[(long)blk * nhp + spec.noise_off[tid]] =
                red[0] + red[DMAX + 2] + red[2 * (DMAX + 2)] + red[3 * (DMAX + 2)];
    }
}

// The contraction's fast body (round 4): fp64, ONE squared-exponential child (the common Compose([SE, WN])) and d <= 8 -- what round 3
// did for the covariance build (kb_body<FAST>).  The general kernel above walks a thread's sixteen elements two at a time, each behind
// its own degree-13 exponential chain: 0.82 ms at N = 16384, D = 8 for 1.07 GB of K^-1 (1.3 TB/s), a third of the fp64 issue rate.  Here
//   * a thread owns a 4 x 4 micro-tile (rows ty*4 + r, columns v*32 + tx*2 + e: 16-byte loads of K^-1, 256-byte runs per row) whose
//     sixteen elements are INDEPENDENT chains: distances, exponentials and weights interleave;
//   * the inverse length scales are folded into the staged coordinates (one subtract + one FMA per coordinate; the partial sums of the
//     length-scale entries then hold l_k^2 D_k^2 and the reduce divides by l_k: `presc`), differences stay DIRECT -- the per-coordinate
//     squares are needed anyway, and they keep the weights of near-duplicate points exact;
//   * the exponential is pg_exp_tab with sigma^2 folded into its table;
//   * tiles strictly below the diagonal and inside the real points take a body without per-element tests (weight 2 everywhere).
template <int DMAX>
__global__ __launch_bounds__(256, 3) void pg_grad_fast_kernel(pg_covspec spec, const double* __restrict__ hp, const double* __restrict__ X,
                                                           long ldx, int n, int d, const double* __restrict__ Kinv, long ldk,
                                                           const double* __restrict__ alpha, double* __restrict__ part, int nhp, GradBatch gb) {
    typedef double vec_t __attribute__((ext_vector_type(2)));
    X += blockIdx.z * gb.eX; hp += blockIdx.z * gb.ehp; Kinv += blockIdx.z * gb.eK; alpha += blockIdx.z * gb.ea; part += blockIdx.z * gb.epart;
    const int tr = blockIdx.y;
    const int c0 = blockIdx.x * GCH, c1 = min(c0 + GCH, tr + 1);
    const int blk = tr * gridDim.x + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < nhp; idx += 256) part[(long)blk * nhp + idx] = 0.0;
    if (c0 > tr) return;
    __shared__ double red[4][DMAX + 2];
    __shared__ double xr[DMAX * KT], xc[2][DMAX * KT], ar[KT], ac[2][KT], tab[32];
    const int o = spec.off[0];
    const double* scale = hp + o + 1;
    auto stage = [&](double* dst, double* adst, int t0) {
        for (int idx = tid; idx < KT * DMAX; idx += 256) {
            const int p = idx / DMAX, k = idx % DMAX;
            const int g = t0 * KT + p;
            dst[k * KT + p] = (k < d && g < n) ? X[(long)g * ldx + k] * scale[k] : 0.0;
        }
        if (tid < KT) adst[tid] = (t0 * KT + tid < n) ? alpha[t0 * KT + tid] : 0.0;
    };
    stage(xr, ar, tr);
    if (tid < 32) { const double sg = hp[o]; tab[tid] = sg * sg * pg_exp2_32[tid]; }
    const int tx = tid & 15, ty = tid >> 4;
    double acc[DMAX + 1];
#pragma unroll
    for (int k = 0; k <= DMAX; ++k) acc[k] = 0.0;
    double tr_w = 0.0;
    for (int tc = c0, it = 0; tc < c1; ++tc, ++it) {
        double* xb = xc[it & 1];
        double* ab = ac[it & 1];
        stage(xb, ab, tc);
        const bool interior = tc < tr && (tr + 1) * KT <= n;
        __syncthreads();   // publishes the staged tile; also orders this buffer's previous readers (two tiles ago) before the writes above
        // The micro-tile in two halves of 4 rows x 2 columns (eight independent chains each).  All sixteen elements at once, or halves
        // whose coordinates the compiler keeps in registers across both passes, need 170-270 VGPRs (one wave per SIMD): the column
        // points of a half are held on purpose (2 DMAX doubles), the row points are re-read from LDS in the second pass.
#pragma unroll 1
        for (int v = 0; v < 2; ++v) {
            const int cb = v * 32 + tx * 2;
            const int gj0 = tc * KT + cb;
            vec_t kin[4];     // this half's eight K^-1 values: in flight while the distances are summed (the row's stride ldk is even and covers the padding)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                kin[r] = *reinterpret_cast<const vec_t*>(Kinv + (long)min(tr * KT + ty * 4 + r, n - 1) * ldk + min(gj0, (int)ldk - 2));
            vec_t b[DMAX];
#pragma unroll
            for (int k = 0; k < DMAX; ++k) b[k] = *reinterpret_cast<const vec_t*>(xb + k * KT + cb);
            const vec_t aj = *reinterpret_cast<const vec_t*>(ab + cb);
            double sq[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r) sq[r][0] = sq[r][1] = 0.0;
#pragma unroll
            for (int k = 0; k < DMAX; ++k) {
                const vec_t a01 = *reinterpret_cast<const vec_t*>(xr + k * KT + ty * 4), a23 = *reinterpret_cast<const vec_t*>(xr + k * KT + ty * 4 + 2);
                const double a[4] = {a01[0], a01[1], a23[0], a23[1]};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const double df = a[r] - b[k][c];
                        sq[r][c] = __builtin_fma(df, df, sq[r][c]);
                    }
            }
            // W = weight * (K^-1 - a a^T), times the covariance value
            double wb[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ai = ar[ty * 4 + r];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    double w = __builtin_fma(-ai, aj[c], kin[r][c]);
                    if (interior) w *= 2.0;
                    else {
                        const int gi = tr * KT + ty * 4 + r, gj = gj0 + c;
                        if (gi >= n || gj > gi) w = 0.0;
                        else if (gj < gi) w *= 2.0;
                        else tr_w += w;
                    }
                    wb[r][c] = w * pg_exp_tab(-sq[r][c], tab);
                    acc[0] += wb[r][c];
                }
            }
            asm volatile("" ::: "memory");    // second pass: re-read the row points instead of keeping 4 DMAX doubles alive
#pragma unroll
            for (int k = 0; k < DMAX; ++k) {
                const vec_t a01 = *reinterpret_cast<const vec_t*>(xr + k * KT + ty * 4), a23 = *reinterpret_cast<const vec_t*>(xr + k * KT + ty * 4 + 2);
                const double a[4] = {a01[0], a01[1], a23[0], a23[1]};
                double s = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const double df = a[r] - b[k][c];
                        s = __builtin_fma(wb[r][c], df * df, s);
                    }
                acc[1 + k] += s;
            }
        }
    }
#pragma unroll
    for (int k = 0; k <= DMAX; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    {
        const double s = wave_sum(tr_w);
        if (lane == 0) red[wave][DMAX + 1] = s;
    }
    __syncthreads();
    if (tid <= d) part[(long)blk * nhp + o + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    if (tid < spec.nnoise)
        part[(long)blk * nhp + spec.noise_off[tid]] = red[0][DMAX + 1] + red[1][DMAX + 1] + red[2][DMAX + 1] + red[3][DMAX + 1];
}

// grad[p] = scale_p * sum_blocks part[b][p]
__global__ __launch_bounds__(256) void pg_grad_reduce_kernel(pg_covspec spec, const double* __restrict__ hp,
                                                             const double* __restrict__ part, int nblk, int nhp,
                                                             int d, double* __restrict__ grad, int presc, GradBatch gb) {
    __shared__ double red[4];
    hp += blockIdx.z * gb.ehp; part += blockIdx.z * gb.epart; grad += blockIdx.z * gb.egrad;
    const int p = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int b = tid; b < nblk; b += 256) s += part[(long)b * nhp + p];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        s = red[0] + red[1] + red[2] + red[3];
        double scale = 0.0;
        bool mine = false;   // entries of children that are not in this spec belong to another pass of a long Compose
        for (int c = 0; c < spec.ncomp; ++c) {
            const int o = spec.off[c];
            if (p == o) { scale = 0.5 * 2.0 / hp[o]; mine = true; }       // dK/dsigma = 2K/sigma
            else if (p > o && p <= o + d) {
                scale = (spec.kind[c] == PG_KIND_RBF) ? 0.5 * -2.0 * hp[p]   // -2 l_k D_k^2 K
                                                      : 0.5 * -(5.0 / 3.0) * hp[p];
                // the fast contraction summed (l_k D_k)^2: -l_k S = -S' / l_k (l_k = 0: S' = 0 and the derivative is 0)
                if (presc) scale = hp[p] != 0.0 ? ((spec.kind[c] == PG_KIND_RBF) ? -1.0 : -5.0 / 6.0) / hp[p] : 0.0;
                mine = true;
            }
        }
        for (int i = 0; i < spec.nnoise; ++i)
            if (p == spec.noise_off[i]) { scale = 0.5 * 2.0 * hp[p]; mine = true; }   // dK/dsigma_n = 2 sigma_n I
        if (mine) grad[p] = scale * s;
    }
}

template <typename T, int DMAX>
static int launch_grad(hipStream_t st, const pg_covspec& spec, const double* hp, const T* X, long ldx, int n,
                       int d, const T* Kinv, long ldk, const T* alpha, double* part, int nhp, int tiles, const GradBatch& gb, int nexp) {
    const size_t lds = (size_t)(3 * KT * DMAX + PG_MAX_COMP * DMAX) * sizeof(T) + 4 * (DMAX + 2) * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {   // d > 32 needs more than the 64 KB a kernel gets without opting in
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_grad_kernel<T, DMAX>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((pg_grad_kernel<T, DMAX>), dim3((tiles + GCH - 1) / GCH, tiles, nexp), dim3(256), lds, st, spec, hp, X, ldx,
                       n, d, Kinv, ldk, alpha, part, nhp, gb);
    PG_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
int pg_nlml_grad_t(hipStream_t st, const pg_covspec& spec, const double* hp, const T* X, long ldx, int n, int d,
                   const T* Kinv, long ldk, const T* alpha, double* grad, int nhp, double* work, long lwork,
                   int nexp, long ehp, long eX, long eK, long ea, long egrad) {
    const int tiles = (n + KT - 1) / KT;
    const long need = (long)tiles * tiles * nhp;
    if (nexp < 1 || nexp > 65535) { pg_set_error("pg_nlml_grad: 1 <= nexp <= 65535"); return -2; }
    if (lwork < need * nexp) { pg_set_error("pg_nlml_grad: workspace %ld < %ld doubles", lwork, need * nexp); return -3; }
    const GradBatch gb = {eX, ehp, eK, ea, need, egrad};
    int rc;
    static const int fast_env = getenv("PG_GRAD_FAST") ? atoi(getenv("PG_GRAD_FAST")) : 1;
    // One stationary component, d <= 16: the contraction on the matrix pipe (kmfma.hip; PG_GRAD_MFMA=0 restores the VALU kernels)
    const int mfma_env = getenv("PG_GRAD_MFMA") ? atoi(getenv("PG_GRAD_MFMA")) : 1;   // (read per call: tests compare the bodies in one process)
    if (mfma_env && spec.ncomp == 1 && d <= 16 && n >= 1 && (spec.kind[0] == PG_KIND_RBF || spec.kind[0] == PG_KIND_MATERN52)) {
        int nblk = 0;
        if ((rc = pg_grad_mfma<T>(st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, work, nhp, tiles, gb, nexp, &nblk))) return rc;
        hipLaunchKernelGGL(pg_grad_reduce_kernel, dim3(nhp, 1, nexp), dim3(256), 0, st, spec, hp, work, nblk, nhp, d, grad, 1, gb);
        PG_CHECK(hipGetLastError());
        return 0;
    }
    int presc = 0;
    if constexpr (sizeof(T) == 8) {
        // (d <= 8: with sixteen coordinates the held column points alone are 64 VGPRs and the kernel spills)
        if (fast_env && spec.ncomp == 1 && spec.kind[0] == PG_KIND_RBF && d <= 8 && n >= 2 && ldk % 2 == 0) {
            const dim3 grid((tiles + GCH - 1) / GCH, tiles, nexp);
            if (d <= 4) hipLaunchKernelGGL(pg_grad_fast_kernel<4>, grid, dim3(256), 0, st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, work, nhp, gb);
            else hipLaunchKernelGGL(pg_grad_fast_kernel<8>, grid, dim3(256), 0, st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, work, nhp, gb);
            PG_CHECK(hipGetLastError());
            presc = 1;
        }
    }
    if (presc) rc = 0;
    else if (d <= 4) rc = launch_grad<T, 4>(st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, work, nhp, tiles, gb, nexp);
    else if (d <= 8) rc = launch_grad<T, 8>(st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, work, nhp, tiles, gb, nexp);
    else if (d <= 16) rc = launch_grad<T, 16>(st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, work, nhp, tiles, gb, nexp);
    else if (d <= 32) rc = launch_grad<T, 32>(st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, work, nhp, tiles, gb, nexp);
    else if (d <= 64) rc = launch_grad<T, 64>(st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, work, nhp, tiles, gb, nexp);
    else { pg_set_error("pg_nlml_grad: d=%d > %d", d, PG_MAX_DIM); return -2; }
    if (rc) return rc;
    hipLaunchKernelGGL(pg_grad_reduce_kernel, dim3(nhp, 1, nexp), dim3(256), 0, st, spec, hp, work, tiles * ((tiles + GCH - 1) / GCH), nhp,
                       d, grad, presc, gb);
    PG_CHECK(hipGetLastError());
    return 0;
}
long pg_nlml_grad_worksize_impl(int n, int nhp) {
    const long tiles = (n + KT - 1) / KT;
    return tiles * tiles * nhp;
}
template int pg_nlml_grad_t<double>(hipStream_t, const pg_covspec&, const double*, const double*, long, int, int,
                                    const double*, long, const double*, double*, int, double*, long, int, long, long, long, long, long);
template int pg_nlml_grad_t<float>(hipStream_t, const pg_covspec&, const double*, const float*, long, int, int,
                                   const float*, long, const float*, double*, int, double*, long, int, long, long, long, long, long);
