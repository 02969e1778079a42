// Covariance assembly and the fused NLML-gradient contraction with the pairwise products on the MATRIX pipe (round 5).
//
// The VALU bodies of kbuild.hip spend d subtract/FMA pairs per element on the distance and -- in the contraction -- another 3 d
// operations per element on the per-coordinate sums: at D = 16 (BASELINE configs 4 and 5) the lower-only build ran at 0.40 and the
// contraction at 0.07-0.10 of the HBM peak, bound by fp64 / fp32 VALU issue.  Here both are written the way the reference writes its
// distance (PyGPR/covar.py:102-127: -2 X X^T + |x|^2 + |x'|^2, a matmul) on 16 x 16 blocks of v_mfma_f64_16x16x4 / v_mfma_f32_16x16x4:
//
//   build        sq_ij = |a_i|^2 + |b_j|^2 - 2 a_i . b_j         one MFMA per four coordinates, the norms as the accumulator's start
//   contraction  S_c   = sum_ij G_ij (a_ic - b_jc)^2              with G = W o K (weights times covariance / its radial derivative)
//                      = sum_j [ (G^T A^2)_jc - 2 b_jc (G^T A)_jc + b_jc^2 (G^T 1)_j ]
//                the two products G^T A, G^T A^2 take the block of G STRAIGHT from the registers it was computed in: the accumulator
//                layout of the first product (row on the register index / lane group, column on the lane) is the A-operand layout of
//                a product that sums over the block's ROW index (cdna_hip_programming.md: an accumulator tile as the next MFMA's
//                operand) -- no LDS round trip, no lane movement.
//
// a = (x - x_0) l: coordinates relative to the first point (stationary kernels do not care; the expansion's absolute error is
// eps |a|^2, so uncentred data would lose digits) times the inverse length scales.  What stays on the VALU is what is per ELEMENT:
// the exponential, the weight, the clamp -- about 25 operations, independent of d.  One stationary component (+ white noise), d <= 16;
// everything else keeps the kernels of kbuild.hip.  PG_KB_MFMA=0 / PG_GRAD_MFMA=0 switch back.
#include "kbuild.h"
#include "kfun.h"
#include "kmfma.h"
#include <cstdlib>
#include <type_traits>

#define KT 64

// ------------------------------------------------------------------------------------------------
// a tile of 64 points in LDS, point-major: pts[p][DP + 2] = (X[p0 + p][k] - x0[k]) l_k for k < d, zero up to DP (a multiple of 4)
// ------------------------------------------------------------------------------------------------
template <typename T, int DP> struct PtTile {
    static constexpr int LDP = DP + 2;
    static constexpr int NPF = (KT * DP) / 256;     // elements a thread stages: its coordinate k = tid % DP of NPF points
    static constexpr int PSTEP = 256 / DP;
    T pf[NPF];
    bool ok[NPF];
    // global -> registers: unconditional loads at clamped addresses, so that the prefetch of the next tile stays a prefetch
    __device__ __forceinline__ void load(const T* __restrict__ X, long ldx, int npts, int p0, int d, int tid) {
        const int k = tid % DP, pb = tid / DP;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int gp = p0 + pb + u * PSTEP;
            ok[u] = k < d && gp < npts;
            pf[u] = X[(long)min(max(gp, 0), max(npts - 1, 0)) * ldx + min(k, d - 1)];
        }
    }
    __device__ __forceinline__ void store(T* pts, double sc, double x0k, int tid) const {
        const int k = tid % DP, pb = tid / DP;
#pragma unroll
        for (int u = 0; u < NPF; ++u) pts[(pb + u * PSTEP) * LDP + k] = ok[u] ? (T)(((double)pf[u] - x0k) * sc) : (T)0;
    }
    // nrm[p] = |pts[p]|^2, one thread per point (call with tid < 64)
    static __device__ __forceinline__ void norms(const T* pts, T* nrm, int p) {
        T s = (T)0;
#pragma unroll
        for (int k = 0; k < DP; ++k) { const T v = pts[p * LDP + k]; s += v * v; }
        nrm[p] = s;
    }
};

// covariance value (and the factor `base` of its length-scale derivative, dK/dl_k = coef base l_k D_k^2) from the scaled squared distance
template <typename T, int KIND> struct KmVal;
template <> struct KmVal<double, PG_KIND_RBF> {
    static __device__ __forceinline__ void run(double sq, double sig2, const double* tab, double& kv, double& base) {
        (void)sig2;
        kv = base = pg_exp_tab(-sq, tab);           // sigma^2 folded into the table
    }
};
// fp32: the hardware's own exponential and root (v_exp_f32 on x log2 e, v_sqrt_f32: about 1 ulp each, the argument's rounding adds
// |x| 2^-24 relative) -- the library forms cost some fifteen and eight instructions per element, which is what bound the fp32 bodies
// (VALU issue: the fp32 matrix pipe runs beside it).  2-3 ulp in all, inside the fp32 tolerances of the suite (4e-6 on K).
__device__ __forceinline__ float km_expf(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float km_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }
template <> struct KmVal<float, PG_KIND_RBF> {
    static __device__ __forceinline__ void run(float sq, float sig2, const double*, float& kv, float& base) { kv = base = sig2 * km_expf(-sq); }
};
template <> struct KmVal<double, PG_KIND_MATERN52> {
    static __device__ __forceinline__ void run(double sq, double sig2, const double* tab, double& kv, double& base) {
        (void)sig2;
        const double s5 = 2.23606797749978969641;
        const double r = pg_sqrt_pos(sq), e = pg_exp_tab(-s5 * r, tab);
        base = (1.0 + s5 * r) * e;
        kv = base + (5.0 / 3.0) * sq * e;
    }
};
template <> struct KmVal<float, PG_KIND_MATERN52> {
    static __device__ __forceinline__ void run(float sq, float sig2, const double*, float& kv, float& base) {
        const float s5 = 2.2360679775f;
        const float r = km_sqrtf(sq), e = sig2 * km_expf(-s5 * r);
        base = (1.0f + s5 * r) * e;
        kv = base + (5.0f / 3.0f) * sq * e;
    }
};

// ------------------------------------------------------------------------------------------------
// covariance build: symmetric lower-only builds and cross builds of ONE stationary component (+ white noise on the diagonal)
// ------------------------------------------------------------------------------------------------
// Strips of up to S tiles of one tile row per workgroup, as pg_kbuild_kernel: the row tile, its MFMA fragments and norms are paid once
// per strip, the next tile's column points travel global -> registers -> the other LDS buffer while the current tile is computed.
// Wave w owns rows 16 w .. 16 w + 15 of the 64 x 64 tile: four 16 x 16 blocks, DP / 4 MFMAs each, then per element the clamp, the
// exponential and the store (accumulator layout: a store instruction writes four rows of 16 contiguous elements -- whole 128-byte
// lines in fp64).
#define KM_TLD 65     // odd leading dimension of the transposed tile (mirrored builds)
template <typename T, int DP, int KIND, bool MIRROR>
__global__ __launch_bounds__(256) void pg_kbuild_mfma_kernel(pg_covspec spec, const double* __restrict__ hp, const T* __restrict__ Xr, long ldr,
                                                             int nr, const T* __restrict__ Xc, long ldc, int nc, int d, int symmetric,
                                                             double jitter, T* __restrict__ K, long ldk, int ctile0, int ctile1, int S,
                                                             long eX, long ehp, long eK, long eXr) {
    typedef PtTile<T, DP> PT;
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int LDP = PT::LDP, NS = DP / 4;
    Xr += blockIdx.y * eXr; Xc += blockIdx.y * eX; hp += blockIdx.y * ehp; K += blockIdx.y * eK;
    int tr, tcs, ntile;
    kb_strip_of(blockIdx.x, symmetric, ctile0, ctile1, S, tr, tcs, ntile);
    __shared__ T xr[KT * LDP], xc[2][KT * LDP], nrr[KT], nrc[2][KT];
    __shared__ T tt[MIRROR ? KT * KM_TLD : 1];                          // MIRROR: the tile transposed, for the block above the diagonal
    __shared__ double tab[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int o = spec.off[0];
    const int kk = tid % DP;
    const double sc = kk < d ? hp[o + 1 + kk] : 0.0;
    const double x0k = (kk < d && nc > 0) ? (double)Xc[kk] : 0.0;       // the origin: this expert's first column point
    PT pr, pc;
    pc.load(Xc, ldc, nc, tcs * KT, d, tid);
    pr.load(Xr, ldr, nr, tr * KT, d, tid);
    const double sg = hp[o];
    const T sig2 = (T)(sg * sg);
    double dgd = jitter;
    for (int i = 0; i < spec.nnoise; ++i) { const double s = hp[spec.noise_off[i]]; dgd += s * s; }
    const T dg = (T)dgd;
    if (sizeof(T) == 8 && tid < 32) tab[tid] = sg * sg * pg_exp2_32[tid];
    pr.store(xr, sc, x0k, tid);
    pc.store(xc[0], sc, x0k, tid);
    __syncthreads();
    if (tid < KT) PT::norms(xr, nrr, tid);
    else if (tid < 2 * KT) PT::norms(xc[0], nrc[0], tid - KT);
    __syncthreads();
    T a[NS], nri[4];
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = xr[(16 * wave + c16) * LDP + 4 * s + g];
#pragma unroll
    for (int r = 0; r < 4; ++r) nri[r] = nrr[16 * wave + Mfma<T>::row(lane, r)];
    const int tcl = tcs + ntile - 1;
    const bool interior = (!symmetric || tcl < tr) && (tr + 1) * KT <= nr && (tcl + 1) * KT <= nc;   // workgroup-uniform: no per-element fix-ups
    for (int t = 0; t < ntile; ++t) {
        const int tc = tcs + t, cur = t & 1;
        if (t + 1 < ntile) pc.load(Xc, ldc, nc, (tc + 1) * KT, d, tid);          // in flight while this tile is computed
        const T* xb = xc[cur];
        const T* nb = nrc[cur];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int j0 = 16 * cb;
            T b[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) b[s] = xb[(j0 + c16) * LDP + 4 * s + g];
            const T ncj = nb[j0 + c16];
            acc_t acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = (T)-0.5 * (nri[r] + ncj);      // the product then ENDS as -1/2 of the squared distance
#pragma unroll
            for (int s = 0; s < NS; ++s) acc = Mfma<T>::run(a[s], b[s], acc);
            const int gj = tc * KT + j0 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = tr * KT + 16 * wave + Mfma<T>::row(lane, r);
                T sq = (T)-2 * acc[r];
                sq = sq < (T)0 ? (T)0 : sq;                                      // rounding of the expansion; a NaN stays a NaN
                if (!interior && symmetric && gi == gj && sq == sq) sq = (T)0;   // a point against itself: exactly sigma^2
                T kv, base;
                KmVal<T, KIND>::run(sq, sig2, tab, kv, base);
                if (!interior) {
                    if (gi >= nr || gj >= nc) kv = (symmetric && gi == gj) ? (T)1 : (T)0;     // padding: identity / zeros
                    else if (symmetric && gi == gj) kv += dg;
                }
                K[(long)gi * ldk + gj] = kv;
                if (MIRROR) { if (tc < tr) tt[(j0 + c16) * KM_TLD + 16 * wave + Mfma<T>::row(lane, r)] = kv; }
            }
        }
        if (MIRROR && tc < tr) {      // K[tc-tile rows][tr-tile columns] = the transpose, read back row-wise: 512-byte runs per wave
            __syncthreads();
            for (int idx = tid; idx < KT * KT; idx += 256) {
                const int lj = idx >> 6, li = idx & 63;
                K[(long)(tc * KT + lj) * ldk + tr * KT + li] = tt[lj * KM_TLD + li];
            }
        }
        if (t + 1 < ntile) {
            pc.store(xc[cur ^ 1], sc, x0k, tid);
            __syncthreads();
            if (tid < KT) PT::norms(xc[cur ^ 1], nrc[cur ^ 1], tid);
            __syncthreads();
        }
    }
}

template <typename T, int DP, int KIND, bool MIRROR>
static int kb_mfma_launch(hipStream_t st, const pg_covspec& spec, const double* hp, const T* Xr, long ldr, int nr, const T* Xc, long ldc, int nc,
                          int d, int symmetric, double jitter, T* K, long ldk, int c0, int c1, int S, long strips, int nexp, long eX, long ehp,
                          long eK, long eXr) {
    hipLaunchKernelGGL((pg_kbuild_mfma_kernel<T, DP, KIND, MIRROR>), dim3((unsigned)strips, (unsigned)nexp), dim3(256), 0, st, spec, hp, Xr, ldr, nr,
                       Xc, ldc, nc, d, symmetric, jitter, K, ldk, c0, c1, S, eX, ehp, eK, eXr);
    PG_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
int pg_kbuild_mfma(hipStream_t st, const pg_covspec& spec, const double* hp, const T* Xr, long ldr, int nr, const T* Xc, long ldc, int nc, int d,
                   int symmetric, int mirror, double jitter, T* K, long ldk, int c0, int c1, int S, long strips, int nexp, long eX, long ehp,
                   long eK, long eXr) {
    const int kind = spec.kind[0];
#define KB_ARGS st, spec, hp, Xr, ldr, nr, Xc, ldc, nc, d, symmetric, jitter, K, ldk, c0, c1, S, strips, nexp, eX, ehp, eK, eXr
#define KB_GO(DP)                                                                                                          \
    do {                                                                                                                   \
        if (kind == PG_KIND_RBF) return mirror ? kb_mfma_launch<T, DP, PG_KIND_RBF, true>(KB_ARGS)                         \
                                               : kb_mfma_launch<T, DP, PG_KIND_RBF, false>(KB_ARGS);                       \
        return mirror ? kb_mfma_launch<T, DP, PG_KIND_MATERN52, true>(KB_ARGS)                                             \
                      : kb_mfma_launch<T, DP, PG_KIND_MATERN52, false>(KB_ARGS);                                           \
    } while (0)
    if (d <= 4) KB_GO(4);
    if (d <= 8) KB_GO(8);
    KB_GO(16);
#undef KB_GO
#undef KB_ARGS
}
template int pg_kbuild_mfma<double>(hipStream_t, const pg_covspec&, const double*, const double*, long, int, const double*, long, int, int, int, int,
                                    double, double*, long, int, int, int, long, int, long, long, long, long);
template int pg_kbuild_mfma<float>(hipStream_t, const pg_covspec&, const double*, const float*, long, int, const float*, long, int, int, int, int,
                                   double, float*, long, int, int, int, long, int, long, long, long, long);

// ------------------------------------------------------------------------------------------------
// fused gradient contraction: g_k = 1/2 sum_ij (K^-1 - a a^T)_ij dK_ij/dtheta_k over the lower triangle (loss.py:116-121 by the K^-1 route)
// ------------------------------------------------------------------------------------------------
// A workgroup owns ONE tile column (64 columns j) and walks `gch` tile rows down from (or below) the diagonal; wave w owns columns
// 16 w .. 16 w + 15 of it.  Per 16 x 16 block: the distance product (DP / 4 MFMAs), per element the covariance, the weight
// W = c (K^-1 - a_i a_j) (c = 2 below the diagonal, 1 on it) and G = W base, then G^T A and G^T A^2 (4 MFMAs each; DP <= 8: A | A^2 share
// the sixteen columns of one).  Those two products accumulate in the matrix pipe's registers for the whole walk (their rows are the
// wave's columns j); they, the column sums of G and the sigma / noise sums are folded once per workgroup into part[blk][nhp]
// (entries in the `presc` convention of pg_grad_reduce_kernel: sums of (l_k D_k)^2 terms).
template <typename T, int DP, int KIND, bool GRIDX_COL = true>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 3 : 2) void pg_grad_mfma_kernel(pg_covspec spec, const double* __restrict__ hp, const T* __restrict__ X, long ldx, int n,
                                                           int d, const T* __restrict__ Kinv, long ldk, const T* __restrict__ alpha,
                                                           double* __restrict__ part, int nhp, GradBatch gb, int gch) {
    typedef PtTile<T, DP> PT;
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int LDP = PT::LDP, NS = DP / 4;
    constexpr bool PACK = DP <= 8;
    X += blockIdx.z * gb.eX; hp += blockIdx.z * gb.ehp; Kinv += blockIdx.z * gb.eK; alpha += blockIdx.z * gb.ea; part += blockIdx.z * gb.epart;
    // grid.x = tile column (fastest), grid.y = which stretch of `gch` tile rows below its diagonal: the workgroups in flight together
    // walk the SAME stretch of every tile column -- equally long walks, and K^-1 is read in whole row bands instead of sixteen 512-byte
    // columns of every row at once.  N = 16384, fp64: D = 16 721 -> 437 us, D = 8 491 -> 309 us (PG_GRAD_GRID=0: the transposed mapping of the
    // first version, tile column on grid.y; fp32 does not care: 1031 us both ways at n = 33792).
    constexpr bool colfast = GRIDX_COL;
    const int tc = colfast ? blockIdx.x : blockIdx.y, tiles = colfast ? gridDim.x : gridDim.y;
    const int chunk = colfast ? blockIdx.y : blockIdx.x, nchunk = colfast ? gridDim.y : gridDim.x;
    const int r_begin = tc + chunk * gch, r_end = min(r_begin + gch, tiles);
    const int blk = tc * nchunk + chunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    for (int idx = tid; idx < nhp; idx += 256) part[(long)blk * nhp + idx] = 0.0;
    if (r_begin >= tiles) return;
    __shared__ T xc[KT * LDP], ncs[KT], acs[KT], xr[2][KT * LDP], nrs[2][KT], ars[2][KT], cs[4][16];
    __shared__ double tab[32], tab2[32], red[4][18];   // tab2 = 2 tab: the weight 2 of an interior tile rides in the covariance value
    constexpr bool W2TAB = true;
    const T sig2x2 = (T)(2.0 * hp[spec.off[0]] * hp[spec.off[0]]);
    const int o = spec.off[0];
    const int kk = tid % DP;
    const double sc = kk < d ? hp[o + 1 + kk] : 0.0;
    const double x0k = kk < d ? (double)X[kk] : 0.0;
    const double sg = hp[o];
    const T sig2 = (T)(sg * sg);
    PT pcol, prow;
    pcol.load(X, ldx, n, tc * KT, d, tid);
    prow.load(X, ldx, n, r_begin * KT, d, tid);
    T apf = (T)0;                                                   // the row tile's weights alpha_i, prefetched like its points
    if (tid < KT) {
        const int gc = tc * KT + tid, gr = r_begin * KT + tid;
        acs[tid] = gc < n ? alpha[gc] : (T)0;
        apf = gr < n ? alpha[gr] : (T)0;
    }
    if (sizeof(T) == 8 && tid < 32) { tab[tid] = sg * sg * pg_exp2_32[tid]; tab2[tid] = 2.0 * sg * sg * pg_exp2_32[tid]; }
    pcol.store(xc, sc, x0k, tid);
    prow.store(xr[0], sc, x0k, tid);
    if (tid < KT) ars[0][tid] = apf;
    __syncthreads();
    if (tid < KT) PT::norms(xc, ncs, tid);
    else if (tid < 2 * KT) PT::norms(xr[0], nrs[0], tid - KT);
    __syncthreads();
    // this wave's column block: fragments of the distance product, norm, weight
    const int j0 = 16 * wave;
    const int gj = tc * KT + j0 + c16;
    T bq[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) bq[s] = xc[(j0 + c16) * LDP + 4 * s + g];
    const T ncj = ncs[j0 + c16], aj = acs[j0 + c16];
    acc_t P, P2;
#pragma unroll
    for (int r = 0; r < 4; ++r) { P[r] = (T)0; P2[r] = (T)0; }
    T pgs = (T)0;                     // this lane's share of the column sums of G (column j0 + c16)
    double accs = 0.0, trw = 0.0;     // sum W K (the sigma entry), sum of the diagonal's W (the noise entries)
    const long jcol = min((long)gj, ldk - 1);
    // K^-1 one block ahead of the arithmetic (a whole tile row ahead -- sixteen values per lane in flight -- was measured: no gain, the
    // kernel is not bound by its loads: tools/probe_tile_bodies.py).  The blocks follow one another sixteen rows
    // apart through the whole walk: one running pointer per accumulator register, advanced by 16 ldk per block (a 64-bit multiply per
    // load before); only a block that reaches past the last real row takes the clamped form.
    constexpr bool DEEP = false;
    constexpr int NQ = DEEP ? 4 : 1;
    const T* kp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) kp[r] = Kinv + (long)(r_begin * KT + Mfma<T>::row(lane, r)) * ldk + jcol;
    const long kstep = 16 * ldk;
    int krow = r_begin * KT;          // first row of the block the pointers stand on
    const int krow_end = r_end * KT;
    T kq[NQ][4];                      // [block in flight][register]
    auto kfetch = [&](T (&dst)[4]) {  // the block at krow -> dst, pointers on to the next one
        if (krow + 16 <= n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = *kp[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = Kinv[(long)min(krow + Mfma<T>::row(lane, r), n - 1) * ldk + jcol];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) kp[r] += kstep;
        krow += 16;
    };
#pragma unroll
    for (int q = 0; q < NQ; ++q) kfetch(kq[q]);
    for (int tr = r_begin, it = 0; tr < r_end; ++tr, ++it) {
        const int cur = it & 1;
        const bool more = tr + 1 < r_end;
        if (more) {
            prow.load(X, ldx, n, (tr + 1) * KT, d, tid);
            if (tid < KT) { const int gr = (tr + 1) * KT + tid; apf = gr < n ? alpha[gr] : (T)0; }
        }
        const T* xb = xr[cur];
        const T* nb = nrs[cur];
        const T* ab = ars[cur];
        T accs_t = (T)0;                                             // this tile row's share of sum W K
        typedef float pf2 __attribute__((ext_vector_type(2)));
        pf2 acc2 = {0.0f, 0.0f}, pgs2 = {0.0f, 0.0f};                // (fp32 interior body: the same sums in pairs)
        // one tile row = four 16 x 16 blocks.  INTERIOR (strictly below the diagonal, inside the real points: weight 2 everywhere) is
        // a body without a single per-element test -- one basic block, so the four elements of a lane and the products of
        // neighbouring blocks interleave; the diagonal tile and a ragged last tile take the general body.
        auto tile_row = [&](auto interior_c) {
            constexpr bool INTERIOR = decltype(interior_c)::value;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const int i0 = 16 * rb;
                T kin[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) kin[r] = kq[DEEP ? rb : 0][r];
                if (krow < krow_end) kfetch(kq[DEEP ? rb : 0]);
                T af[NS], nri[4], ai[4];
#pragma unroll
                for (int s = 0; s < NS; ++s) af[s] = xb[(i0 + c16) * LDP + 4 * s + g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    nri[r] = nb[i0 + Mfma<T>::row(lane, r)];
                    ai[r] = ab[i0 + Mfma<T>::row(lane, r)];
                }
                acc_t acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = (T)-0.5 * (nri[r] + ncj);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc = Mfma<T>::run(af[s], bq[s], acc);
                T gr_[4];
                if constexpr (sizeof(T) == 4 && INTERIOR) {
                    // fp32, interior: the four elements of a lane as two PAIRS on the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 /
                    // v_pk_add_f32: two elements per issue slot); only the root and the exponential stay one element at a time
                    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        f2 sq = {(float)acc[r], (float)acc[r + 1]};
                        sq = sq * -2.0f;
                        const f2 ai2 = {(float)ai[r], (float)ai[r + 1]}, kin2 = {(float)kin[r], (float)kin[r + 1]};
                        const f2 w = kin2 - ai2 * (float)aj;                       // (the weight 2 rides in sig2x2)
                        f2 kv, base;
                        if (KIND == PG_KIND_RBF) {
                            const f2 t = sq * -1.44269504088896341f;
                            const f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                            kv = base = e * (float)sig2x2;
                        } else {
                            sq = __builtin_elementwise_max(sq, (f2){0.0f, 0.0f});
                            const f2 rr = {__builtin_amdgcn_sqrtf(sq.x), __builtin_amdgcn_sqrtf(sq.y)};
                            const f2 t = rr * (-2.2360679775f * 1.44269504088896341f);
                            f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                            e = e * (float)sig2x2;
                            base = rr * 2.2360679775f * e + e;
                            kv = sq * (5.0f / 3.0f) * e + base;
                        }
                        const f2 gg = w * base;
                        acc2 += w * kv;
                        pgs2 += gg;
                        gr_[r] = (T)gg.x; gr_[r + 1] = (T)gg.y;
                    }
                } else
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T sq = (T)-2 * acc[r];
                    // (a squared exponential takes the expansion's rounding below zero as it is: exp(1e-16) = 1; a root does not)
                    if (!INTERIOR || KIND != PG_KIND_RBF) sq = sq < (T)0 ? (T)0 : sq;
                    T w = kin[r] - ai[r] * aj;
                    if (INTERIOR) { if (!W2TAB) w *= (T)2; }
                    else {
                        const int gi = tr * KT + i0 + Mfma<T>::row(lane, r);
                        if (gi == gj && sq == sq) sq = (T)0;
                        if (gi >= n || gj > gi) w = (T)0;
                        else if (gj < gi) w *= (T)2;
                        else trw += (double)w;
                    }
                    T kv, base;
                    KmVal<T, KIND>::run(sq, (INTERIOR && W2TAB) ? sig2x2 : sig2, (INTERIOR && W2TAB) ? tab2 : tab, kv, base);
                    accs_t += w * kv;
                    gr_[r] = w * base;
                    pgs += gr_[r];
                }
                // G^T A and G^T A^2: register r of the block is the A operand of the k-step that covers its four rows
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int li = i0 + Mfma<T>::row(lane, r);
                    if (PACK) {
                        const T xv = c16 < 2 * DP ? xb[li * LDP + (c16 < DP ? c16 : c16 - DP)] : (T)0;
                        P = Mfma<T>::run(gr_[r], c16 < DP ? xv : xv * xv, P);
                    } else {
                        const T xv = xb[li * LDP + c16];
                        P = Mfma<T>::run(gr_[r], xv, P);
                        P2 = Mfma<T>::run(gr_[r], xv * xv, P2);
                    }
                }
            }
        };
        if (tr > tc && (tr + 1) * KT <= n) tile_row(std::true_type{});
        else tile_row(std::false_type{});
        accs += (double)accs_t + (double)acc2.x + (double)acc2.y;
        pgs += (T)(pgs2.x + pgs2.y);
        if (more) {
            prow.store(xr[cur ^ 1], sc, x0k, tid);
            if (tid < KT) ars[cur ^ 1][tid] = apf;
            __syncthreads();
            if (tid < KT) PT::norms(xr[cur ^ 1], nrs[cur ^ 1], tid);
            __syncthreads();
        }
    }
    // ---- fold: S_c = sum_j [ (G^T A^2)_jc - 2 b_jc (G^T A)_jc + b_jc^2 (G^T 1)_j ] over this wave's sixteen columns
    pgs += __shfl_xor(pgs, 16, 64);
    pgs += __shfl_xor(pgs, 32, 64);
    if (g == 0) cs[wave][c16] = pgs;                               // column sums of G
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lj = j0 + Mfma<T>::row(lane, r);
        if (PACK) {
            if (c16 < DP) v += -2.0 * (double)xc[lj * LDP + c16] * (double)P[r];
            else if (c16 < 2 * DP) v += (double)P[r];
        } else {
            v += (double)P2[r] - 2.0 * (double)xc[lj * LDP + c16] * (double)P[r];
        }
    }
    if (PACK) v += __shfl_down(v, DP, 64);                         // lane c16 < DP: its own A part + the A^2 part of lane c16 + DP
    __syncthreads();                                               // cs is written
    if (c16 < DP) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lj = 4 * g + q;
            const double xv = (double)xc[(j0 + lj) * LDP + c16];
            v += xv * xv * (double)cs[wave][lj];
        }
    }
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0 && c16 < DP) red[wave][c16] = v;
    {
        const double s1 = wave_sum(accs), s2 = wave_sum(trw);
        if (lane == 0) { red[wave][16] = s1; red[wave][17] = s2; }
    }
    __syncthreads();
    if (tid == 0) part[(long)blk * nhp + o] = red[0][16] + red[1][16] + red[2][16] + red[3][16];
    else if (tid <= d) part[(long)blk * nhp + o + tid] = red[0][tid - 1] + red[1][tid - 1] + red[2][tid - 1] + red[3][tid - 1];
    if (tid < spec.nnoise) part[(long)blk * nhp + spec.noise_off[tid]] = red[0][17] + red[1][17] + red[2][17] + red[3][17];
}

template <typename T, int DP, int KIND>
static int grad_mfma_launch(hipStream_t st, const pg_covspec& spec, const double* hp, const T* X, long ldx, int n, int d, const T* Kinv, long ldk,
                            const T* alpha, double* part, int nhp, int tiles, const GradBatch& gb, int nexp, int gch) {
    const int gridmode = getenv("PG_GRAD_GRID") ? atoi(getenv("PG_GRAD_GRID")) : 1;
    if (gridmode)
        hipLaunchKernelGGL((pg_grad_mfma_kernel<T, DP, KIND, true>), dim3(tiles, (tiles + gch - 1) / gch, nexp), dim3(256), 0, st, spec, hp, X, ldx, n,
                           d, Kinv, ldk, alpha, part, nhp, gb, gch);
    else
        hipLaunchKernelGGL((pg_grad_mfma_kernel<T, DP, KIND, false>), dim3((tiles + gch - 1) / gch, tiles, nexp), dim3(256), 0, st, spec, hp, X, ldx, n,
                           d, Kinv, ldk, alpha, part, nhp, gb, gch);
    PG_CHECK(hipGetLastError());
    return 0;
}

// part: [tiles x ceil(tiles / gch)][nhp] partial rows per expert (fits pg_nlml_grad_worksize's tiles^2 nhp); *nblk receives the row count
template <typename T>
int pg_grad_mfma(hipStream_t st, const pg_covspec& spec, const double* hp, const T* X, long ldx, int n, int d, const T* Kinv, long ldk,
                 const T* alpha, double* part, int nhp, int tiles, const GradBatch& gb, int nexp, int* nblk) {
    // tile rows per workgroup: a sixteenth of the tile column's length, 2 ... 32 (measured, us at gch = 2 / 4 / 8 / 16 / 32 / 64 -- N = 16384, D = 16, fp64:
    // 562 / 481 / 444 / 437 / 467 / 567; n = 4096: 53 / 55 / 57 / 70 / 124 / 233; fp32 Matern n = 33792: 1419 / 1162 / 1060 / 1031 / 1004 / 1045)
    const int gch_env = getenv("PG_GRAD_GCH") ? atoi(getenv("PG_GRAD_GCH")) : 0;
    const int gch = gch_env > 0 ? std::min(gch_env, 64) : std::max(2, std::min(tiles / 16, 32));
    *nblk = tiles * ((tiles + gch - 1) / gch);
    const int kind = spec.kind[0];
#define GR_GO(DP)                                                                                                                          \
    return kind == PG_KIND_RBF                                                                                                             \
               ? grad_mfma_launch<T, DP, PG_KIND_RBF>(st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, part, nhp, tiles, gb, nexp, gch)       \
               : grad_mfma_launch<T, DP, PG_KIND_MATERN52>(st, spec, hp, X, ldx, n, d, Kinv, ldk, alpha, part, nhp, tiles, gb, nexp, gch)
    if (d <= 4) { GR_GO(4); }
    if (d <= 8) { GR_GO(8); }
    GR_GO(16);
#undef GR_GO
}
template int pg_grad_mfma<double>(hipStream_t, const pg_covspec&, const double*, const double*, long, int, int, const double*, long, const double*,
                                  double*, int, int, const GradBatch&, int, int*);
template int pg_grad_mfma<float>(hipStream_t, const pg_covspec&, const double*, const float*, long, int, int, const float*, long, const float*,
                                 double*, int, int, const GradBatch&, int, int*);
