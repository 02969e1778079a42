// Cholesky leaf: factor one 128x128 diagonal block entirely in LDS (one workgroup) and produce
// its inverse, which turns every triangular solve above it into an MFMA GEMM.
//
// Blocked right-looking with 16-wide micro-panels, software-pipelined; per micro-panel jb:
//   A. wave 0 factors the 16x16 diagonal block in REGISTERS: lane r holds row r; per column the pivot and the
//      multipliers are broadcast with v_readlane (1/sqrt by v_rsq + 2 Newton steps).  A non-positive / NaN pivot sets
//      *info = global column + 1 (LAPACK convention, first failure wins) and stops.
//      Meanwhile the other waves finish the trailing update of the PREVIOUS micro-panel (all tiles except its first
//      tile column), which nothing in A or B depends on.
//   B. panel solve  P <- P D^-T  by forward substitution, one thread per row (D broadcast from LDS)
//   C. first tile column of the trailing update  T <- T - P P^T  (next diagonal block + next panel), MFMA
// The inverse X = L^-1 is built inside the same loop by waves that would otherwise idle (block-row forward recurrence on
// 16x16 tiles, X[p][q] = -D_p^-1 sum_{r=q}^{p-1} L[p][r] X[r][q] with X[q][q] = D_q^-1):
//   - phase B of micro-panel jb (panel rows x D^-T, one thread per row) takes 16 more threads that run the same
//     substitution on identity rows: that is Dinv[jb], at no extra time;
//   - during phase A of micro-panel jb+1 the deferred-update work of waves 1-7 shrinks as jb grows while block row jb of X
//     grows: they share one work list.  X[p][q] (p > q) lives in the upper tile (q, p) of S, which the factorisation never
//     touches; after the loop only Dinv[7] and block row 7 are left.
// LDS: S[128][130] + 8 x [16][18] diagonal inverses, all in the dynamic region.
#include "leaf.h"
#include "chainstep.h"
#include <algorithm>
#include <cstdlib>

#define NB 128
#define LD 130
#define DLD 18
// 12 waves: wave 0 runs the serial diagonal step, the MFMA phases are spread over all of them.  Measured per leaf (same box):
// 256 threads 58.2 us, 384 55.2, 512 51.8, 640 51.1, 768 49.7, 896 56.2, 1024 55.5.
#ifndef NTH
#define NTH 768
#endif
#define NWV (NTH / 64)

// 16-byte (two doubles) / 8-byte (two floats) accesses that bypass this CU's L1 and write through the XCD's L2 (sc1): what a
// kernel uses for bytes another resident kernel hands it, or takes from it, without a fence (chainstep.hip; guideline 16).
typedef int pg_v4i __attribute__((ext_vector_type(4)));
typedef double pg_pair_d __attribute__((ext_vector_type(2)));
typedef float pg_pair_f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pg_pair_d ld_pair_wt(const double* base, __amdgpu_buffer_rsrc_t r, long elem_off) {
    (void)base;
    return __builtin_bit_cast(pg_pair_d, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(elem_off * 8), 0, 16));
}
__device__ __forceinline__ void st_pair_wt(double* base, __amdgpu_buffer_rsrc_t r, long elem_off, pg_pair_d v) {
    (void)base;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pg_v4i, v), r, (int)(elem_off * 8), 0, 16);
}
__device__ __forceinline__ pg_pair_f ld_pair_wt(const float* base, __amdgpu_buffer_rsrc_t r, long elem_off) {
    (void)r;
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(base + elem_off), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(pg_pair_f, u);
}
__device__ __forceinline__ void st_pair_wt(float* base, __amdgpu_buffer_rsrc_t r, long elem_off, pg_pair_f v) {
    (void)r;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(base + elem_off), __builtin_bit_cast(unsigned long long, v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T> __device__ __forceinline__ T bcast_lane(T v, int src);
template <> __device__ __forceinline__ double bcast_lane<double>(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
template <> __device__ __forceinline__ float bcast_lane<float>(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

__device__ __forceinline__ double inv_sqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
// 1 / x to working precision: hardware seed + Newton (two steps in fp64, one in fp32)
__device__ __forceinline__ double recip(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}
__device__ __forceinline__ float recip(float x) {
    float y = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(y, e, y);
}
__device__ __forceinline__ float inv_sqrt(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    r = r * (1.5f - 0.5f * x * r * r);
    return r;
}

// C(16x16) = beta * C + alpha * A(16xK) * op(B); A row-major [16][K] at lda; B either [K][16] (TB = false)
// or [16][K] (TB = true) at ldb; all in LDS.  One wave.  Fragments are read 16 k at a time ahead of the MFMAs
// (and the C tile up front) so the LDS latency is paid once per group instead of once per k-step.
template <typename T, bool TB>
__device__ __forceinline__ void lds_tile_mm(T* C, int ldc, const T* A, int lda, const T* B, int ldb, int K, T alpha,
                                            T beta, int lane) {
    typename Mfma<T>::acc_t acc;
    const int fr = lane & 15, fk = lane >> 4;
    T cin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc[r] = (T)0;
        cin[r] = (beta != (T)0) ? C[Mfma<T>::row(lane, r) * ldc + fr] : (T)0;
    }
    for (int k = 0; k < K; k += 16) {
        T a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = A[fr * lda + k + 4 * q + fk];
            b[q] = TB ? B[fr * ldb + k + 4 * q + fk] : B[(k + 4 * q + fk) * ldb + fr];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = Mfma<T>::run(a[q], b[q], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) C[Mfma<T>::row(lane, r) * ldc + fr] = alpha * acc[r] + beta * cin[r];
}

// C(16x16) -= A(16x16) B(16x16)^T, all three row-major at leading dimension LD in LDS: the K = 16 update tile of the leaf's deferred
// trailing update with ALL twelve LDS reads in flight before the first MFMA (lds_tile_mm as compiled read two fragments, waited, ran two
// MFMAs, read two more, waited, ... and fetched the C tile last: four LDS round trips and four dependent MFMAs in series, 0.6 us per
// tile stamped inside the leaf).
template <typename T>
__device__ __forceinline__ void lds_tile_update16(T* C, const T* A, const T* B, int lane) {
    typename Mfma<T>::acc_t acc;
    const int fr = lane & 15, fk = lane >> 4;
    T cin[4], a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cin[r] = C[Mfma<T>::row(lane, r) * LD + fr];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[q] = A[fr * LD + 4 * q + fk];
        b[q] = B[fr * LD + 4 * q + fk];
    }
    __builtin_amdgcn_sched_barrier(0);          // nothing below moves above the reads' issue
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (T)0;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = Mfma<T>::run(a[q], b[q], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[Mfma<T>::row(lane, r) * LD + fr] = cin[r] - acc[r];
}

// One 16x16 tile of the inverse, X[p][q] (p > q), by one wave: W = sum_{r=q}^{p-1} L[p][r] X[r][q], X[p][q] = -Dinv[p] W.
// X[r][q] for r > q is read from upper tile (q, r); the result goes to upper tile (q, p) (also the scratch for W).
// The two halves are also available on their own: the sum only needs block row p of L and the rows of X above it, so
// for the last block row it runs before the last diagonal inverse exists.
template <typename T>
__device__ __forceinline__ void inv_tile_sum(T* S, const T* Dinv, int p, int q, int lane) {
    typename Mfma<T>::acc_t acc;
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (T)0;
    const T* Lp = S + (16 * p + fr) * LD;
    for (int r = q; r < p; ++r) {
        T a[4], b[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a[kk] = Lp[16 * r + 4 * kk + fk];
            b[kk] = (r == q) ? Dinv[q * 16 * DLD + (4 * kk + fk) * DLD + fr] : S[(16 * q + 4 * kk + fk) * LD + 16 * r + fr];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = Mfma<T>::run(a[kk], b[kk], acc);
    }
    T* Wt = S + (16 * q) * LD + 16 * p;            // tile (q, p): W
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[Mfma<T>::row(lane, r) * LD + fr] = acc[r];
}
template <typename T>
__device__ __forceinline__ void inv_tile_finish(T* S, const T* Dinv, int p, int q, int lane) {
    typename Mfma<T>::acc_t acc;
    const int fr = lane & 15, fk = lane >> 4;
    T* Wt = S + (16 * q) * LD + 16 * p;
    T a[4], b[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a[kk] = Dinv[p * 16 * DLD + fr * DLD + 4 * kk + fk];
        b[kk] = Wt[(4 * kk + fk) * LD + fr];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (T)0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = Mfma<T>::run(a[kk], b[kk], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[Mfma<T>::row(lane, r) * LD + fr] = -acc[r];
}
template <typename T>
__device__ __forceinline__ void inv_tile(T* S, const T* Dinv, int p, int q, int lane) {
    typename Mfma<T>::acc_t acc;
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (T)0;
    const T* Lp = S + (16 * p + fr) * LD;
    for (int r = q; r < p; ++r) {
        T a[4], b[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a[kk] = Lp[16 * r + 4 * kk + fk];
            b[kk] = (r == q) ? Dinv[q * 16 * DLD + (4 * kk + fk) * DLD + fr] : S[(16 * q + 4 * kk + fk) * LD + 16 * r + fr];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = Mfma<T>::run(a[kk], b[kk], acc);
    }
    T* Wt = S + (16 * q) * LD + 16 * p;            // tile (q, p): W, then X[p][q]
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[Mfma<T>::row(lane, r) * LD + fr] = acc[r];
    T a[4], b[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a[kk] = Dinv[p * 16 * DLD + fr * DLD + 4 * kk + fk];
        b[kk] = Wt[(4 * kk + fk) * LD + fr];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (T)0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = Mfma<T>::run(a[kk], b[kk], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[Mfma<T>::row(lane, r) * LD + fr] = -acc[r];
}

// x <- x D^-T for one 16-long row x (forward substitution; D = factored 16x16 diagonal block in LDS, rd = reciprocals of
// its diagonal).  Panel rows use it for P <- P D^-T; identity rows e_i give column i of D^-1.
template <typename T>
__device__ __forceinline__ void row_solve16(T (&x)[16], const T* D, const T* rd) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        T sacc = x[c];
#pragma unroll
        for (int k = 0; k < c; ++k) sacc -= x[k] * D[c * LD + k];
        x[c] = sacc * rd[c];
    }
}

template <typename T>
__global__ __launch_bounds__(NTH) void pg_leaf_kernel(T* __restrict__ A, long lda, T* __restrict__ inv, long ldi,
                                                      int* __restrict__ info, int col0, int ablate, long eA, long eInv) {
    A += blockIdx.x * eA;                     // batched experts: one workgroup each, one info word each
    if (inv) inv += blockIdx.x * eInv;
    info += blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* S = reinterpret_cast<T*>(smem_raw);
    T* Dinv = S + NB * LD;                                  // [8][16][DLD]
    int& fail = *reinterpret_cast<int*>(Dinv + 8 * 16 * DLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (*info != 0) return;
    if (tid == 0) fail = 0;
    typedef T pair_t __attribute__((ext_vector_type(2)));
    for (int idx = tid; idx < NB * NB / 2; idx += NTH) {       // two columns per thread, 64 pairs per row
        const int i = idx >> 6, k = (idx & 63) * 2;
        pair_t v = {(T)0, (T)0};
        if (k <= i) {
            v = *reinterpret_cast<const pair_t*>(A + (long)i * lda + k);
            if (k + 1 > i) v[1] = (T)0;
        }
        *reinterpret_cast<pair_t*>(S + i * LD + k) = v;
    }
    __syncthreads();

    T* Rd = Dinv + 8 * 16 * DLD + 2;   // [16] reciprocals of the current diagonal block's diagonal (behind the flag)
    const bool want_inv = inv != nullptr && !(ablate & 2);   // ablate bit 1: no inverse
    for (int jb = 0; jb < ((ablate & 1) ? 0 : NB / 16); ++jb) {   // ablate bit 0: skip the factorisation loop
        const int c0 = jb * 16, r0 = c0 + 16;
        if (wave == 0 && !(ablate & 8)) {   // bit 3: skip the diagonal step
            // ---- A: factor the 16x16 diagonal block in registers of wave 0 (lane r = row r)
            const int r = lane & 15;
            T* D = S + c0 * LD + c0;
            T row[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) row[c] = D[r * LD + c];
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const T piv = bcast_lane(row[c], c);
                if (ok && !(piv > (T)0)) {   // wave-uniform
                    ok = false;
                    if (lane == 0) { fail = 1; atomicCAS(info, 0, col0 + c0 + c + 1); }
                }
                const T rs = ok ? inv_sqrt(piv) : (T)0;
                const T lrc = row[c] * rs;           // lane c: sqrt(piv); lanes r < c: 0 (upper part is zero)
                row[c] = lrc;
                if (lane == c) Rd[c] = rs;
#pragma unroll
                for (int k = c + 1; k < 16; ++k) row[k] -= lrc * bcast_lane(lrc, k);
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) D[r * LD + c] = (c <= r) ? row[c] : (T)0;
            }
        } else if (wave > 0 && jb > 0) {
            // ---- deferred part of the previous trailing update: tiles (ti, tj) with 1 <= tj <= ti; then block row
            // jb - 1 of the inverse (its diagonal inverse was formed during the previous phase B)
            const int pc0 = c0 - 16, pr0 = c0;           // previous panel's columns / first trailing row
            const int pnt = (NB - pr0) / 16;
            const int nrest = pnt * (pnt - 1) / 2;       // pairs (ti, tj): 1 <= tj <= ti <= pnt - 1
            const int nx = want_inv ? jb - 1 : 0;        // tiles X[jb-1][q], q = 0 .. jb-2
            for (int t = wave - 1; t < nrest + nx; t += NWV - 1) {
                if (t < nrest) {
                    int u = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                    while (u * (u + 1) / 2 > t) --u;
                    while ((u + 1) * (u + 2) / 2 <= t) ++u;
                    const int ti = u + 1, tj = t - u * (u + 1) / 2 + 1;
                    lds_tile_mm<T, true>(S + (pr0 + ti * 16) * LD + pr0 + tj * 16, LD, S + (pr0 + ti * 16) * LD + pc0, LD,
                                         S + (pr0 + tj * 16) * LD + pc0, LD, 16, (T)-1, (T)1, lane);
                } else {
                    inv_tile<T>(S, Dinv, jb - 1, t - nrest, lane);
                }
            }
        }
        __syncthreads();
        if (fail) return;
        if (r0 >= NB) break;
        const int nt = (NB - r0) / 16;   // 16-row tiles below the diagonal block
        // ---- B: P <- P D^-T by forward substitution, one thread per row; D and 1/diag are broadcast reads from LDS
        // the 16 threads after the panel rows run the same substitution on identity rows: row e_i D^-T = column i of D^-1
        // (on wave 7, which has no panel rows -- NB - r0 <= 112 -- so the two kinds of row never share a wave and diverge)
        if (tid < NB - r0) {
            T* Prow = S + (r0 + tid) * LD + c0;
            T x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = Prow[c];
            row_solve16<T>(x, S + c0 * LD + c0, Rd);
#pragma unroll
            for (int c = 0; c < 16; ++c) Prow[c] = x[c];
        } else if (wave == NWV - 1 && lane < 16 && want_inv) {
            T x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = (c == lane) ? (T)1 : (T)0;
            row_solve16<T>(x, S + c0 * LD + c0, Rd);
            T* dst = Dinv + jb * 16 * DLD + lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c * DLD] = x[c];
        }
        __syncthreads();
        // ---- C (first tile column): next diagonal block and next panel, T <- T - P P^T
        for (int t = wave; t < nt; t += NWV)
            lds_tile_mm<T, true>(S + (r0 + t * 16) * LD + r0, LD, S + (r0 + t * 16) * LD + c0, LD, S + r0 * LD + c0, LD, 16,
                                 (T)-1, (T)1, lane);
        __syncthreads();
    }

    // L back to global (diagonal tiles of C above wrote the strictly upper 16x16 corners: mask them)
    for (int idx = tid; idx < NB * NB / 2; idx += NTH) {
        const int i = idx >> 6, k = (idx & 63) * 2;
        pair_t v = *reinterpret_cast<const pair_t*>(S + i * LD + k);
        if (k > i) v[0] = (T)0;
        if (k + 1 > i) v[1] = (T)0;
        *reinterpret_cast<pair_t*>(A + (long)i * lda + k) = v;
    }
    if (!want_inv) return;
    // ---- block rows 0..6 of the inverse are final (last barrier of the loop): their stores go out first and overlap
    // with what is left to compute, the last diagonal inverse and block row 7
    auto store_inv_rows = [&](int row_lo, int row_hi) {
        for (int idx = row_lo * 64 + tid; idx < row_hi * 64; idx += NTH) {
            const int i = idx >> 6, k = (idx & 63) * 2;
            const int pb = i >> 4, qb = k >> 4, ii = i & 15;
            pair_t v = {(T)0, (T)0};
            if (qb == pb) {
                const T* Dv = Dinv + pb * 16 * DLD + ii * DLD + (k & 15);
                v[0] = (k <= i) ? Dv[0] : (T)0;
                v[1] = (k + 1 <= i) ? Dv[1] : (T)0;
            } else if (qb < pb) {
                const T* Xt = S + (16 * qb + ii) * LD + 16 * pb + (k & 15);   // X[pb][qb] lives in upper tile (qb, pb)
                v[0] = Xt[0];
                v[1] = Xt[1];
            }
            *reinterpret_cast<pair_t*>(inv + (long)i * ldi + k) = v;
        }
    };
    store_inv_rows(0, 112);
    if (tid < 16) {   // Rd still holds block 7's reciprocals
        T x[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = (c == tid) ? (T)1 : (T)0;
        row_solve16<T>(x, S + 112 * LD + 112, Rd);
#pragma unroll
        for (int c = 0; c < 16; ++c) Dinv[7 * 16 * DLD + c * DLD + tid] = x[c];
    }
    __syncthreads();
    for (int q = wave; q < 7; q += NWV) inv_tile<T>(S, Dinv, 7, q, lane);
    __syncthreads();
    store_inv_rows(112, 128);
}


// ------------------------------------------------------------------------------------------------
// Leaf, second form (default): the diagonal step and the panel solve are ONE register-resident tall-panel step.
// Waves 0-2 each hold the 16 rows of the diagonal block in lanes 0-15 (redundantly) and 48 rows of the panel below it in
// lanes 16-63; wave 3 holds the diagonal rows and 16 identity rows.  Per column c: the pivot and the multipliers l_kc come
// from the diagonal lanes by v_readlane (wave-uniform SGPRs), so the same instructions that factor the diagonal block
// perform the right-looking solve of the panel rows (and of the identity rows: D^-1) -- no LDS round trip and no barrier
// between "A" and "B", no branches (a bad pivot is recorded with a compare/select and acted on after the 16 columns).
// Waves 4-11 meanwhile run the deferred trailing update of the previous micro-panel and the inverse's block row, as before.
// ------------------------------------------------------------------------------------------------
template <typename T, bool WT>   // WT: tile and inverse move with write-through / L1-bypassing accesses (the coupled chain)
__device__ __forceinline__ void leaf2_body(char* smem_raw, T* __restrict__ A, long lda, T* __restrict__ inv, long ldi,
                                           int* __restrict__ info, int col0, int ablate, long long* tlog = nullptr) {
#define LTL(i) do { if (tlog && threadIdx.x == 0) tlog[i] = wall_clock64(); } while (0)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(A, 0, (int)((127 * lda + 128) * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(inv, 0, inv ? (int)((127 * ldi + 128) * sizeof(T)) : 0, 0x00020000);
    T* S = reinterpret_cast<T*>(smem_raw);
    T* Dinv = S + NB * LD;                                  // [8][16][DLD]
    int& fail = *reinterpret_cast<int*>(Dinv + 8 * 16 * DLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NAB = 4;                                  // waves of the tall-panel step (3 x 48 panel rows + identity rows)

    if (*info != 0) return;
    if (tid == 0) fail = 0;
    typedef T pair_t __attribute__((ext_vector_type(2)));
    // ablate bit 4: the round-2 data movement (whole tile loaded before the first step, L and the inverse stored after the last)
    const bool prog = !(ablate & 16);
    // Tile -> LDS in two sets, all loads issued up front.  Set A = columns 0-31 (what the first micro-panel's step and its first
    // update column touch) goes to LDS at once; set B = the rest is written behind the first step's barrier, so that its load
    // latency (and 3/4 of the tile's bytes) hides behind that step instead of in front of it.
    constexpr int NA = (NB * 16 + NTH - 1) / NTH, NBS = (NB * 48 + NTH - 1) / NTH;
    pair_t va[NA], vb[NBS];
    auto tile_load = [&](int i, int k) -> pair_t {
        pair_t v = pair_t{(T)0, (T)0};
        if (k <= i) {
            if (WT) v = ld_pair_wt(A, rA, (long)i * lda + k);
            else v = *reinterpret_cast<const pair_t*>(A + (long)i * lda + k);
        }
        return v;
    };
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int idx = tid + u * NTH, i = idx >> 4, k = (idx & 15) * 2;
        va[u] = (idx < NB * 16) ? tile_load(i, k) : pair_t{(T)0, (T)0};
    }
#pragma unroll
    for (int u = 0; u < NBS; ++u) {
        const int idx = tid + u * NTH, i = idx / 48, k = (16 + idx % 48) * 2;
        vb[u] = (idx < NB * 48) ? tile_load(i, k) : pair_t{(T)0, (T)0};
    }
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int idx = tid + u * NTH, i = idx >> 4, k = (idx & 15) * 2;
        if (k + 1 > i) va[u][1] = (T)0;
        if (idx < NB * 16) *reinterpret_cast<pair_t*>(S + i * LD + k) = va[u];
    }
    auto set_b_to_lds = [&]() {
#pragma unroll
        for (int u = 0; u < NBS; ++u) {
            const int idx = tid + u * NTH, i = idx / 48, k = (16 + idx % 48) * 2;
            if (k + 1 > i) vb[u][1] = (T)0;
            if (idx < NB * 48) *reinterpret_cast<pair_t*>(S + i * LD + k) = vb[u];
        }
    };
    if (!prog || (ablate & 1)) set_b_to_lds();
    __syncthreads();
    LTL(16);

    const bool want_inv = inv != nullptr && !(ablate & 2);
    // 16 columns of L (all 128 rows: zeros above the diagonal block) -> global, by the threads t0, t0 + nth, ...
    auto store_l_panel = [&](int jbp, int t0, int nth) {
        const int cc0 = 16 * jbp;
        for (int idx = t0; idx < NB * 8; idx += nth) {
            const int i = idx >> 3, k = cc0 + (idx & 7) * 2;
            pair_t v = {(T)0, (T)0};
            if (i >= cc0) v = *reinterpret_cast<const pair_t*>(S + i * LD + k);
            if (k > i) v[0] = (T)0;                                                // right of the diagonal inside the diagonal block
            if (k + 1 > i) v[1] = (T)0;
            if (WT) st_pair_wt(A, rA, (long)i * lda + k, v);
            else *reinterpret_cast<pair_t*>(A + (long)i * lda + k) = v;
        }
    };
    auto store_inv_rows = [&](int row_lo, int row_hi, int t0, int nth) {
        for (int idx = row_lo * 64 + t0; idx < row_hi * 64; idx += nth) {
            const int i = idx >> 6, k = (idx & 63) * 2;
            const int pb = i >> 4, qb = k >> 4, ii = i & 15;
            pair_t v = {(T)0, (T)0};
            if (qb == pb) {
                const T* Dv = Dinv + pb * 16 * DLD + ii * DLD + (k & 15);
                v[0] = (k <= i) ? Dv[0] : (T)0;
                v[1] = (k + 1 <= i) ? Dv[1] : (T)0;
            } else if (qb < pb) {
                const T* Xt = S + (16 * qb + ii) * LD + 16 * pb + (k & 15);
                v[0] = Xt[0];
                v[1] = Xt[1];
            }
            if (WT) st_pair_wt(inv, rI, (long)i * ldi + k, v);
            else *reinterpret_cast<pair_t*>(inv + (long)i * ldi + k) = v;
        }
    };
    constexpr int STW = 7;                                  // waves STW .. NWV-1 never have a tile of the first update column (nt <= 7)
    for (int jb = 0; jb < ((ablate & 1) ? 0 : NB / 16); ++jb) {
        const int c0 = jb * 16, r0 = c0 + 16;
        if (wave < NAB) {
            // ---- tall-panel step: lanes 0-15 = diagonal rows, lanes 16-63 = panel rows (waves 0-2) / identity rows (wave 3)
            const int pl = lane - 16;                                   // panel slot of this lane
            const int prow = r0 + 48 * wave + pl;                       // its row of S
            const bool is_diag = lane < 16;
            const bool is_panel = !is_diag && wave < 3 && prow < NB;
            const bool is_ident = !is_diag && wave == 3 && pl < 16 && want_inv;
            const T* src = S + (is_diag ? c0 + lane : (is_panel ? prow : c0)) * LD + c0;
            T row[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const T v = src[c];
                row[c] = (is_diag || is_panel) ? v : ((is_ident && c == pl) ? (T)1 : (T)0);
            }
            int first_bad = 16;
            if (ablate & 32) {        // round 2's chain: 1 / sqrt(pivot) between two pivots
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const T piv = bcast_lane(row[c], c);                    // wave-uniform
                    first_bad = (piv > (T)0) ? first_bad : min(first_bad, c);
                    const T rs = inv_sqrt(piv);
                    const T lrc = row[c] * rs;
                    row[c] = lrc;
#pragma unroll
                    for (int k = c + 1; k < 16; ++k) row[k] -= lrc * bcast_lane(lrc, k);
                }
            } else {
                // Square-root-free chain (L D L^T inside the micro-panel): between two pivots lie 1 / pivot (hardware seed + two
                // Newton steps: four dependent operations against the seven of 1 / sqrt), one multiply and one update; the sixteen
                // square roots follow afterwards, off the chain, as independent instruction streams.  Same numbers to rounding
                // (tools/micro/tallstep.hip: 4.4e-16 against a long-double Cholesky for both forms; 1.44 -> 1.04 us per step).
                //   u_rc = a_rc - sum_{j<c} u_rj u_cj / d_j,  d_c = u_cc,  L_rc = u_rc / sqrt(d_c)
                T piv[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    piv[c] = bcast_lane(row[c], c);                         // wave-uniform
                    first_bad = (piv[c] > (T)0) ? first_bad : min(first_bad, c);
                    const T w = row[c] * recip(piv[c]);
#pragma unroll
                    for (int k = c + 1; k < 16; ++k) row[k] -= row[c] * bcast_lane(w, k);
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) row[c] *= inv_sqrt(piv[c]);
            }
            if (first_bad < 16) {                                        // wave-uniform (piv is)
                if (wave == 0 && lane == 0) { fail = 1; atomicCAS(info, 0, col0 + c0 + first_bad + 1); }
            } else {
                // every lane that holds a real row writes it back whole: the four waves' copies of the diagonal rows are the same
                // bits (same instructions on the same data), and what a diagonal row carries right of the diagonal is masked
                // where L leaves the workgroup (nothing in LDS reads it) -- one store block instead of one per kind of lane
                if (is_diag || is_panel) {
                    T* P = S + (is_diag ? c0 + lane : prow) * LD + c0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) P[c] = row[c];
                } else if (is_ident) {
                    T* dst = Dinv + jb * 16 * DLD + pl;                  // column pl of D^-1
#pragma unroll
                    for (int c = 0; c < 16; ++c) dst[c * DLD] = row[c];
                }
            }
        } else if (jb > 0) {
            // ---- deferred part of the previous trailing update: tiles (ti, tj) with 1 <= tj <= ti; then block row
            // jb - 1 of the inverse (its diagonal inverse was formed by wave 3 in the previous step)
            const int pc0 = c0 - 16, pr0 = c0;
            const int pnt = (NB - pr0) / 16;
            const int nrest = pnt * (pnt - 1) / 2;
            const int nx = want_inv ? jb - 1 : 0;
            // last step: the wave that finishes X[6][q] goes on to the sum of X[7][q] (rows 0-6 of X and block row 7 of L
            // are final; only the last diagonal inverse, formed by wave 3 right now, is missing)
            const bool last = want_inv && jb == NB / 16 - 1;
            for (int t = wave - NAB; t < nrest + nx + (last ? 1 : 0); t += NWV - NAB) {
                if (last && t == nrest + nx) {
                    inv_tile_sum<T>(S, Dinv, jb, jb - 1, lane);          // q = 6: L[7][6] Dinv[6]
                    continue;
                }
                if (t < nrest) {
                    int u = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                    while (u * (u + 1) / 2 > t) --u;
                    while ((u + 1) * (u + 2) / 2 <= t) ++u;
                    const int ti = u + 1, tj = t - u * (u + 1) / 2 + 1;
                    lds_tile_mm<T, true>(S + (pr0 + ti * 16) * LD + pr0 + tj * 16, LD, S + (pr0 + ti * 16) * LD + pc0, LD,
                                         S + (pr0 + tj * 16) * LD + pc0, LD, 16, (T)-1, (T)1, lane);
                } else {
                    inv_tile<T>(S, Dinv, jb - 1, t - nrest, lane);
                    if (last) inv_tile_sum<T>(S, Dinv, jb, t - nrest, lane);
                }
            }
            // ---- then, still beside the tall-panel step (the longer half of this phase), these waves send out what the PREVIOUS
            // step made final: its 16 columns of L and block row jb - 2 of the inverse.  Nothing reads or writes those again
            // except the reads of later updates, so after the last step only that step's own columns and two block rows of the
            // inverse are left to store.  (Stores placed in the short update phase behind the barrier stretched every step.)
            if (prog) {
                store_l_panel(jb - 1, tid - 64 * NAB, NTH - 64 * NAB);
                if (want_inv && jb >= 2) store_inv_rows(16 * (jb - 2), 16 * (jb - 1), tid - 64 * NAB, NTH - 64 * NAB);
            }
        }
        __syncthreads();
        LTL(17 + 2 * jb);
        if (fail) return;
        if (prog && jb == 0) set_b_to_lds();     // columns 32-127: nothing has touched them yet; the next barrier publishes them
        if (r0 >= NB) break;
        const int nt = (NB - r0) / 16;
        // ---- first tile column of the trailing update: next diagonal block and next panel, T <- T - P P^T
        for (int t = wave; t < nt; t += NWV)
            lds_tile_mm<T, true>(S + (r0 + t * 16) * LD + r0, LD, S + (r0 + t * 16) * LD + c0, LD, S + r0 * LD + c0, LD, 16,
                                 (T)-1, (T)1, lane);
        __syncthreads();
        LTL(18 + 2 * jb);
    }

    if (!prog || (ablate & 1)) {
        for (int idx = tid; idx < NB * NB / 2; idx += NTH) {
            const int i = idx >> 6, k = (idx & 63) * 2;
            pair_t v = *reinterpret_cast<const pair_t*>(S + i * LD + k);
            if (k > i) v[0] = (T)0;
            if (k + 1 > i) v[1] = (T)0;
            if (WT) st_pair_wt(A, rA, (long)i * lda + k, v);
            else *reinterpret_cast<pair_t*>(A + (long)i * lda + k) = v;
        }
        if (!want_inv) return;
        store_inv_rows(0, 112, tid, NTH);
        if (ablate & 1) return;
        for (int q = wave; q < 7; q += NWV) inv_tile_finish<T>(S, Dinv, 7, q, lane);   // sums and Dinv[7] are in place (last step)
        __syncthreads();
        store_inv_rows(112, 128, tid, NTH);
        return;
    }
    // progressive form: the last step's columns of L, block row 6 of the inverse (final since the loop's last barrier) and, once
    // its seven products are done, block row 7
    if (want_inv && wave < 7) inv_tile_finish<T>(S, Dinv, 7, wave, lane);
    if (wave >= STW) {
        store_l_panel(NB / 16 - 1, tid - 64 * STW, NTH - 64 * STW);
        if (want_inv) store_inv_rows(96, 112, tid - 64 * STW, NTH - 64 * STW);
    }
    (void)STW;
    if (!want_inv) return;
    __syncthreads();
    LTL(34);
    store_inv_rows(112, 128, tid, NTH);
#undef LTL
}

// ------------------------------------------------------------------------------------------------
// Leaf, third form (round 3): the panel solve and the first update column leave the pivot loop.
// The second form runs every panel row through the register-resident pivot loop (a row per lane), writes the rows back to LDS,
// meets at a barrier and only then forms the next column block's update by MFMA -- a step is pivot loop (1.4 us) + LDS round trip of
// 128 rows (1.2) + update column (0.5) + two barriers = 3.4 us, stamped in-kernel.  Here ONE wave factors the 16 x 16 diagonal block
// alone (sixteen rows + sixteen identity rows: D and D^-1, square-root-free chain), and everything below it is two chained MFMA
// products per 16-row tile that never leave the accumulator layout:
//     X_t^T = D^-1 A_t^T          (A operand D^-1, B operand the tile as stored)        -> lane holds X_t[l & 15][k(r)], r = 0..3
//     T_t  -= X_t X_0^T           (A operand = those four registers, B operand = the same four of tile 0, recomputed per wave)
// because the C/D layout of the first product IS the A/B fragment layout of the second (k and the tile row swap roles).  The wave
// that owns tile 0 -- the next diagonal block -- goes straight on to factor it while the others finish the step (their tiles, then
// the deferred trailing update and the inverse's block row): per step the critical wave runs factor (1.0-1.4 us) + one tile's
// chain (0.4) + two barriers.
// ------------------------------------------------------------------------------------------------
// The third form's serial core as a function of its own (NOT inlined): inside the 168-VGPR / 106-SGPR leaf kernel the compiler
// had two scalar registers left for the loop's broadcasts (v_readlane -> s[0:1] -> v_fma, fifteen times per column through the
// same pair: 2.0 us per step); compiled on its own it pipelines them through as many pairs as it likes (1.04 us in isolation,
// tools/micro/tallstep.hip).  One wave: lanes 0-15 the block's rows, lanes 16-31 identity rows (they leave as D^-1).
// Returns the first non-positive pivot's column (16: none).
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) float lds_f32;
template <typename T> struct LdsPtr;
template <> struct LdsPtr<double> { typedef lds_f64* type; };
template <> struct LdsPtr<float> { typedef lds_f32* type; };
template <typename T>
__device__ __forceinline__ int leaf3_factor_diag(typename LdsPtr<T>::type D, typename LdsPtr<T>::type Dv, int lane, bool store) {
    const int fr = lane & 15;
    const bool is_diag = lane < 16, is_ident = lane >= 16 && lane < 32;
    T row[16], piv[16];
    // every lane loads its (fr-th) row unconditionally, all sixteen values in flight, and only then selects: written as
    // `is_diag ? D[..] : ident` the compiler put each load behind its own exec-mask branch (sixteen serial LDS round trips)
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = D[fr * LD + c];
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(row[c]));
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = is_diag ? row[c] : ((is_ident && c == fr) ? (T)1 : (T)0);
    int first_bad = 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        piv[c] = bcast_lane(row[c], c);                         // wave-uniform
        first_bad = (piv[c] > (T)0) ? first_bad : min(first_bad, c);
        const T w = row[c] * recip(piv[c]);
#pragma unroll
        for (int k = c + 1; k < 16; ++k) row[k] -= row[c] * bcast_lane(w, k);
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] *= inv_sqrt(piv[c]);
    if (first_bad == 16 && store) {
        if (is_diag) {
#pragma unroll
            for (int c = 0; c < 16; ++c) D[fr * LD + c] = row[c];      // (right of the diagonal: masked where L leaves the workgroup)
        } else if (is_ident) {
#pragma unroll
            for (int c = 0; c < 16; ++c) Dv[c * DLD + fr] = row[c];    // column fr of D^-1
        }
    }
    return first_bad;
}

// The same factor with its rank-4 trailing updates on the matrix pipe (fp64; round 3, second half).  The row-per-lane form above is
// ISSUE-bound, not latency-bound: about 1250 instructions for one wave at four clocks each = 5000 shader clocks = 2.1 us, whatever runs
// beside it (tools/micro/blockfac.hip, tools/micro/issue_rate.hip; the instruction cache is not it: SQC_ICACHE_MISSES = 6 per leaf).
// Here the block lives in MFMA accumulator layout: by symmetry register r of lane (g, i) -- C[g + 4r][i] of v_mfma_f64_16x16x4 -- is
// read as T[i][4g + r], so lane group g holds four COLUMNS of every row.  Four panels of four columns: the group that holds the panel
// eliminates it with broadcasts inside the panel only (six multipliers instead of up to fifteen per column), hands -w, a and the
// identity rows' panel through a small LDS staging area, and the rank-4 update of everything right of the panel is ONE MFMA (plus one
// for the identity rows that leave as D^-1).  MFMA row rho = g + 4r stands for matrix column 4g + r, so the A operand's lane (k, rho)
// takes the multiplier of matrix row 4 (rho & 3) + (rho >> 2); rows of finished columns get a zero multiplier and keep their values.
// About 570 instructions, 3400 clocks in isolation (5000 for the form above), same numbers to the last bit or two (4.4e-16 against a
// long-double Cholesky either way).
#define SLOT_STORES_FROM 8   // blocked form: the previous step's panel / block row is stored by the solve phase's tile-less waves (8: always; from step 3 on by the slot's workers instead: measured equal, the work only moves)
#define F3_SLD 6          // staging rows of four doubles padded to six: 16-byte aligned writes, conflict-free 8-byte reads
#define F3_STAGE (48 * F3_SLD + 16)
__device__ __forceinline__ int leaf3_factor_blk(lds_f64* D, lds_f64* Dv, lds_f64* stage, int lane, bool store) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) d2 lds_d2;
    const int g = lane >> 4, i = lane & 15;
    lds_f64* Sw = stage;                    // [16][F3_SLD]  -w of the panel (A operand)
    lds_f64* Sa = stage + 16 * F3_SLD;      // [16][F3_SLD]  the panel's unscaled columns (B operand of T's update)
    lds_f64* Sz = stage + 32 * F3_SLD;      // [16][F3_SLD]  the identity rows' panel (B operand of Z's update)
    lds_f64* Sp = stage + 48 * F3_SLD;      // [16] pivots
    pg_d4 t, z;
    {
        const d2 lo = *reinterpret_cast<lds_d2*>(D + i * LD + 4 * g), hi = *reinterpret_cast<lds_d2*>(D + i * LD + 4 * g + 2);
        t[0] = lo[0]; t[1] = lo[1]; t[2] = hi[0]; t[3] = hi[1];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = (4 * g + r == i) ? 1.0 : 0.0;
    const int jrow = 4 * (i & 3) + (i >> 2);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (g == p) {                       // the group that holds this panel's columns; the others wait for the update
            double wn[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double piv = bcast_lane(t[c], 16 * p + 4 * p + c);
                wn[c] = -t[c] * recip(piv);
#pragma unroll
                for (int k = c + 1; k < 4; ++k) {
                    const double m = bcast_lane(wn[c], 16 * p + 4 * p + k);
                    t[k] = __builtin_fma(t[c], m, t[k]);
                    z[k] = __builtin_fma(z[c], m, z[k]);
                }
            }
            if (p < 3) {
                *reinterpret_cast<lds_d2*>(Sw + i * F3_SLD) = d2{wn[0], wn[1]};
                *reinterpret_cast<lds_d2*>(Sw + i * F3_SLD + 2) = d2{wn[2], wn[3]};
                *reinterpret_cast<lds_d2*>(Sa + i * F3_SLD) = d2{t[0], t[1]};
                *reinterpret_cast<lds_d2*>(Sa + i * F3_SLD + 2) = d2{t[2], t[3]};
                *reinterpret_cast<lds_d2*>(Sz + i * F3_SLD) = d2{z[0], z[1]};
                *reinterpret_cast<lds_d2*>(Sz + i * F3_SLD + 2) = d2{z[2], z[3]};
            }
        }
        if (p == 3) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double aw = Sw[jrow * F3_SLD + g];
        double bt = Sa[i * F3_SLD + g];
        const double bz = Sz[i * F3_SLD + g];
        aw = (jrow >= 4 * p + 4) ? aw : 0.0;
        // what lies right of the diagonal in the block is scratch (finite for finite input); a NaN there must not reach a finished
        // entry through 0 x NaN (the LAPACK convention names the FIRST bad minor: test_kernel_build_propagates_nan)
        bt = (i >= 4 * p + g) ? bt : 0.0;
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, bt, t, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, bz, z, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();    // the staging area is re-used by the next panel
    }
    {   // pivots = the diagonal: lane (g, i) with i >> 2 == g holds T[i][i] in register i & 3
        double d = t[0];
        d = ((i & 3) == 1) ? t[1] : d;
        d = ((i & 3) == 2) ? t[2] : d;
        d = ((i & 3) == 3) ? t[3] : d;
        if ((i >> 2) == g) Sp[i] = d;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const d2 p01 = *reinterpret_cast<lds_d2*>(Sp + 4 * g), p23 = *reinterpret_cast<lds_d2*>(Sp + 4 * g + 2);
    const bool bad = !(p01[0] > 0.0) || !(p01[1] > 0.0) || !(p23[0] > 0.0) || !(p23[1] > 0.0);
    if (__builtin_amdgcn_ballot_w64(bad) != 0) {   // a non-positive (or NaN) pivot: name the first one's column, store nothing
        int first_bad = 16;
#pragma unroll 1
        for (int c = 15; c >= 0; --c)
            if (!(Sp[c] > 0.0)) first_bad = c;
        return first_bad;
    }
    if (store) {
        const double rs0 = inv_sqrt(p01[0]), rs1 = inv_sqrt(p01[1]), rs2 = inv_sqrt(p23[0]), rs3 = inv_sqrt(p23[1]);
        *reinterpret_cast<lds_d2*>(D + i * LD + 4 * g) = d2{t[0] * rs0, t[1] * rs1};
        *reinterpret_cast<lds_d2*>(D + i * LD + 4 * g + 2) = d2{t[2] * rs2, t[3] * rs3};
        Dv[(4 * g + 0) * DLD + i] = z[0] * rs0;
        Dv[(4 * g + 1) * DLD + i] = z[1] * rs1;
        Dv[(4 * g + 2) * DLD + i] = z[2] * rs2;
        Dv[(4 * g + 3) * DLD + i] = z[3] * rs3;
    }
    return 16;
}

// (Round 3 also tried the chain as a LOOP -- registers rotate instead of the code: row[k-1] = row[k] - row[0] w_k, the pivot always in
// row[0], its column a run-time lane of v_readlane; two loops of eight pivots, 0.9 KB of code instead of 9 KB, bit-identical factor.
// A step took 4.6 us instead of 2.9 (n = 4096: 1.54 -> 1.85 ms): every iteration pays all fifteen broadcasts, and the wave issues
// about one instruction per 7 clocks whether the code is short or long.  Removed again; DESIGN.md has the numbers.)
template <typename T, bool WT, bool BLK = false>
__device__ __forceinline__ void leaf3_body(char* smem_raw, T* __restrict__ A, long lda, T* __restrict__ inv, long ldi,
                                           int* __restrict__ info, int col0, long long* tlog = nullptr, int nf3 = 1,
                                           int* early = nullptr) {
#define LTL(i) do { if (tlog && threadIdx.x == 0) tlog[i] = wall_clock64(); } while (0)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(A, 0, (int)((127 * lda + 128) * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(inv, 0, inv ? (int)((127 * ldi + 128) * sizeof(T)) : 0, 0x00020000);
    T* S = reinterpret_cast<T*>(smem_raw);
    T* Dinv = S + NB * LD;                                  // [8][16][DLD]
    int& fail = *reinterpret_cast<int*>(Dinv + 8 * 16 * DLD);
    int* ecnt = &fail + 1;                                  // waves that have drained their early stores
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar branches and index arithmetic
    const int fr = lane & 15, fk = lane >> 4;
    typedef typename Mfma<T>::acc_t acc_t;
    typedef T pair_t __attribute__((ext_vector_type(2)));

    if (*info != 0) {
        if (early && tid == 0) __hip_atomic_store(early, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody may wait for it
        return;
    }
    if (tid == 0) { fail = 0; *ecnt = 0; ecnt[1] = 0; }
    {   // lower triangle -> LDS, two columns per thread, 64 pairs per row; all loads are issued before the first use
        constexpr int NLD = (NB * NB / 2 + NTH - 1) / NTH;
        pair_t v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + u * NTH, i = idx >> 6, k = (idx & 63) * 2;
            v[u] = pair_t{(T)0, (T)0};
            if (idx < NB * NB / 2 && k <= i) {
                if (WT) v[u] = ld_pair_wt(A, rA, (long)i * lda + k);
                else v[u] = *reinterpret_cast<const pair_t*>(A + (long)i * lda + k);
            }
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + u * NTH, i = idx >> 6, k = (idx & 63) * 2;
            if (k + 1 > i) v[u][1] = (T)0;
            if (idx < NB * NB / 2) *reinterpret_cast<pair_t*>(S + i * LD + k) = v[u];
        }
    }
    __syncthreads();
    LTL(16);
    const bool want_inv = inv != nullptr;
    // 16 columns of L (all 128 rows: zeros above the diagonal) / rows of the inverse -> global, by the threads t0, t0 + nth, ...
    auto store_l_panel = [&](int jbp, int t0, int nth) {
        const int cc0 = 16 * jbp;
        if (BLK) {   // rows from the diagonal block down (the zeros above it are out already)
#pragma unroll 2
            for (int idx = t0; idx < (NB - cc0) * 8; idx += nth) {
                const int i = cc0 + (idx >> 3), k = cc0 + (idx & 7) * 2;
                pair_t v = *reinterpret_cast<const pair_t*>(S + i * LD + k);
                if (k > i) v[0] = (T)0;
                if (k + 1 > i) v[1] = (T)0;
                if (WT) st_pair_wt(A, rA, (long)i * lda + k, v);
                else *reinterpret_cast<pair_t*>(A + (long)i * lda + k) = v;
            }
            return;
        }
        for (int idx = t0; idx < NB * 8; idx += nth) {
            const int i = idx >> 3, k = cc0 + (idx & 7) * 2;
            pair_t v = {(T)0, (T)0};
            if (i >= cc0) v = *reinterpret_cast<const pair_t*>(S + i * LD + k);
            if (k > i) v[0] = (T)0;
            if (k + 1 > i) v[1] = (T)0;
            if (WT) st_pair_wt(A, rA, (long)i * lda + k, v);
            else *reinterpret_cast<pair_t*>(A + (long)i * lda + k) = v;
        }
    };
    auto store_inv_rows = [&](int row_lo, int row_hi, int t0, int nth) {
        if (BLK) {   // one block row (row_hi == row_lo + 16): its pb + 1 column blocks, eight pairs x sixteen rows each
            const int pb = row_lo >> 4;
#pragma unroll 2
            for (int idx = t0; idx < 128 * (pb + 1); idx += nth) {
                const int qb = idx >> 7, ii = (idx >> 3) & 15, kk = (idx & 7) * 2, i = row_lo + ii, k = 16 * qb + kk;
                pair_t v;
                if (qb == pb) {
                    const T* Dv = Dinv + pb * 16 * DLD + ii * DLD + kk;
                    v[0] = (kk <= ii) ? Dv[0] : (T)0;
                    v[1] = (kk + 1 <= ii) ? Dv[1] : (T)0;
                } else {
                    const T* Xt = S + (16 * qb + ii) * LD + 16 * pb + kk;
                    v[0] = Xt[0];
                    v[1] = Xt[1];
                }
                if (WT) st_pair_wt(inv, rI, (long)i * ldi + k, v);
                else *reinterpret_cast<pair_t*>(inv + (long)i * ldi + k) = v;
            }
            return;
        }
        for (int idx = row_lo * 64 + t0; idx < row_hi * 64; idx += nth) {
            const int i = idx >> 6, k = (idx & 63) * 2;
            const int pb = i >> 4, qb = k >> 4, ii = i & 15;
            pair_t v = {(T)0, (T)0};
            if (qb == pb) {
                const T* Dv = Dinv + pb * 16 * DLD + ii * DLD + (k & 15);
                v[0] = (k <= i) ? Dv[0] : (T)0;
                v[1] = (k + 1 <= i) ? Dv[1] : (T)0;
            } else if (qb < pb) {
                const T* Xt = S + (16 * qb + ii) * LD + 16 * pb + (k & 15);
                v[0] = Xt[0];
                v[1] = Xt[1];
            }
            if (WT) st_pair_wt(inv, rI, (long)i * ldi + k, v);
            else *reinterpret_cast<pair_t*>(inv + (long)i * ldi + k) = v;
        }
    };
    const int NF3 = nf3 & 15;                               // waves that run the factor's instructions (wave 0 for real)
    // bit 4 (default; PG_LEAF3_ALONE=0 clears it): the waves that share wave 0's SIMD (4, 8: waves go round the four SIMDs) sit out the
    // slot beside the factor: their MFMA / LDS work took issue cycles from the pivot chain (a step 2.9 -> 2.2-2.7 us; n = 4096
    // 1.535 -> 1.475 ms, the nine remaining workers finish within the factor's time except in the last slot)
    const bool alone = (nf3 & 16) != 0;
    const bool worker = alone ? (wave & 3) != 0 : wave >= NF3;
    const int widx = alone ? (wave >> 2) * 3 + (wave & 3) - 1 : wave - NF3, nwork = alone ? (NWV / 4) * 3 : NWV - NF3;
    acc_t x0;                                               // tile 0's X in accumulator layout (every wave's own copy)
    for (int jb = 0; jb < NB / 16; ++jb) {
        const int c0 = jb * 16, r0 = c0 + 16;
        if (BLK && jb == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero stores of block rows 0..3 are out before the early flag can be raised
        if (wave < ((BLK && sizeof(T) == 8) ? 1 : NF3)) {
            // ---- F: the diagonal block alone.  lanes 0-15: its rows; lanes 16-31: identity rows, which leave the loop as D^-1.
            // (PG_LEAF3_NF = 2..4: waves 1 .. NF3-1 run the SAME instructions on the same data and store nothing -- an experiment on
            // whether company on the other SIMDs shares the instruction fetch of this 9 KB of straight-line code: it does not,
            // the factor takes 5100-5400 clocks either way; default 1)
            int first_bad;
            if constexpr (BLK && sizeof(T) == 8)
                first_bad = leaf3_factor_blk((lds_f64*)(S + c0 * LD + c0), (lds_f64*)(Dinv + jb * 16 * DLD), (lds_f64*)(Dinv + 8 * 16 * DLD + 18), lane, true);
            else
                first_bad = leaf3_factor_diag<T>((typename LdsPtr<T>::type)(S + c0 * LD + c0), (typename LdsPtr<T>::type)(Dinv + jb * 16 * DLD),
                                                 lane, wave == 0);
            // (s_setprio 3 around the factor: no change, n = 4096 1.473 vs 1.480 ms)
            if (wave == 0 && lane == 0) {
                if (tlog) tlog[35 + jb] = wall_clock64();                                  // the factor alone (probe_cs_tlog.py)
                if (nf3 & 64) __hip_atomic_store(ecnt + 1, jb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (wave == 0 && first_bad < 16 && lane == 0) { fail = 1; atomicCAS(info, 0, col0 + c0 + first_bad + 1); }
        }
        __syncthreads();                                               // B: D and D^-1 of this step are published
        LTL(17 + 2 * jb);
        if (fail) return;
        if (r0 >= NB) break;
        const int nt = (NB - r0) / 16;
        // ---- M: panel solve + first update column, one 16-row tile per wave, all in MFMA fragments
        if (wave < nt) {
            const int t = wave;
            T df[4], a0[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                df[q] = Dinv[jb * 16 * DLD + fr * DLD + 4 * q + fk];            // A operand: D^-1[fr][4q + fk]
                a0[q] = S[(r0 + fr) * LD + c0 + 4 * q + fk];                    // B operand: A_0[fr][4q + fk] (= A_0^T[4q + fk][fr])
            }
            acc_t ct;                                                            // T_t in C layout
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x0[r] = (T)0;
                ct[r] = S[(r0 + 16 * t + Mfma<T>::row(lane, r)) * LD + r0 + fr];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) x0 = Mfma<T>::run(df[q], a0[q], x0);    // X_0^T: lane holds X_0[fr][row(lane, r)]
            acc_t xt = x0;
            if (t > 0) {
                T at[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) at[q] = S[(r0 + 16 * t + fr) * LD + c0 + 4 * q + fk];
#pragma unroll
                for (int r = 0; r < 4; ++r) xt[r] = (T)0;
#pragma unroll
                for (int q = 0; q < 4; ++q) xt = Mfma<T>::run(df[q], at[q], xt);
            }
            acc_t u;
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = (T)0;
#pragma unroll
            for (int r = 0; r < 4; ++r) u = Mfma<T>::run(xt[r], x0[r], u);      // sum over k = row(lane, r): X_t[i][k] X_0[j][k]
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(r0 + 16 * t + Mfma<T>::row(lane, r)) * LD + r0 + fr] = ct[r] - u[r];
            if (t > 0 || nt == 1) {                                              // the panel's rows of L (tile 0: see below)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(r0 + 16 * t + fr) * LD + c0 + Mfma<T>::row(lane, r)] = xt[r];
            }
        } else if (BLK && jb >= 1 && jb < SLOT_STORES_FROM) {
            // ---- the waves without a tile (5 + jb of them): what the PREVIOUS step made final goes out -- its sixteen columns of L and
            // block row jb - 1 of the inverse.  (With the row-per-lane factor these stores sat in the slot beside the factor, whose nine
            // workers had time to spare; beside the blocked factor that slot is what the step waits for: tiles 1.3 us + stores 0.6 us
            // against 1.5 us of factor, stamped.)  After step 4's M the inverse's first four block rows are out: the storing waves drain,
            // count themselves in LDS, and the last one raises *early (chainstep.hip's two-phase hand-over).
            const int nidle = NWV - nt, t0 = 64 * (wave - nt) + lane, nth = 64 * nidle;
            store_l_panel(jb - 1, t0, nth);
            if (want_inv) store_inv_rows(16 * (jb - 1), 16 * jb, t0, nth);
            if (early && jb == 4) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    const int before = atomicAdd(ecnt, 1);                         // LDS
                    if (before == nidle - 1) __hip_atomic_store(early, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();                                               // B': every tile's X and updated column are in LDS
        LTL(18 + 2 * jb);
        if (wave == 0) {
            // tile 0's rows of L go out only now: the other waves read the unsolved tile 0 during M (with a single tile nobody
            // does, and the inverse's last block row -- formed in the slot below -- already reads these rows: stored in M then)
            if (nt > 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(r0 + fr) * LD + c0 + Mfma<T>::row(lane, r)] = x0[r];
            }
        } else if (worker) {
            // (bit 6, PG_LEAF3_DIAG=1: timing diagnostic -- the slot's work starts only when the factor beside it has finished, so the
            // stamps give both durations without the other's interference)
            if (nf3 & 64) {
                while (__hip_atomic_load(ecnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < jb + 2) __builtin_amdgcn_s_sleep(2);
            }
            // ---- beside the next step's factor: the rest of this step's trailing update (tiles (ti, tj), 1 <= tj <= ti), block row
            // jb of the inverse (its D^-1 exists since B), and -- in the last such slot -- the sums of the inverse's last block row
            const bool wstamp = tlog && wave == 1 && lane == 0 && jb == 2;       // one worker's slot, stamped (probe_cs_tlog.py)
            if (wstamp) tlog[43] = wall_clock64();
            const int nrest = nt * (nt - 1) / 2;
            const int nx = want_inv ? jb : 0;
            const bool last = want_inv && jb == NB / 16 - 2;           // after it only the last diagonal block is factored
            for (int w = widx; w < nrest + nx + (last ? 1 : 0); w += nwork) {
                if (last && w == nrest + nx) {
                    inv_tile_sum<T>(S, Dinv, jb + 1, jb, lane);        // q = 6: L[7][6] Dinv[6]
                    continue;
                }
                if (w < nrest) {
                    int uu = 0;                                        // w = uu (uu + 1) / 2 + (tj - 1), uu < 6: scalar arithmetic
                    while ((uu + 1) * (uu + 2) / 2 <= w) ++uu;
                    const int ti = uu + 1, tj = w - uu * (uu + 1) / 2 + 1;
                    if (BLK) lds_tile_update16<T>(S + (r0 + ti * 16) * LD + r0 + tj * 16, S + (r0 + ti * 16) * LD + c0, S + (r0 + tj * 16) * LD + c0, lane);
                    else lds_tile_mm<T, true>(S + (r0 + ti * 16) * LD + r0 + tj * 16, LD, S + (r0 + ti * 16) * LD + c0, LD,
                                              S + (r0 + tj * 16) * LD + c0, LD, 16, (T)-1, (T)1, lane);
                } else {
                    inv_tile<T>(S, Dinv, jb, w - nrest, lane);
                    if (last) inv_tile_sum<T>(S, Dinv, jb + 1, w - nrest, lane);
                }
            }
            // ---- still beside the factor (these waves idle for most of it): what the PREVIOUS step made final goes out -- its
            // sixteen columns of L and block row jb - 1 of the inverse -- so that the tail stores two panels and two block rows
            // instead of everything.  After step 4's slot the inverse's first four block rows are out: each storing wave drains,
            // counts itself in LDS, and the last one raises *early -- the rows of block row k+1 start their solve against
            // those 64 rows while the leaf still has three steps and its tail to run (chainstep.hip).
            if (wstamp) tlog[46] = wall_clock64();
            if (BLK) {
                // the zeros above the diagonal of both output tiles, block row jb's share per slot: the panel / block-row stores then only
                // carry what is on or below the diagonal (1152 pairs per step instead of 2048).  (All of them beside the FIRST factor
                // took one CU's store path 1.7 us and slowed that factor; in the load phase they held the tile's way into LDS behind
                // their write acknowledgements: load 2.0 -> 4-5 us.)
                const pair_t zz = {(T)0, (T)0};
                for (int idx = 64 * widx + lane; idx < 16 * 64; idx += 64 * nwork) {
                    const int i = 16 * jb + (idx >> 6), k = (idx & 63) * 2;
                    if (k > i) {
                        if (WT) st_pair_wt(A, rA, (long)i * lda + k, zz);
                        else *reinterpret_cast<pair_t*>(A + (long)i * lda + k) = zz;
                        if (inv) {
                            if (WT) st_pair_wt(inv, rI, (long)i * ldi + k, zz);
                            else *reinterpret_cast<pair_t*>(inv + (long)i * ldi + k) = zz;
                        }
                    }
                }
            }
            if ((!BLK && jb >= 1) || (BLK && jb >= SLOT_STORES_FROM)) {   // (blocked form: from step 3 on the slot has few tiles left and time to spare)
                const int t0 = 64 * widx + lane, nth = 64 * nwork;
                store_l_panel(jb - 1, t0, nth);
                if (want_inv) store_inv_rows(16 * (jb - 1), 16 * jb, t0, nth);
                if (wstamp) tlog[47] = wall_clock64();
                if (early && jb == 4) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) {
                        const int before = atomicAdd(ecnt, 1);                     // LDS
                        if (before == nwork - 1) __hip_atomic_store(early, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
    }
    // the loop ends behind B of the last step: L is complete; of the inverse only the last block row's final products are missing.
    // Panels 0-5 of L and block rows 0-5 of the inverse went out while the loop ran.
    if (want_inv) {
        for (int q = wave; q < 7; q += NWV) inv_tile_finish<T>(S, Dinv, 7, q, lane);
    }
    if (BLK) {   // the last block row's zeros above the diagonal (the slots wrote those of block rows 0..6)
        const pair_t zz = {(T)0, (T)0};
        for (int idx = tid; idx < 16 * 64; idx += NTH) {
            const int i = NB - 16 + (idx >> 6), k = (idx & 63) * 2;
            if (k > i) {
                if (WT) st_pair_wt(A, rA, (long)i * lda + k, zz);
                else *reinterpret_cast<pair_t*>(A + (long)i * lda + k) = zz;
                if (inv) {
                    if (WT) st_pair_wt(inv, rI, (long)i * ldi + k, zz);
                    else *reinterpret_cast<pair_t*>(inv + (long)i * ldi + k) = zz;
                }
            }
        }
    }
    store_l_panel(6, tid, NTH);
    store_l_panel(7, tid, NTH);
    if (!want_inv) return;
    store_inv_rows(96, 112, tid, NTH);
    __syncthreads();
    LTL(34);
    store_inv_rows(112, 128, tid, NTH);
#undef LTL
}

template <typename T>
__global__ __launch_bounds__(NTH) void pg_leaf2_kernel(T* __restrict__ A, long lda, T* __restrict__ inv, long ldi,
                                                       int* __restrict__ info, int col0, int ablate, long eA, long eInv) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // batched experts: one workgroup each, one info word each
    leaf2_body<T, false>(smem_raw, A + blockIdx.x * eA, lda, inv ? inv + blockIdx.x * eInv : nullptr, ldi, info + blockIdx.x, col0, ablate);
}

// The leaf of the flag-coupled chain (chainstep.hip): resident before its tile exists.  Waits until the `want` workgroups that
// own the tile have published it (*ready; they store it write-through), loads it past this CU's L1, factors, stores L and the
// inverse write-through, drains, and sets *done: no fence on either side (every handed-off byte is an sc1 store read by sc1
// loads, or read behind the reader's own acquire).
// *done is set on every path (bad pivot, earlier failure, timeout): the rows below wait for it.
template <typename T, bool WT>
__global__ __launch_bounds__(NTH) void pg_leaf2s_kernel(T* __restrict__ A, long lda, T* __restrict__ inv, int* __restrict__ info,
                                                        int col0, int* ready, int want, int* done, CsWait tmo, long long* tlog, int ablate) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (tlog && threadIdx.x == 0) tlog[0] = wall_clock64();
    if (threadIdx.x == 0) {
        if (!cs_spin_ge(ready, want, tmo)) atomicCAS(info, 0, -1);
        if (!WT) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (tlog && threadIdx.x == 0) tlog[1] = wall_clock64();
    leaf2_body<T, WT>(smem_raw, A, lda, inv, NB, info, col0, ablate, tlog);
    if (tlog && threadIdx.x == 0) tlog[2] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tlog && threadIdx.x == 0) tlog[3] = wall_clock64();
    if (threadIdx.x == 0) {
        if (!WT) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_store(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The third form as kernels of its own: sharing a kernel with the second form (a run-time switch between two inlined bodies) left
// the pivot loop two scalar registers for its broadcasts.
template <typename T, bool WT, bool BLK = false>
__global__ __launch_bounds__(NTH) void pg_leaf3s_kernel(T* __restrict__ A, long lda, T* __restrict__ inv, int* __restrict__ info,
                                                        int col0, int* ready, int want, int* done, CsWait tmo, long long* tlog, int nf3,
                                                        int* early, CsBatch cb) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (cb.nexp > 1) {      // experts together: blockIdx.x = expert
        const long e = blockIdx.x;
        A += e * cb.eA; inv += e * cb.eInv; info += e;
        ready += e * cb.eF; done += e * cb.eF; tmo.tmo += e * cb.eF;
        if (early) early += e * cb.eF;
        tlog = nullptr;
    }
    if (tlog && threadIdx.x == 0) tlog[0] = wall_clock64();
    if (threadIdx.x == 0) {
        if (!cs_spin_ge(ready, want, tmo)) atomicCAS(info, 0, -1);
        if (!WT) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (tlog && threadIdx.x == 0) { tlog[1] = wall_clock64(); tlog[44] = (long long)__builtin_amdgcn_s_memtime(); }
    leaf3_body<T, WT, BLK>(smem_raw, A, lda, inv, NB, info, col0, tlog, nf3, WT ? early : nullptr);
    if (tlog && threadIdx.x == 0) { tlog[2] = wall_clock64(); tlog[45] = (long long)__builtin_amdgcn_s_memtime(); }   // [45]-[44] over [2]-[1]: the shader clock the leaf ran at
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tlog && threadIdx.x == 0) tlog[3] = wall_clock64();
    if (threadIdx.x == 0) {
        if (!WT) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (early) __hip_atomic_store(early, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every path: nobody waits for ever
        __hip_atomic_store(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <typename T, bool BLK = false>
__global__ __launch_bounds__(NTH) void pg_leaf3_kernel(T* __restrict__ A, long lda, T* __restrict__ inv, long ldi, int* __restrict__ info,
                                                       int col0, long eA, long eInv, int nf3) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    leaf3_body<T, false, BLK>(smem_raw, A + blockIdx.x * eA, lda, inv ? inv + blockIdx.x * eInv : nullptr, ldi, info + blockIdx.x, col0, nullptr, nf3);
}

static bool pg_leaf3_blk() {
    static const bool v = !(getenv("PG_LEAF3_BLK") && !atoi(getenv("PG_LEAF3_BLK")));
    return v;
}
bool pg_leaf_has_early() {   // the coupled leaf raises its early flag before its done flag (third form, write-through hand-off)
    static const bool v = !(getenv("PG_LEAF3") && !atoi(getenv("PG_LEAF3"))) && !(getenv("PG_CS_LEAF_WT") && !atoi(getenv("PG_CS_LEAF_WT")));
    return v;
}
template <typename T> int pg_leaf_sync(hipStream_t st, T* A, long lda, T* inv, int* info, int col0, int* ready, int want, int* done,
                                       const CsWait& tmo, int* early, const CsBatch* cbp) {
    const CsBatch cb = cbp ? *cbp : CsBatch{1, 0, 0, 0};
    const size_t lds = (size_t)(NB * LD + 8 * 16 * DLD + 2 + 16 + F3_STAGE) * sizeof(T) + 16;
    static bool attr_done = false;
    if (!attr_done) {
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf2s_kernel<T, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf2s_kernel<T, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf3s_kernel<T, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf3s_kernel<T, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf3s_kernel<T, true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf3s_kernel<T, false, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    static const int wt = getenv("PG_CS_LEAF_WT") ? atoi(getenv("PG_CS_LEAF_WT")) : 1;
    // Round 3 built two variants of the leaf's body and measured them against round 2's (same box, tools/probe_potrf_only.py and
    // the in-kernel stamps of tools/probe_cs_tlog.py); round 2's form stays the default, the variants are kept behind switches:
    //   PG_LEAF_PROG=1  tile loaded in two sets (columns 0-31 first), L columns and inverse block rows stored while the loop runs:
    //                   load 2.2 -> 1.5 us, but the stores' LDS reads stretch every step by 0.1-0.3 us: n = 4096 1.63 -> 1.65 ms
    //   PG_LEAF_LDL=1   square-root-free pivot chain (1.44 -> 1.04 us per 16-column step in isolation, tools/micro/tallstep.hip),
    //                   no gain inside the kernel: a step's 2.9-3.5 us are the chain (1.4), the rows' LDS round trip (1.2) and
    //                   two barriers, and the shorter chain only moves the waiting: n = 4096 1.63 -> 1.69 ms
    static const int abl = ((getenv("PG_LEAF_PROG") && atoi(getenv("PG_LEAF_PROG"))) ? 0 : 16) |
                           ((getenv("PG_LEAF_LDL") && atoi(getenv("PG_LEAF_LDL"))) ? 0 : 32) |
                           ((getenv("PG_LEAF3") && !atoi(getenv("PG_LEAF3"))) ? 0 : 64);           // third form (default); PG_LEAF3=0: second
    long long* tl = getenv("PG_CS_TLOG") ? reinterpret_cast<long long*>(tmo.tmo) + 512 + 48 * (col0 / NB) : nullptr;
    if (abl & 64) {
        static const int nf3 = (getenv("PG_LEAF3_NF") ? std::max(1, std::min(4, atoi(getenv("PG_LEAF3_NF")))) : 1) |
                               ((getenv("PG_LEAF3_ALONE") && !atoi(getenv("PG_LEAF3_ALONE"))) ? 0 : 16) |
                               ((getenv("PG_LEAF3_DIAG") && atoi(getenv("PG_LEAF3_DIAG"))) ? 64 : 0);
        // blocked diagonal factor (leaf3_factor_blk: fp64, default; PG_LEAF3_BLK=0: the row-per-lane factor) -- a kernel of its own per form
        if (pg_leaf3_blk() && sizeof(T) == 8) {
            if (wt) hipLaunchKernelGGL((pg_leaf3s_kernel<T, true, true>), dim3(cb.nexp), dim3(NTH), lds, st, A, lda, inv, info, col0, ready, want, done, tmo, tl, nf3, early, cb);
            else hipLaunchKernelGGL((pg_leaf3s_kernel<T, false, true>), dim3(cb.nexp), dim3(NTH), lds, st, A, lda, inv, info, col0, ready, want, done, tmo, tl, nf3, early, cb);
        } else if (wt) hipLaunchKernelGGL((pg_leaf3s_kernel<T, true>), dim3(cb.nexp), dim3(NTH), lds, st, A, lda, inv, info, col0, ready, want, done, tmo, tl, nf3, early, cb);
        else hipLaunchKernelGGL((pg_leaf3s_kernel<T, false>), dim3(cb.nexp), dim3(NTH), lds, st, A, lda, inv, info, col0, ready, want, done, tmo, tl, nf3, early, cb);
        PG_CHECK(hipGetLastError());
        return 0;
    }
    if (cb.nexp > 1) { pg_set_error("pg_leaf_sync: the second leaf form (PG_LEAF3=0) does not batch experts"); return -2; }
    if (wt) hipLaunchKernelGGL((pg_leaf2s_kernel<T, true>), dim3(1), dim3(NTH), lds, st, A, lda, inv, info, col0, ready, want, done, tmo,
                               getenv("PG_CS_TLOG") ? reinterpret_cast<long long*>(tmo.tmo) + 512 + 48 * (col0 / NB) : nullptr, abl);
    else hipLaunchKernelGGL((pg_leaf2s_kernel<T, false>), dim3(1), dim3(NTH), lds, st, A, lda, inv, info, col0, ready, want, done, tmo,
                               getenv("PG_CS_TLOG") ? reinterpret_cast<long long*>(tmo.tmo) + 512 + 48 * (col0 / NB) : nullptr, abl);
    PG_CHECK(hipGetLastError());
    return 0;
}
template int pg_leaf_sync<double>(hipStream_t, double*, long, double*, int*, int, int*, int, int*, const CsWait&, int*, const CsBatch*);
template int pg_leaf_sync<float>(hipStream_t, float*, long, float*, int*, int, int*, int, int*, const CsWait&, int*, const CsBatch*);

template <typename T> int pg_leaf(hipStream_t st, T* A, long lda, T* inv, long ldi, int* info, int col0, int ablate, int nexp, long eA,
                                  long eInv) {
    const size_t lds = (size_t)(NB * LD + 8 * 16 * DLD + 2 + 16 + F3_STAGE) * sizeof(T) + 16;
    static bool attr_done = false;
    static const int form = getenv("PG_LEAF") ? atoi(getenv("PG_LEAF")) : 2;   // 1: round-1 leaf (A / B / C phases), 2: fused tall-panel step
    if (!(getenv("PG_LEAF_PROG") && atoi(getenv("PG_LEAF_PROG")))) ablate ^= 16;     // default: round 2's data movement (bit 4 set);
    if (!(getenv("PG_LEAF_LDL") && atoi(getenv("PG_LEAF_LDL")))) ablate ^= 32;       // a caller's bit asks for the other form
    if (!(getenv("PG_LEAF3") && !atoi(getenv("PG_LEAF3")))) ablate ^= 64;           // third form unless PG_LEAF3=0 (or the caller's bit 6)
    if (!attr_done) {
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf2_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf3_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_leaf3_kernel<T, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    if ((ablate & 64) && form != 1 && !(ablate & 15)) {      // third form (no ablation switches)
        static const int nf3 = (getenv("PG_LEAF3_NF") ? std::max(1, std::min(4, atoi(getenv("PG_LEAF3_NF")))) : 1) |
                               ((getenv("PG_LEAF3_ALONE") && !atoi(getenv("PG_LEAF3_ALONE"))) ? 0 : 16) |
                               ((getenv("PG_LEAF3_DIAG") && atoi(getenv("PG_LEAF3_DIAG"))) ? 64 : 0);
        if (pg_leaf3_blk() && sizeof(T) == 8) hipLaunchKernelGGL((pg_leaf3_kernel<T, true>), dim3(nexp), dim3(NTH), lds, st, A, lda, inv, ldi, info, col0, eA, eInv, nf3);
        else hipLaunchKernelGGL(pg_leaf3_kernel<T>, dim3(nexp), dim3(NTH), lds, st, A, lda, inv, ldi, info, col0, eA, eInv, nf3);
        PG_CHECK(hipGetLastError());
        return 0;
    }
    if (form == 1) hipLaunchKernelGGL(pg_leaf_kernel<T>, dim3(nexp), dim3(NTH), lds, st, A, lda, inv, ldi, info, col0, ablate, eA, eInv);
    else hipLaunchKernelGGL(pg_leaf2_kernel<T>, dim3(nexp), dim3(NTH), lds, st, A, lda, inv, ldi, info, col0, ablate, eA, eInv);
    PG_CHECK(hipGetLastError());
    return 0;
}
template int pg_leaf<double>(hipStream_t, double*, long, double*, long, int*, int, int, int, long, long);
template int pg_leaf<float>(hipStream_t, float*, long, float*, long, int*, int, int, int, long, long);
