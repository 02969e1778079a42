// Cholesky leaf: factor one 128x128 diagonal block entirely in LDS (one 256-thread workgroup) and produce
// its inverse, which turns every triangular solve above it into an MFMA GEMM.
//
// Blocked right-looking with 16-wide micro-panels; per micro-panel jb:
//   A. wave 0 factors the 16x16 diagonal block in REGISTERS (lane r holds row r; pivots and multipliers
//      are broadcast with v_readlane, 1/sqrt by v_rsq + 2 Newton steps) and inverts it the same way
//      (lane c solves column c).  A non-positive / NaN pivot sets *info = global column + 1 (LAPACK
//      convention, first failure wins) and stops.
//   B. panel solve  P <- P inv(D)^T  for the rows below, 16x16x4 MFMAs on LDS operands
//   C. trailing update  T <- T - P P^T  (lower 16x16 tiles), MFMA with the tile as the accumulator
// The inverse of the whole block then follows by recursive doubling from the eight 16x16 inverses:
// X21 = -X22 (L21 X11), level by level (16, 32, 64), again MFMA on LDS; the mirrored (upper) block is
// the scratch for L21 X11.  LDS: S[128][130] + 8 x [16][18] inverses, all in the dynamic region.
#include "leaf.h"

#define NB 128
#define LD 130
#define DLD 18

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
__device__ __forceinline__ float inv_sqrt(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    r = r * (1.5f - 0.5f * x * r * r);
    return r;
}

// C(16x16) = beta * C + alpha * A(16xK) * op(B); A row-major [16][K] at lda; B either [K][16] (TB = false)
// or [16][K] (TB = true) at ldb; all in LDS.  One wave.
template <typename T, bool TB>
__device__ __forceinline__ void lds_tile_mm(T* C, int ldc, const T* A, int lda, const T* B, int ldb, int K, T alpha,
                                            T beta, int lane) {
    typename Mfma<T>::acc_t acc;
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (T)0;
    for (int k = 0; k < K; k += 4) {
        const T a = A[fr * lda + k + fk];
        const T b = TB ? B[fr * ldb + k + fk] : B[(k + fk) * ldb + fr];
        acc = Mfma<T>::run(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        T* c = C + Mfma<T>::row(lane, r) * ldc + fr;
        T v = alpha * acc[r];
        if (beta != (T)0) v += beta * *c;
        *c = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pg_leaf_kernel(T* __restrict__ A, long lda, T* __restrict__ inv, long ldi,
                                                      int* __restrict__ info, int col0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* S = reinterpret_cast<T*>(smem_raw);
    T* Dinv = S + NB * LD;                                  // [8][16][DLD]
    int& fail = *reinterpret_cast<int*>(Dinv + 8 * 16 * DLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (*info != 0) return;
    if (tid == 0) fail = 0;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx >> 7, k = idx & 127;
        S[i * LD + k] = (k <= i) ? A[(long)i * lda + k] : (T)0;
    }
    __syncthreads();

    for (int jb = 0; jb < NB / 16; ++jb) {
        const int c0 = jb * 16, r0 = c0 + 16;
        // ---- A: diagonal 16x16 block, registers of wave 0
        if (wave == 0) {
            const int r = lane & 15;
            T* D = S + c0 * LD + c0;
            T row[16], rdiag[16];
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
                rdiag[c] = rs;
                const T lrc = row[c] * rs;
                row[c] = lrc;
#pragma unroll
                for (int k = c + 1; k < 16; ++k) row[k] -= lrc * bcast_lane(lrc, k);
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) D[r * LD + c] = (c <= r) ? row[c] : (T)0;
            }
            // inverse: lane c (0..15) solves L x = e_c; L[i][k] lives in lane i's row[k]
            T x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                T s = (i == r) ? (T)1 : (T)0;
#pragma unroll
                for (int k = 0; k < i; ++k) s -= bcast_lane(row[k], i) * x[k];
                x[i] = s * rdiag[i];
            }
            if (lane < 16) {
                T* Dv = Dinv + jb * 16 * DLD;
#pragma unroll
                for (int i = 0; i < 16; ++i) Dv[i * DLD + r] = x[i];
            }
        }
        __syncthreads();
        if (fail) return;
        if (r0 >= NB) break;
        const int nt = (NB - r0) / 16;   // 16-row tiles below the diagonal block
        // ---- B: P <- P inv(D)^T
        for (int t = wave; t < nt; t += 4)
            lds_tile_mm<T, true>(S + (r0 + t * 16) * LD + c0, LD, S + (r0 + t * 16) * LD + c0, LD, Dinv + jb * 16 * DLD,
                                 DLD, 16, (T)1, (T)0, lane);
        __syncthreads();
        // ---- C: T <- T - P P^T on the lower tiles
        const int ntri = nt * (nt + 1) / 2;
        for (int t = wave; t < ntri; t += 4) {
            int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (ti * (ti + 1) / 2 > t) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            const int tj = t - ti * (ti + 1) / 2;
            lds_tile_mm<T, true>(S + (r0 + ti * 16) * LD + r0 + tj * 16, LD, S + (r0 + ti * 16) * LD + c0, LD,
                                 S + (r0 + tj * 16) * LD + c0, LD, 16, (T)-1, (T)1, lane);
        }
        __syncthreads();
    }

    // L back to global (diagonal tiles of C above wrote the strictly upper 16x16 corners: mask them)
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx >> 7, k = idx & 127;
        A[(long)i * lda + k] = (k <= i) ? S[i * LD + k] : (T)0;
    }
    if (inv == nullptr) return;
    __syncthreads();

    // ---- inverse, level 0: drop the 16x16 inverses on the diagonal, clear everything above it
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx >> 7, k = idx & 127;
        if (k > i) S[i * LD + k] = (T)0;
        else if ((i >> 4) == (k >> 4)) S[i * LD + k] = Dinv[(i >> 4) * 16 * DLD + (i & 15) * DLD + (k & 15)];
    }
    __syncthreads();

    // ---- levels s = 16, 32, 64: X21 = -X22 (L21 X11)
    for (int s = 16; s < NB; s <<= 1) {
        const int tps = s / 16;
        const int tiles = (NB / (2 * s)) * tps * tps;
        for (int t = wave; t < tiles; t += 4) {
            const int pr = t / (tps * tps), tt = t % (tps * tps), bi = tt / tps, bj = tt % tps;
            const int q0 = pr * 2 * s;
            lds_tile_mm<T, false>(S + (q0 + bi * 16) * LD + q0 + s + bj * 16, LD,      // scratch tile (mirror)
                                  S + (q0 + s + bi * 16) * LD + q0, LD,                // L21 rows
                                  S + q0 * LD + q0 + bj * 16, LD, s, (T)1, (T)0, lane);   // X11 cols
        }
        __syncthreads();
        for (int t = wave; t < tiles; t += 4) {
            const int pr = t / (tps * tps), tt = t % (tps * tps), bi = tt / tps, bj = tt % tps;
            const int q0 = pr * 2 * s;
            lds_tile_mm<T, false>(S + (q0 + s + bi * 16) * LD + q0 + bj * 16, LD,      // X21 tile
                                  S + (q0 + s + bi * 16) * LD + q0 + s, LD,            // X22 rows
                                  S + q0 * LD + q0 + s + bj * 16, LD, s, (T)-1, (T)0, lane);   // scratch cols
        }
        __syncthreads();
        for (int idx = tid; idx < (NB / (2 * s)) * s * s; idx += 256) {
            const int pr = idx / (s * s), e = idx % (s * s), i = e / s, k = e % s;
            S[(pr * 2 * s + i) * LD + pr * 2 * s + s + k] = (T)0;
        }
        __syncthreads();
    }
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx >> 7, k = idx & 127;
        inv[(long)i * ldi + k] = S[i * LD + k];
    }
}

template <typename T> int pg_leaf(hipStream_t st, T* A, long lda, T* inv, long ldi, int* info, int col0) {
    const size_t lds = (size_t)(NB * LD + 8 * 16 * DLD) * sizeof(T) + 16;
    static bool attr_done = false;
    auto kern = pg_leaf_kernel<T>;
    if (!attr_done) {
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, st, A, lda, inv, ldi, info, col0);
    PG_CHECK(hipGetLastError());
    return 0;
}
template int pg_leaf<double>(hipStream_t, double*, long, double*, long, int*, int);
template int pg_leaf<float>(hipStream_t, float*, long, float*, long, int*, int);
