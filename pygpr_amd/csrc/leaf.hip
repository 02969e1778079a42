// Cholesky leaf: factor one 128x128 diagonal block entirely in LDS (one workgroup of 16 waves) and
// produce its inverse, which turns every triangular solve above it into an MFMA GEMM.
//   1. load the lower triangle into LDS S[128][130]
//   2. unscaled right-looking elimination, one barrier per column:
//        S[i][k] -= S[i][j] * S[k][j] / S[j][j]     (j < k <= i)
//      column j then holds L_ij * L_jj; a final pass divides by sqrt(S_jj).  A non-positive (or NaN)
//      pivot sets *info = global column + 1 (LAPACK convention, first failure wins) and stops.
//   3. write L back (upper part of the block zeroed)
//   4. inverse by recursive doubling: 16x16 diagonal blocks by forward substitution in registers (one
//      lane per column), then X21 = -X22 (L21 X11) level by level with 16x16x4 MFMAs on LDS operands;
//      the mirrored (upper) block is the scratch for L21 X11.
#include "leaf.h"

#define NB 128
#define LD 130

template <typename T>
__device__ __forceinline__ void lds_tile_mm(T* C, const T* A, const T* B, int K, T alpha, int lane) {
    // C(16x16) = alpha * A(16xK) * B(Kx16); all three row-major in LDS with leading dimension LD
    typename Mfma<T>::acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (T)0;
    const int fr = lane & 15, fk = lane >> 4;
    for (int k = 0; k < K; k += 4) {
        const T a = A[fr * LD + k + fk];
        const T b = B[(k + fk) * LD + fr];
        acc = Mfma<T>::run(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) C[Mfma<T>::row(lane, r) * LD + fr] = alpha * acc[r];
}

template <typename T>
__global__ __launch_bounds__(1024) void pg_leaf_kernel(T* __restrict__ A, long lda, T* __restrict__ inv,
                                                       long ldi, int* __restrict__ info, int col0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* S = reinterpret_cast<T*>(smem_raw);
    // all LDS lives in the dynamic region (keeps its base 16-byte aligned): the flag sits behind S
    int& fail = *reinterpret_cast<int*>(smem_raw + (size_t)NB * LD * sizeof(T));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (*info != 0) return;
    if (tid == 0) fail = 0;
    for (int idx = tid; idx < NB * NB; idx += 1024) {
        const int i = idx >> 7, k = idx & 127;
        S[i * LD + k] = (k <= i) ? A[(long)i * lda + k] : (T)0;
    }

    const int ti = tid >> 5, tk = tid & 31;
    for (int j = 0; j < NB; ++j) {
        __syncthreads();
        const T piv = S[j * LD + j];
        if (!(piv > (T)0)) {   // uniform: every thread reads the same pivot
            if (tid == 0) { fail = 1; atomicCAS(info, 0, col0 + j + 1); }
            break;
        }
        const T rp = (T)1 / piv;
        for (int i = j + 1 + ti; i < NB; i += 32) {
            const T lij = S[i * LD + j] * rp;
            for (int k = j + 1 + tk; k <= i; k += 32) S[i * LD + k] -= lij * S[k * LD + j];
        }
    }
    __syncthreads();
    if (fail) return;

    // scale columns: off-diagonals first (they read the unscaled diagonal), then the diagonal
    for (int idx = tid; idx < NB * NB; idx += 1024) {
        const int i = idx >> 7, k = idx & 127;
        if (k < i) S[i * LD + k] = S[i * LD + k] / sqrt(S[k * LD + k]);
    }
    __syncthreads();
    if (tid < NB) S[tid * LD + tid] = sqrt(S[tid * LD + tid]);
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += 1024) {
        const int i = idx >> 7, k = idx & 127;
        A[(long)i * lda + k] = S[i * LD + k];   // upper part of S is zero
    }
    if (inv == nullptr) return;
    __syncthreads();

    // ---- inverse, level 0: the eight 16x16 diagonal blocks (waves 0..7, lanes 0..15 = columns)
    if (wave < 8 && lane < 16) {
        const T* D = S + (wave * 16) * LD + wave * 16;
        T x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            T s = (i == lane) ? (T)1 : (T)0;
#pragma unroll
            for (int k = 0; k < i; ++k) s -= D[i * LD + k] * x[k];
            x[i] = s / D[i * LD + i];
        }
        T* Dw = S + (wave * 16) * LD + wave * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) Dw[i * LD + lane] = x[i];
    }
    __syncthreads();

    // ---- levels s = 16, 32, 64
    for (int s = 16; s < NB; s <<= 1) {
        const int tps = s / 16;                    // 16-tiles per block side
        const int tiles = (NB / (2 * s)) * tps * tps;
        // T = L21 * X11 -> mirrored block (r0, r0+s)
        for (int t = wave; t < tiles; t += 16) {
            const int pr = t / (tps * tps), tt = t % (tps * tps), bi = tt / tps, bj = tt % tps;
            const int r0 = pr * 2 * s;
            lds_tile_mm<T>(S + (r0 + bi * 16) * LD + r0 + s + bj * 16,      // T tile
                           S + (r0 + s + bi * 16) * LD + r0,                // L21 rows
                           S + r0 * LD + r0 + bj * 16,                      // X11 cols
                           s, (T)1, lane);
        }
        __syncthreads();
        // X21 = -X22 * T
        for (int t = wave; t < tiles; t += 16) {
            const int pr = t / (tps * tps), tt = t % (tps * tps), bi = tt / tps, bj = tt % tps;
            const int r0 = pr * 2 * s;
            lds_tile_mm<T>(S + (r0 + s + bi * 16) * LD + r0 + bj * 16,      // X21 tile
                           S + (r0 + s + bi * 16) * LD + r0 + s,            // X22 rows
                           S + r0 * LD + r0 + s + bj * 16,                  // T cols
                           s, (T)-1, lane);
        }
        __syncthreads();
        // clear the scratch blocks again (they sit in the upper triangle)
        for (int idx = tid; idx < (NB / (2 * s)) * s * s; idx += 1024) {
            const int pr = idx / (s * s), e = idx % (s * s), i = e / s, k = e % s;
            const int r0 = pr * 2 * s;
            S[(r0 + i) * LD + r0 + s + k] = (T)0;
        }
        __syncthreads();
    }
    for (int idx = tid; idx < NB * NB; idx += 1024) {
        const int i = idx >> 7, k = idx & 127;
        inv[(long)i * ldi + k] = S[i * LD + k];
    }
}

template <typename T> int pg_leaf(hipStream_t st, T* A, long lda, T* inv, long ldi, int* info, int col0) {
    const size_t lds = (size_t)NB * LD * sizeof(T) + 16;
    static bool attr_done = false;
    auto kern = pg_leaf_kernel<T>;
    if (!attr_done) {
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, A, lda, inv, ldi, info, col0);
    PG_CHECK(hipGetLastError());
    return 0;
}
template int pg_leaf<double>(hipStream_t, double*, long, double*, long, int*, int);
template int pg_leaf<float>(hipStream_t, float*, long, float*, long, int*, int);
