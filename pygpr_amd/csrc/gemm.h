// MFMA GEMM core of libpygpr_hip: one tiled kernel family that carries every O(n^3) step
// (SYRK / GEMM updates of the Cholesky, TRSM-by-inverse, triangular inverse, L^-T L^-1,
// predictive-variance product).  fp64 uses v_mfma_f64_16x16x4_f64, fp32 v_mfma_f32_16x16x4_f32.
#pragma once
#include "common.h"

// C[M x N] = beta * C + alpha * opA(A) * opB(B)
//   TA == false: A holds opA as M x K row-major (K contiguous);  TA == true: A holds K x M (M contiguous)
//   TB == false: B holds opB as K x N row-major (N contiguous);  TB == true: B holds N x K (K contiguous)
template <typename T> struct GemmP {
    const T* A;
    const T* B;
    T* C;
    long lda, ldb, ldc;
    int M, N, K;      // multiples of the block tile / 16
    T alpha, beta;
    int tri;          // 1: only tiles on or below the diagonal (BM == BN; N < M: the leading N x N triangle + the full rows below it).
                      // What lies strictly ABOVE the diagonal inside the diagonal tiles is unspecified afterwards: whole 128 x 128 tiles
                      // update it, 64 x 64 tiles and the quarter tiles of a mixed launch do not.  Consumers (the leaf, symmetrize, the
                      // triangular products) read the lower triangle only: test_factor_ignores_the_upper_triangle.
    int klo, khi;     // K-range from a triangular operand: 0 none, 1 follows the tile row, 2 the tile column
    int krev;         // klo launches: walk each tile's K range from its end downwards (all tiles start at the same k)
    long sA, sB, sC;  // batch strides (elements), grid.y = batch * nexp
    int batch;
    int nexp;         // independent problems of the same shape in one launch (batched experts): z = blockIdx.y -> (z % batch, z / batch)
    long eA, eB, eC;  // their strides (elements)
    int einfo;        // stride of `info` between them (0: one shared flag)
    T* part;          // EPI == 1: partial column sums of squares, [M/64][ldp]
    long ldp;
    const int* info;  // device flag: kernels exit at once when *info != 0 (failed factorisation)
    int noxcd;        // 1: keep launch order instead of the XCD-chunked tile order (experiments)
    int atomic_c;     // set by pg_gemm (fp64, beta == 1): the epilogue adds into C with no-return atomics instead of load + store
};

enum GemmVariant {
    GEMM_NT_128 = 0,   // C = a A B^T + b C           (SYRK / panel updates)
    GEMM_NT_RP = 1,    // 64 x 256 row-panel tile: in-place  B <- B inv(L)^T
    GEMM_NN_128 = 2,   // triangular inverse steps
    GEMM_TN_128 = 3,   // L^-T L^-1
    GEMM_NN_128_SS = 4, // C not stored: column sums of squares of the product (predictive variance)
    GEMM_TT_128 = 5,    // C = a A^T B^T (parks (L21 X11)^T in the mirrored block of the triangular inverse)
    GEMM_NT_64 = 6,     // NT with 64 x 64 block tiles (skinny outputs: 4x the workgroups of the 128 tile)
    GEMM_NT_64x128 = 7, // NT with 64 x 128 block tiles: in-place panel solve against a 128 x 128 inverse
    GEMM_NT_32x64 = 8,  // the two skinny products of the Cholesky chain when they have fewer than two 64-row workgroups per
    GEMM_NT_32x128 = 9, // CU: half the row tile (and a 32-deep K tile) puts two waves on every SIMD (45 % -> MFMA use)
    GEMM_TT_64 = 10,    // TT with 64 x 64 tiles: the small levels of the triangular inverse (few, long 128 x 128 tiles otherwise)
    GEMM_NT_128_SS = 12, // the same epilogue on C = A B^T: the cross-covariance stored test-point-major (K-contiguous B operand)
    GEMM_TN_64 = 13,    // TN with 64 x 64 tiles: K** - V^T V for a few thousand test points (136 lower 128 x 128 tiles leave most CUs idle)
    GEMM_NT_32x32 = 11  // NT with 32 x 32 tiles and a 64-deep K tile: the chain's U product in the chain-bound tail of the Cholesky
                        // (latency per launch, not throughput, is what counts there: 4x the workgroups, half the K iterations)
};

template <typename T> int pg_gemm(pg_ctx* ctx, hipStream_t st, int variant, const GemmP<T>& p);
double pg_gemm_flops(int variant, int M, int N, int K, int tri, int klo, int khi, int batch);
