// Bare v_mfma_f64_16x16x4_f64 issue rate: NACC independent accumulators, no memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void k(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
#pragma unroll 4
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu, int threads) {
    int ncu = 256, iters = 4000;
    double* out; hipMalloc(&out, sizeof(double) * ncu * blocks_per_cu * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) k<NACC><<<ncu * blocks_per_cu, threads>>>(out, iters, 1.0, 2.0);   // clock ramp
    hipDeviceSynchronize();
    float ms = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k<NACC><<<ncu * blocks_per_cu, threads>>>(out, iters, 1.0, 2.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (t < ms) ms = t;
    }
    double waves = (double)ncu * blocks_per_cu * threads / 64;
    double flops = waves * iters * NACC * 2048.0;
    double waves_per_simd = waves / (ncu * 4);
    printf("NACC=%2d blocks/CU=%d threads=%d waves/SIMD=%.1f: %.3f ms  %.1f TFLOP/s  (cycles per MFMA per SIMD at 2.4 GHz: %.1f)\n", NACC,
           blocks_per_cu, threads, waves_per_simd, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * NACC * waves_per_simd));
    hipFree(out);
}
int main() {
    run<16>(1, 256); run<16>(2, 256); run<4>(1, 256); run<4>(2, 256); run<1>(1, 256); run<16>(4, 256);
    return 0;
}
