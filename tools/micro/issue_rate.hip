// Does a lone wave's instruction issue rate depend on which other waves are RESIDENT on its SIMD, even when they are idle?
// The leaf's sixteen-column pivot loop (tall_step<5> of tallstep.hip: ~1070 straight-line instructions, no memory access) is timed on
// wave 0 of a workgroup of NW waves whose other waves (mode 0) end at once, (mode 1) wait at a barrier until wave 0 is done,
// (mode 2) sleep-poll an LDS flag.  The same compiled kernel for every NW and mode.  Prints each wave's SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue_rate tools/micro/issue_rate.hip && tools/micro/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ double bcast_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double inv_sqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
__device__ __forceinline__ double recip(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}
__device__ __forceinline__ long long rdclk() {   // volatile: stays between the volatile register pins around the step
    long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
__device__ __forceinline__ void tall_step(double (&row)[16]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double piv = bcast_lane(row[c], c);
        const double y = recip(piv);
        const double w = row[c] * y;
        row[c] *= inv_sqrt(piv);
#pragma unroll
        for (int k = c + 1; k < 16; ++k) row[k] -= row[c] * bcast_lane(w, k);   // (row[c] already scaled: timing only)
    }
}

template <int LB>
__global__ __launch_bounds__(LB) void occ(const double* __restrict__ in, double* __restrict__ out, long long* t, int* simd, int reps, int mode,
                                          int nactive) {
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (lane == 0) simd[wave] = (__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) >> 4) & 3;
    if (tid == 0) flag = 0;
    __syncthreads();
    if (wave >= nactive) {
        if (mode == 0) return;
        if (mode == 1) { __syncthreads(); return; }
        while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(8);
        return;
    }
    double base[16], row[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) base[c] = in[(lane & 15) * 16 + c] + (lane >= 16 ? 0.25 * lane : 0.0);
    long long acc = 0, first = 0;
    for (int rep = 0; rep < reps; ++rep) {
        const long long c0 = rdclk();
#pragma unroll
        for (int c = 0; c < 16; ++c) { row[c] = base[c]; asm volatile("" : "+v"(row[c])); }
        tall_step(row);
#pragma unroll
        for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(row[c]));
        const long long dt = rdclk() - c0;
        acc += dt;
        if (rep == 0) first = dt;
    }
    if (wave == 0) {
        if (lane == 0) { t[0] = acc / reps; t[1] = first; }
        out[lane] = row[lane & 15];
        if (mode == 2 && lane == 0) __hip_atomic_store(&flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (mode == 1) __syncthreads();
}

int main() {
    std::vector<double> h(256);
    srand(1);
    double m[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) m[i][j] = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 1.0 : 0.0; for (int k = 0; k < 16; ++k) s += m[i][k] * m[j][k] / 16; h[i * 16 + j] = s; }
    double *din, *dout; long long* dt; int* ds;
    hipMalloc(&din, 256 * 8); hipMalloc(&dout, 64 * 8); hipMalloc(&dt, 16); hipMalloc(&ds, 64);
    hipMemcpy(din, h.data(), 256 * 8, hipMemcpyHostToDevice);
    long long ht[2]; int hs[16];
    const int nws[] = {1, 2, 4, 5, 8, 9, 12, 16};
    for (int lb = 0; lb < 2; ++lb)
        for (int mode = 0; mode < 3; ++mode)
            for (int nw : nws) {
                if (lb == 0 && nw > 4) continue;
                for (int nactive = 1; nactive <= (nw >= 4 ? 4 : 1); nactive += 3) {
                    if (lb == 0) occ<256><<<1, 64 * nw>>>(din, dout, dt, ds, 400, mode, nactive);
                    else occ<1024><<<1, 64 * nw>>>(din, dout, dt, ds, 400, mode, nactive);
                    hipDeviceSynchronize();
                    hipMemcpy(ht, dt, 16, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, 64, hipMemcpyDeviceToHost);
                    printf("launch_bounds %4d mode %d (%s) waves %2d active %d: %5lld clocks per step (first %5lld); SIMD of waves:", lb ? 1024 : 256, mode,
                           mode == 0 ? "others end      " : (mode == 1 ? "others at barrier" : "others sleep-poll"), nw, nactive, ht[0], ht[1]);
                    for (int w = 0; w < nw; ++w) printf(" %d", hs[w]);
                    printf("\n");
                }
            }
    return 0;
}
