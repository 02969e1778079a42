// Micro-benchmark of the leaf's serial core: the register-resident 16-column tall-panel step (leaf.hip), one wave per SIMD.
// Variants of the pivot chain are timed (shader clocks per 16-column step) and checked against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/tallstep tools/micro/tallstep.hip && tools/micro/tallstep
#include <hip/hip_runtime.h>
#include <cmath>
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

template <int V> __device__ __forceinline__ void tall_step(double (&row)[16]);

// V0: the shipped form -- right-looking, 1/sqrt(pivot) on the chain
template <> __device__ __forceinline__ void tall_step<0>(double (&row)[16]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double piv = bcast_lane(row[c], c);
        const double rs = inv_sqrt(piv);
        const double lrc = row[c] * rs;
        row[c] = lrc;
#pragma unroll
        for (int k = c + 1; k < 16; ++k) row[k] -= lrc * bcast_lane(lrc, k);
    }
}
// V1: square-root-free chain (L D L^T inside the micro-panel): 1/pivot on the chain, the square roots afterwards, off it
template <> __device__ __forceinline__ void tall_step<1>(double (&row)[16]) {
    double piv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        piv[c] = bcast_lane(row[c], c);
        const double y = recip(piv[c]);
        const double w = row[c] * y;
#pragma unroll
        for (int k = c + 1; k < 16; ++k) row[k] -= row[c] * bcast_lane(w, k);
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] *= inv_sqrt(piv[c]);
}
// V5: V1 with each pivot's square root taken inside its own iteration (off the chain: nothing waits for it), so that the sixteen
// pivots do not stay in scalar registers until the end (32 SGPRs live across the loop left the broadcasts two pairs to go through)
template <> __device__ __forceinline__ void tall_step<5>(double (&row)[16]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double piv = bcast_lane(row[c], c);
        const double w = row[c] * recip(piv);
        const double rs = inv_sqrt(piv);
#pragma unroll
        for (int k = c + 1; k < 16; ++k) row[k] -= row[c] * bcast_lane(w, k);
        row[c] *= rs;
    }
}
// V2: two columns per pivot step (2 x 2 diagonal blocks, one reciprocal of the determinant per pair)
template <> __device__ __forceinline__ void tall_step<2>(double (&row)[16]) {
    double pa[8], pb[8], pdet[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int c = 2 * p;
        const double a = bcast_lane(row[c], c), b = bcast_lane(row[c], c + 1), d = bcast_lane(row[c + 1], c + 1);
        const double det = __builtin_fma(a, d, -(b * b));
        const double r = recip(det);
        const double i11 = d * r, i12 = -b * r, i22 = a * r;
        const double w1 = __builtin_fma(row[c + 1], i12, row[c] * i11);
        const double w2 = __builtin_fma(row[c + 1], i22, row[c] * i12);
#pragma unroll
        for (int k = c + 2; k < 16; ++k) {
            row[k] -= row[c] * bcast_lane(w1, k);
            row[k] -= row[c + 1] * bcast_lane(w2, k);
        }
        pa[p] = a; pb[p] = b; pdet[p] = det;
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int c = 2 * p;
        const double ra = recip(pa[p]);
        const double rs_a = inv_sqrt(pa[p]);
        const double t = __builtin_fma(-row[c], pb[p] * ra, row[c + 1]);    // u_{r,c+1} - u_{r,c} b / a
        row[c + 1] = t * inv_sqrt(pdet[p] * ra);
        row[c] *= rs_a;
    }
}
// V3: V0 with the coupled (Goldschmidt) refinement of 1/sqrt: two dependent operations per iteration instead of three
template <> __device__ __forceinline__ void tall_step<3>(double (&row)[16]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double piv = bcast_lane(row[c], c);
        const double y0 = __builtin_amdgcn_rsq(piv);
        double g = piv * y0, h = 0.5 * y0;
        double r = __builtin_fma(-g, h, 0.5);
        g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
        r = __builtin_fma(-g, h, 0.5);
        h = __builtin_fma(h, r, h);
        const double lrc = (row[c] + row[c]) * h;
        row[c] = lrc;
#pragma unroll
        for (int k = c + 1; k < 16; ++k) row[k] -= lrc * bcast_lane(lrc, k);
    }
}

// V4: no scalar registers at all -- broadcasts by DPP row_newbcast inside each row of sixteen lanes.  Every 16-lane row of the wave
// carries its own copy of the diagonal block's sixteen rows (register set d[]) AND sixteen more rows (register set x[]: identity or
// panel rows); lane k of a row broadcasts to its row only, which is all a row needs.  Two updates per broadcast instead of one, but
// no v_readlane -> SGPR -> VALU hop (and no dependence on how many scalar registers the surrounding kernel has left).
template <int K> __device__ __forceinline__ double row_bcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int C> struct ColStep {
    static __device__ __forceinline__ void run(double (&d)[16], double (&x)[16], double (&piv)[16]) {
        piv[C] = row_bcast<C>(d[C]);
        const double y = recip(piv[C]);
        const double w = d[C] * y;
        upd<C + 1>(d, x, w);
        ColStep<C + 1>::run(d, x, piv);
    }
    template <int Kk> static __device__ __forceinline__ void upd(double (&d)[16], double (&x)[16], double w) {
        if constexpr (Kk < 16) {
            const double wk = row_bcast<Kk>(w);
            d[Kk] -= d[C] * wk;
            x[Kk] -= x[C] * wk;
            upd<Kk + 1>(d, x, w);
        }
    }
};
template <> struct ColStep<16> { static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double (&)[16]) {} };
template <> __device__ __forceinline__ void tall_step<4>(double (&row)[16]) {
    // lanes of every 16-lane row: d = the diagonal block's row (lane & 15), x = this lane's own extra row (here: the input row)
    double d[16], piv[16];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = __shfl(row[c], lane & 15, 64);      // set-up only (the leaf would load both sets from LDS)
    ColStep<0>::run(d, row, piv);
#pragma unroll
    for (int c = 0; c < 16; ++c) { const double rs = inv_sqrt(piv[c]); row[c] *= rs; d[c] *= rs; }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) row[c] = d[c];                           // report the diagonal rows from lanes 0-15 like the other variants
    }
}

template <int V> __global__ __launch_bounds__(256) void bench(const double* __restrict__ in, double* __restrict__ out, long long* t, int reps) {
    const int lane = threadIdx.x & 63;
    double orig[16], row[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) orig[c] = in[lane * 16 + c];
    double sink = 0.0;
    const long long w0 = wall_clock64(), c0 = clock64();
    long long cfirst = 0;
#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { row[c] = orig[c] + sink; asm volatile("" : "+v"(row[c])); }   // sink: every step waits for the one before
        tall_step<V>(row);
        sink = row[15] * 0.0;          // (finite inputs: exactly 0, but a true dependence on the step's last value)
        if (rep == 0) { asm volatile("" : "+v"(sink)); cfirst = clock64() - c0; }
    }
    const long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { t[0] = (c1 - c0) / reps; t[1] = (w1 - w0) * 10 / reps; t[2] = cfirst; }      // shader clocks, ns, clocks of the FIRST (cold-code) step
    if (threadIdx.x < 64 && blockIdx.x == 0)
        for (int c = 0; c < 16; ++c) out[lane * 16 + c] = row[c];
}

// The step as the leaf runs it: rows come from LDS (S[128][130], one row per lane), go back to LDS, barrier.  768 threads, waves 4-11
// only take part in the barriers.  W selects how the rows are written back: 0 = as shipped (one masked block per lane class),
// 1 = every lane writes its 16 values unconditionally (no select, no divergence), 2 = no write at all (floor)
#define LD 130
template <int V, int W> __global__ __launch_bounds__(768) void phase_bench(const double* __restrict__ in, double* __restrict__ out, long long* t, int reps) {
    __shared__ double S[128 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 128; i += 768) S[(i >> 7) * LD + (i & 127)] = (i >> 7) == (i & 127) ? 4.0 : 0.01 * ((i * 7) % 13);
    __syncthreads();
    long long acc[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < reps; ++rep) {
        const int c0 = 16 * (rep & 3), r0 = c0 + 16;
        if (wave < 4) {
            const int pl = lane - 16, prow = r0 + 48 * wave + pl;
            const bool is_diag = lane < 16, is_panel = !is_diag && wave < 3 && prow < 128;
            const double* src = S + (is_diag ? c0 + lane : (is_panel ? prow : c0)) * LD + c0;
            double row[16];
            const long long t0 = wall_clock64();
#pragma unroll
            for (int c = 0; c < 16; ++c) { const double v = src[c]; row[c] = (is_diag || is_panel) ? v : (c == pl ? 1.0 : 0.0); }
            // keep the matrix the same every rep: use the pristine diagonal block from `in` for the diagonal lanes
#pragma unroll
            for (int c = 0; c < 16; ++c) { if (is_diag) row[c] = in[lane * 16 + c]; asm volatile("" : "+v"(row[c])); }
            const long long t1 = wall_clock64();
            tall_step<V>(row);
#pragma unroll
            for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(row[c]));
            const long long t2 = wall_clock64();
            if (W == 0) {
                if (is_diag) { if (wave == 0) { double* D = S + (c0 + lane) * LD + c0;
#pragma unroll
                        for (int c = 0; c < 16; ++c) D[c] = (c <= lane) ? row[c] : 0.0; } }
                else if (is_panel) { double* P = S + prow * LD + c0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) P[c] = row[c]; }
            } else if (W == 1) {
                double* P = S + (is_diag ? (wave == 0 ? c0 + lane : 127) : (is_panel ? prow : 127)) * LD + (is_diag && wave ? 112 : c0);
#pragma unroll
                for (int c = 0; c < 16; ++c) P[c] = row[c];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
            const long long t3 = wall_clock64();
            acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2;
        }
        const long long t4 = wall_clock64();
        __syncthreads();
        acc[3] += wall_clock64() - t4;
    }
    if (tid == 0) for (int i = 0; i < 4; ++i) t[i] = acc[i] * 10 / reps;
    if (tid < 64) out[tid] = S[tid * LD + tid];
}

// The third leaf's structure in isolation: 768 threads; per repetition wave 0 runs the square-root-free step on LDS-resident rows
// (loads, loop, stores) while the other eleven waves either wait at the barrier (BUSY = 0) or run LDS-fed fp64 MFMA tiles (BUSY = 1),
// or every wave runs the step on its own copy (BUSY = 2: does company on other SIMDs, fetching the same code, change wave 0's time?).
typedef double d4 __attribute__((ext_vector_type(4)));
template <int BUSY> __global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void struct_bench(const double* __restrict__ in, double* __restrict__ out, long long* t, int reps) {
    __shared__ double S[128 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: a SCALAR branch below
    for (int i = tid; i < 128 * LD; i += 768) S[i] = 0.01 * ((i * 7) % 13);
    __syncthreads();
    for (int i = tid; i < 256; i += 768) S[(i >> 4) * LD + (i & 15)] = in[i];      // the SPD diagonal block (rows 0-15)
    __syncthreads();
    long long acc = 0, first = 0;
    for (int rep = 0; rep < reps; ++rep) {
        if (wave == 0 || BUSY == 2) {
            double row[16];
            const long long c0 = clock64();
#pragma unroll
            for (int c = 0; c < 16; ++c) row[c] = S[(lane & 15) * LD + c];
#pragma unroll
            for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(row[c]));
            tall_step<5>(row);
#pragma unroll
            for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(row[c]));
            if (wave == 0 && lane >= 32 && lane < 48) {                               // results go to a scratch block (the input stays)
#pragma unroll
                for (int c = 0; c < 16; ++c) S[(32 + (lane & 15)) * LD + 16 + c] = row[c];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const long long dt = clock64() - c0;
            if (wave == 0) { acc += dt; if (rep == 0) first = dt; }
        } else if (BUSY == 1) {
            // LDS-fed MFMA tiles like the leaf's deferred update: 6 tiles of K = 16 per wave and repetition
            const int fr = lane & 15, fk = lane >> 4;
            for (int tile = 0; tile < 6; ++tile) {
                d4 a4 = {0, 0, 0, 0};
                const double* Ap = S + (48 + 16 * (tile % 5) + fr) * LD + 32;
                const double* Bp = S + (64 + fr) * LD + 48;
#pragma unroll
                for (int q = 0; q < 4; ++q) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[4 * q + fk], Bp[4 * q + fk], a4, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(96 + fk + 4 * r) * LD + 64 + 16 * (wave % 4) + fr] = a4[r];
            }
        }
        __syncthreads();
    }
    if (tid == 0) { t[0] = acc / reps; t[1] = first; }
    if (tid < 64) out[tid] = S[(32 + (tid & 15)) * LD + 16 + (tid & 15)];
}

// dependent-chain probes: cycles per dependent op
__global__ void chain_probe(double* out, long long* t) {
    double x = 1.0 + 1e-9 * threadIdx.x, y = 0.999999;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fma(x, y, 1e-12);
    }
    long long t1 = clock64();
    double z = 1.5 + 1e-9 * threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) z = __builtin_amdgcn_rsq(z) + 1.0;
    }
    long long t2 = clock64();
    double w = 1.25 + 1e-9 * threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) w = bcast_lane(w, j) * 1.0000001 + (double)(threadIdx.x & 1) * 1e-30;
    }
    long long t3 = clock64();
    out[threadIdx.x] = x + z + w;
    if (threadIdx.x == 0) { t[0] = (t1 - t0) / 16; t[1] = (t2 - t1) / 16; t[2] = (t3 - t2) / 16; }
}

__global__ void noise_kernel(int* p) { if (p && threadIdx.x == 1000) *p = 1; }

int main() {
    std::vector<double> h(64 * 16), ref(64 * 16);
    srand(1);
    double m[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) m[i][j] = (double)rand() / RAND_MAX - 0.5;
    double spd[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 1.0 : 0.0; for (int k = 0; k < 16; ++k) s += m[i][k] * m[j][k] / 16; spd[i][j] = s; }
    for (int l = 0; l < 64; ++l) for (int c = 0; c < 16; ++c) h[l * 16 + c] = l < 16 ? (c <= l ? spd[l][c] : 0.0) : (double)rand() / RAND_MAX - 0.5;
    // host reference: L = chol(spd) in long double; panel rows: x L^-T
    long double L[16][16] = {};
    for (int j = 0; j < 16; ++j) { long double s = spd[j][j]; for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k]; L[j][j] = sqrtl(s);
        for (int i = j + 1; i < 16; ++i) { long double t = spd[i][j]; for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k]; L[i][j] = t / L[j][j]; } }
    for (int l = 0; l < 64; ++l) for (int c = 0; c < 16; ++c) {
        if (l < 16) { ref[l * 16 + c] = c <= l ? (double)L[l][c] : 0.0; continue; }
        long double t = h[l * 16 + c]; for (int k = 0; k < c; ++k) t -= (long double)ref[l * 16 + k] * L[c][k]; ref[l * 16 + c] = (double)(t / L[c][c]); }
    double *din, *dout; long long* dt;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8); hipMalloc(&dt, 64 * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    long long ht[8]; std::vector<double> o(64 * 16);
    chain_probe<<<1, 64>>>(dout, dt); hipDeviceSynchronize(); hipMemcpy(ht, dt, 24, hipMemcpyDeviceToHost);
    printf("dependent chain, shader clocks per op (x1000 iterations of 16): v_fma_f64 %.1f  v_rsq_f64+add %.1f  readlane pair + fma %.1f\n", ht[0] / 1000.0, ht[1] / 1000.0, ht[2] / 1000.0);
#define RUN(V)                                                                                                     \
    { bench<V><<<1, 256>>>(din, dout, dt, 2000); hipDeviceSynchronize();                                            \
      hipMemcpy(ht, dt, 24, hipMemcpyDeviceToHost); hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost); \
      double err = 0; for (int l = 0; l < 64; ++l) for (int c = 0; c < 16; ++c) if (l >= 16 || c <= l) err = fmax(err, fabs(o[l * 16 + c] - ref[l * 16 + c])); \
      printf("variant %d: %lld clock64 ticks, %lld ns per 16-column step (%.0f ns per column), FIRST step of the launch (cold instruction cache) %lld ticks, max abs err vs long-double Cholesky %.2e\n", V, ht[0], ht[1], ht[1] / 16.0, ht[2], err); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    // Does a DISPATCH elsewhere on the GPU cost a running wave its instruction cache?  The same hot loop, 40000 repetitions, while the
    // host launches empty kernels: (a) none, (b) back to back on one other stream, (c) on two streams that wait for each other's events
    // (every launch behind a barrier packet: the form the factorisation's look-ahead produces).
    {
        hipStream_t sa, sb, sm; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
        hipStreamCreateWithFlags(&sm, hipStreamNonBlocking);
        hipEvent_t ea, eb; hipEventCreateWithFlags(&ea, hipEventDisableTiming); hipEventCreateWithFlags(&eb, hipEventDisableTiming);
        for (int mode = 0; mode < 3; ++mode) {
            bench<3><<<1, 256, 0, sm>>>(din, dout, dt, 40000);
            int launched = 0;
            while (hipStreamQuery(sm) == hipErrorNotReady && mode > 0) {
                if (mode == 1) { noise_kernel<<<1, 64, 0, sa>>>(nullptr); ++launched; }
                else { noise_kernel<<<1, 64, 0, sa>>>(nullptr); hipEventRecord(ea, sa); hipStreamWaitEvent(sb, ea, 0);
                       noise_kernel<<<1, 64, 0, sb>>>(nullptr); hipEventRecord(eb, sb); hipStreamWaitEvent(sa, eb, 0); launched += 2; }
            }
            hipDeviceSynchronize(); hipMemcpy(ht, dt, 24, hipMemcpyDeviceToHost);
            printf("dispatch noise mode %d (%s): %lld clock64 ticks per step, %d kernels launched meanwhile\n", mode,
                   mode == 0 ? "quiet" : (mode == 1 ? "one stream, back to back" : "two streams, event ping-pong"), ht[0], launched);
        }
    }
#define PRUN(V, W)                                                                                                  \
    { phase_bench<V, W><<<1, 768>>>(din, dout, dt, 400); hipDeviceSynchronize(); hipMemcpy(ht, dt, 32, hipMemcpyDeviceToHost); \
      printf("phase V%d W%d: LDS reads %lld ns, 16-column loop %lld ns, LDS writes + drain %lld ns, barrier %lld ns\n", V, W, ht[0], ht[1], ht[2], ht[3]); }
#define SRUN(B)                                                                                                     \
    { struct_bench<B><<<1, 768>>>(din, dout, dt, 400); hipDeviceSynchronize(); hipMemcpy(ht, dt, 16, hipMemcpyDeviceToHost); \
      printf("leaf-3 structure, companions %s: wave 0's step (LDS loads + loop + stores) %lld shader clocks per repetition, first %lld\n", \
             B == 0 ? "idle at the barrier" : (B == 1 ? "running LDS-fed MFMA tiles" : "running the same step"), ht[0], ht[1]); }
    SRUN(0) SRUN(1) SRUN(2)
    return 0;
}
