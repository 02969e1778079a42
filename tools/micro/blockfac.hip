// Micro-benchmark of the leaf's 16 x 16 diagonal-block factor (leaf.hip: leaf3_factor_diag) against a blocked form whose rank-4
// trailing updates run on the matrix pipe:
//   old: one row per lane, sixteen columns of (pivot broadcast, reciprocal, up to fifteen broadcast + fma pairs): ~1250 instructions
//   new: the block lives in MFMA accumulator layout (by symmetry a lane group holds four COLUMNS of every row), four panels of four
//        columns are eliminated with broadcasts inside the panel only, and each panel's rank-4 update of what is right of it is ONE
//        v_mfma_f64_16x16x4 (plus one for the identity rows that become D^-1): ~500 instructions
// Both read the block from LDS (lower triangle valid) and leave L (rows) and D^-1 (transposed) in LDS; checked against a long-double
// Cholesky on the host.    hipcc --offload-arch=gfx950 -O3 -o tools/micro/blockfac tools/micro/blockfac.hip && tools/micro/blockfac
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define LD 130
#define DLD 18
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

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
__device__ __forceinline__ long long rdclk() {
    long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}

// ---- the shipped form (leaf.hip, leaf3_factor_diag) ----
__device__ __forceinline__ int factor_old(double* D, double* Dv, int lane) {
    const int fr = lane & 15;
    const bool is_diag = lane < 16, is_ident = lane >= 16 && lane < 32;
    double row[16], piv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = D[fr * LD + c];
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(row[c]));
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = is_diag ? row[c] : ((is_ident && c == fr) ? 1.0 : 0.0);
    int first_bad = 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        piv[c] = bcast_lane(row[c], c);
        first_bad = (piv[c] > 0.0) ? first_bad : min(first_bad, c);
        const double w = row[c] * recip(piv[c]);
#pragma unroll
        for (int k = c + 1; k < 16; ++k) row[k] -= row[c] * bcast_lane(w, k);
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] *= inv_sqrt(piv[c]);
    if (first_bad == 16) {
        if (is_diag) {
#pragma unroll
            for (int c = 0; c < 16; ++c) D[fr * LD + c] = row[c];
        } else if (is_ident) {
#pragma unroll
            for (int c = 0; c < 16; ++c) Dv[c * DLD + fr] = row[c];
        }
    }
    return first_bad;
}

// ---- blocked form ----
// lane = (g, i), g = lane >> 4, i = lane & 15.  t[r] = T[i][4g + r] (row i, the four columns of group g), z[r] = Z[i][4g + r] (row i of
// the identity block).  As the C operand of v_mfma_f64_16x16x4 register r of lane (g, i) is C[g + 4r][i], so MFMA row rho = g + 4r
// stands for matrix column 4g + r; the A operand's lane (k, rho) therefore takes the multiplier of matrix row 4 (rho & 3) + (rho >> 2).
#define SLD 6   // staging rows of four doubles, padded to six (16-byte aligned, conflict-free b64 reads)
__device__ __forceinline__ int factor_blk(double* D, double* Dv, double* stage, int lane) {
    const int g = lane >> 4, i = lane & 15;
    double* Sw = stage;                 // [16][SLD]  -w of the panel (A operand)
    double* Sa = stage + 16 * SLD;      // [16][SLD]  the panel's unscaled columns (B operand, T)
    double* Sz = stage + 32 * SLD;      // [16][SLD]  ... of the identity rows (B operand, Z)
    double* Sp = stage + 48 * SLD;      // [16] pivots
    d4 t, z;
    {
        const d2 lo = *reinterpret_cast<const d2*>(D + i * LD + 4 * g), hi = *reinterpret_cast<const d2*>(D + i * LD + 4 * g + 2);
        t[0] = lo[0]; t[1] = lo[1]; t[2] = hi[0]; t[3] = hi[1];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = (4 * g + r == i) ? 1.0 : 0.0;
    const int jrow = 4 * (i & 3) + (i >> 2);        // matrix row whose multiplier this lane supplies as A operand
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (g == p) {                               // the group that holds this panel's columns; the others wait for the update
            double wn[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int gc = 4 * p + c;
                const double piv = bcast_lane(t[c], 16 * p + gc);
                wn[c] = -t[c] * recip(piv);
#pragma unroll
                for (int k = c + 1; k < 4; ++k) {
                    const double m = bcast_lane(wn[c], 16 * p + 4 * p + k);
                    t[k] = __builtin_fma(t[c], m, t[k]);
                    z[k] = __builtin_fma(z[c], m, z[k]);
                }
            }
            if (p < 3) {
                *reinterpret_cast<d2*>(Sw + i * SLD) = d2{wn[0], wn[1]};
                *reinterpret_cast<d2*>(Sw + i * SLD + 2) = d2{wn[2], wn[3]};
                *reinterpret_cast<d2*>(Sa + i * SLD) = d2{t[0], t[1]};
                *reinterpret_cast<d2*>(Sa + i * SLD + 2) = d2{t[2], t[3]};
                *reinterpret_cast<d2*>(Sz + i * SLD) = d2{z[0], z[1]};
                *reinterpret_cast<d2*>(Sz + i * SLD + 2) = d2{z[2], z[3]};
            }
        }
        if (p == 3) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double aw = Sw[jrow * SLD + g];
        const double bt = Sa[i * SLD + g], bz = Sz[i * SLD + g];
        aw = (jrow >= 4 * p + 4) ? aw : 0.0;                    // finished columns (and this panel's) keep their values
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, bt, t, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, bz, z, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();                        // the staging area is re-used by the next panel
    }
    // pivots = the diagonal: lane (g, i) with i >> 2 == g holds T[i][i] in register i & 3
    {
        double d = t[0];
        d = ((i & 3) == 1) ? t[1] : d;
        d = ((i & 3) == 2) ? t[2] : d;
        d = ((i & 3) == 3) ? t[3] : d;
        if ((i >> 2) == g) Sp[i] = d;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const d2 p01 = *reinterpret_cast<const d2*>(Sp + 4 * g), p23 = *reinterpret_cast<const d2*>(Sp + 4 * g + 2);
    int first_bad = 16;
    const bool bad = !(p01[0] > 0.0) || !(p01[1] > 0.0) || !(p23[0] > 0.0) || !(p23[1] > 0.0);
    if (__builtin_amdgcn_ballot_w64(bad) != 0) {     // a non-positive (or NaN) pivot: the first one's column
#pragma unroll 1
        for (int c = 15; c >= 0; --c) if (!(Sp[c] > 0.0)) first_bad = c;
        return first_bad;
    }
    const double rs0 = inv_sqrt(p01[0]), rs1 = inv_sqrt(p01[1]), rs2 = inv_sqrt(p23[0]), rs3 = inv_sqrt(p23[1]);
    *reinterpret_cast<d2*>(D + i * LD + 4 * g) = d2{t[0] * rs0, t[1] * rs1};
    *reinterpret_cast<d2*>(D + i * LD + 4 * g + 2) = d2{t[2] * rs2, t[3] * rs3};
    Dv[(4 * g + 0) * DLD + i] = z[0] * rs0;
    Dv[(4 * g + 1) * DLD + i] = z[1] * rs1;
    Dv[(4 * g + 2) * DLD + i] = z[2] * rs2;
    Dv[(4 * g + 3) * DLD + i] = z[3] * rs3;
    return first_bad;
}

template <int V>
__global__ __launch_bounds__(768) void bench(const double* __restrict__ in, double* __restrict__ outL, double* __restrict__ outV, long long* t,
                                             int reps) {
    __shared__ double S[16 * LD + 16 * DLD + 64 * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* D = S; double* Dv = S + 16 * LD; double* stage = Dv + 16 * DLD;
    long long acc = 0, first = 0;
    int fb = 0;
    for (int rep = 0; rep < reps; ++rep) {
        __syncthreads();
        if (tid < 256) D[(tid >> 4) * LD + (tid & 15)] = in[tid];     // lower triangle valid, upper = finite junk
        __syncthreads();
        if (wave == 0) {
            const long long c0 = rdclk();
            fb = V == 0 ? factor_old(D, Dv, lane) : factor_blk(D, Dv, stage, lane);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const long long dt = rdclk() - c0;
            acc += dt;
            if (rep == 0) first = dt;
        }
    }
    __syncthreads();
    if (tid == 0) { t[0] = acc / reps; t[1] = first; t[2] = fb; }
    if (tid < 256) { outL[tid] = D[(tid >> 4) * LD + (tid & 15)]; outV[tid] = Dv[(tid >> 4) * DLD + (tid & 15)]; }
}

int main() {
    std::vector<double> h(256), L(256, 0.0), V(256, 0.0);
    srand(3);
    double m[16][16], spd[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) m[i][j] = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 0.3 : 0.0; for (int k = 0; k < 16; ++k) s += m[i][k] * m[j][k] / 16; spd[i][j] = s; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = j <= i ? spd[i][j] : 7.5 + i - j;   // junk above the diagonal
    long double Lr[16][16] = {}, Vr[16][16] = {};
    for (int j = 0; j < 16; ++j) { long double s = spd[j][j]; for (int k = 0; k < j; ++k) s -= Lr[j][k] * Lr[j][k]; Lr[j][j] = sqrtl(s);
        for (int i = j + 1; i < 16; ++i) { long double tt = spd[i][j]; for (int k = 0; k < j; ++k) tt -= Lr[i][k] * Lr[j][k]; Lr[i][j] = tt / Lr[j][j]; } }
    for (int j = 0; j < 16; ++j) { Vr[j][j] = 1 / Lr[j][j]; for (int i = j + 1; i < 16; ++i) { long double s = 0; for (int k = j; k < i; ++k) s += Lr[i][k] * Vr[k][j]; Vr[i][j] = -s / Lr[i][i]; } }
    double *din, *dL, *dV; long long* dt;
    hipMalloc(&din, 2048); hipMalloc(&dL, 2048); hipMalloc(&dV, 2048); hipMalloc(&dt, 64);
    hipMemcpy(din, h.data(), 2048, hipMemcpyHostToDevice);
    long long ht[3];
    for (int v = 0; v < 2; ++v)
        for (int nw : {4, 12}) {
            if (v == 0) bench<0><<<1, 64 * nw>>>(din, dL, dV, dt, 400); else bench<1><<<1, 64 * nw>>>(din, dL, dV, dt, 400);
            hipDeviceSynchronize();
            hipMemcpy(ht, dt, 24, hipMemcpyDeviceToHost); hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost); hipMemcpy(V.data(), dV, 2048, hipMemcpyDeviceToHost);
            double eL = 0, eV = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) { eL = fmax(eL, fabs(L[i * 16 + j] - (double)Lr[i][j])); eV = fmax(eV, fabs(V[i * 16 + j] - (double)Vr[i][j])); }
            printf("%s, %2d waves resident: %5lld shader clocks per 16 x 16 factor incl. LDS in/out (first %5lld), first_bad %lld, max |L - ref| %.2e, max |D^-1 - ref| %.2e\n",
                   v == 0 ? "row-per-lane (shipped)" : "blocked, MFMA updates ", nw, ht[0], ht[1], ht[2], eL, eV);
        }
    return 0;
}
