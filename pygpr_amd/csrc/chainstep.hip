// Flag-coupled chain of the Cholesky's latency-bound tail.
//
// In the classic chain every 128-column step is three dependent launches on one stream (U: bring the block column up to date,
// leaf: factor + invert the diagonal tile, T: solve the rows below), each behind the previous one's completion and a launch gap:
// 62 us per step at n <= 8192, of which the leaf is 34.  Here a step is TWO kernels on two streams that are both resident before
// their inputs exist and hand over through flags in device memory (agent-scope, cdna_hip_programming.md guideline 16):
//
//   leaf stream : leaf(k)   waits diag[k] == NCRIT, factors tile (k, k), stores L_kk and inv_k, releases done[k]
//   rows stream : rows(k)   one workgroup per 16 (block row k+1) / 32 rows below tile (k, k):
//                   before done[k]:  acc = X[rows, o0:k0] X[blk k+1, o0:k0]^T      (the next block column's update by the panel's
//                                    earlier columns: everything it reads is final since rows(k-1) ended)
//                   after  done[k]:  X[rows, blk k] = A[rows, blk k] inv_k^T        (T)
//                                    the NCRIT workgroups owning block row k+1 publish their X rows (brow[k] += 1)
//                   after  brow[k] == NCRIT:  acc += X[rows, blk k] X[blk k+1, blk k]^T;  A[rows, blk k+1] -= acc   (U)
//                                    the NCRIT workgroups owning tile (k+1, k+1) publish it (diag[k+1] += 1)
//
// so between two leaves lie one K = 128 solve, one K = 128 product and three flag hops instead of two full launches, and the
// long part of the update runs while the leaf does.  The scheme NEEDS the two queues to run concurrently: where kernels are
// executed one at a time in an order of the tool's choosing (a counter-collecting profiler), leaf(k+1) can be started before the
// rows(k) it waits for.  pg_create probes for that once and keeps the classic chain there (capi.hip); beyond that every spin is
// bounded: on expiry a sticky timeout word ends all later spins at once and *info is set to -1 (the caller sees a failed
// factorisation, not a hang -- seen exactly once, under rocprofv3 --pmc before the probe existed).
// Every kernel performs all of its signals whatever it saw (bad pivot, timeout), so nothing downstream waits for ever.
#include "chainstep.h"

#define NB 128
#define CS_NTH 512
#define CS_KC 32
#define CS_CLD (CS_KC + 2)
#define CS_XLD (NB + 2)

#define RLX_AGENT PG_RLX_AGENT

// Whole workgroup: wait until *flag >= want, then make the publisher's bytes loadable (one lane's agent-scope acquire drops this
// CU's L1 lines; its wait holds the barrier for the invalidate).
__device__ __forceinline__ void cs_wg_wait(int* flag, int want, const CsWait& tmo, int* info) {
    if (threadIdx.x == 0) {
        if (!cs_spin_ge(flag, want, tmo)) atomicCAS(info, 0, -1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// Whole workgroup, after its write-through (sc1) stores: every storing wave drains, one lane adds to the counter.
__device__ __forceinline__ void cs_wg_signal(int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1, RLX_AGENT);
}

template <typename T> __device__ __forceinline__ void st_wt(T* p, T v);   // write-through store (sc1)
template <> __device__ __forceinline__ void st_wt<double>(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), RLX_AGENT);
}
template <> __device__ __forceinline__ void st_wt<float>(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), RLX_AGENT);
}

template <typename T> __device__ __forceinline__ T ld_wt(const T* p);   // load past this CU's L1 (sc1)
template <> __device__ __forceinline__ double ld_wt<double>(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), RLX_AGENT));
}
template <> __device__ __forceinline__ float ld_wt<float>(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), RLX_AGENT));
}

// Whole workgroup: wait for *flag >= want WITHOUT an acquire: for workgroups whose every later load of handed-off bytes is ld_wt
__device__ __forceinline__ void cs_wg_wait_wt(int* flag, int want, const CsWait& tmo, int* info) {
    if (threadIdx.x == 0) {
        if (!cs_spin_ge(flag, want, tmo)) atomicCAS(info, 0, -1);
    }
    __syncthreads();
}

// acc[NJ] += Aop[ROWS x K] B[128 x K]^T for this wave's 16 x 16 NJ tile.  RG = 2: 32 rows, waves as 2 row groups x 4 column
// groups of 32 columns (NJ = 2);  RG = 1: 16 rows, 8 column groups of 16 columns (NJ = 1).
//   A_LDS false: A rows from global (Ag, lda), staged;  true: A = the workgroup's LDS tile Xl[ROWS][CS_XLD], columns 0 .. K-1
//   B from global (Bg, ldb), staged in K chunks of 32 with the next chunk's loads in flight during the MFMAs.
//   lower_b: B[n][k] = 0 for k > n (the leaf's inverse): chunks past the wave's last column are skipped.
template <typename T, int RG, bool A_LDS>
__device__ __forceinline__ void cs_mm(typename Mfma<T>::acc_t (&acc)[RG], const T* __restrict__ Ag, long lda, const T* Xl,
                                      const T* __restrict__ Bg, long ldb, int K, bool lower_b, T* As, T* Bs) {
    constexpr int ROWS = 16 * RG, NJ = RG, WC = 16 * NJ;      // rows of the workgroup, column tiles and columns per wave
    constexpr int VE = 16 / sizeof(T);
    constexpr int VPR = CS_KC / VE;                              // 16-byte vectors per row of a chunk
    constexpr int NVB = NB * VPR, NVA = ROWS * VPR;
    constexpr int UB = (NVB + CS_NTH - 1) / CS_NTH, UA = (NVA + CS_NTH - 1) / CS_NTH;
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rg = RG == 2 ? wave >> 2 : 0, cg = RG == 2 ? wave & 3 : wave, fr = lane & 15, fk = lane >> 4;
    // TWO chunks in flight (two named register sets, the loop unrolled by two): with one, a chunk's loads are issued when the
    // previous chunk starts its 0.2-0.4 us of MFMAs and the loop runs at one load latency per chunk -- 48 us for the 896-deep
    // window of a 512-column panel's last step, longer than the leaf it is meant to hide behind.  K is a multiple of 128.
    vec_t vb0[UB], va0[UA], vb1[UB], va1[UA];
    auto issue = [&](vec_t (&vb)[UB], vec_t (&va)[UA], int kc) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int v = tid + u * CS_NTH;
            if (v < NVB) vb[u] = *reinterpret_cast<const vec_t*>(Bg + (long)(v / VPR) * ldb + kc + (v % VPR) * VE);
        }
        if (!A_LDS) {
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int v = tid + u * CS_NTH;
                if (v < NVA) va[u] = *reinterpret_cast<const vec_t*>(Ag + (long)(v / VPR) * lda + kc + (v % VPR) * VE);
            }
        }
    };
    auto stage = [&](vec_t (&vb)[UB], vec_t (&va)[UA]) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int v = tid + u * CS_NTH;
            if (v < NVB) {
                T* d = Bs + (v / VPR) * CS_CLD + (v % VPR) * VE;
#pragma unroll
                for (int e = 0; e < VE; ++e) d[e] = vb[u][e];
            }
        }
        if (!A_LDS) {
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int v = tid + u * CS_NTH;
                if (v < NVA) {
                    T* d = As + (v / VPR) * CS_CLD + (v % VPR) * VE;
#pragma unroll
                    for (int e = 0; e < VE; ++e) d[e] = va[u][e];
                }
            }
        }
    };
    auto compute = [&](int kc) {
        if (lower_b && kc > WC * cg + WC - 1) return;             // wave-uniform; the barriers are outside
#pragma unroll
        for (int ks = 0; ks < CS_KC / 4; ++ks) {
            T b[NJ];
            const T a = A_LDS ? Xl[(16 * rg + fr) * CS_XLD + kc + 4 * ks + fk] : As[(16 * rg + fr) * CS_CLD + 4 * ks + fk];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = Bs[(WC * cg + 16 * j + fr) * CS_CLD + 4 * ks + fk];
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = Mfma<T>::run(a, b[j], acc[j]);
        }
    };
    if (K > 0) {
        issue(vb0, va0, 0);
        issue(vb1, va1, CS_KC);
    }
    for (int kc = 0; kc < K; kc += 2 * CS_KC) {
        __syncthreads();                                          // the previous chunk's readers are done
        stage(vb0, va0);
        __syncthreads();
        if (kc + 2 * CS_KC < K) issue(vb0, va0, kc + 2 * CS_KC);
        compute(kc);
        __syncthreads();
        stage(vb1, va1);
        __syncthreads();
        if (kc + 3 * CS_KC < K) issue(vb1, va1, kc + 3 * CS_KC);
        compute(kc + CS_KC);
    }
}

// The K = 128 stages of the workgroups the next leaf waits for (16 rows, one 16 x 16 tile per wave): with 8 MFMAs per wave and
// chunk the staged loop above is a chain of load latencies (a chunk's loads are issued when the previous chunk starts computing:
// 0.2 us of MFMA against 1.5-2 us of latency, four times over).  Here ALL of B (128 x 128) is in flight at once, 16 vectors per
// thread, L1-bypassing (published by another resident kernel, no acquire in front), and goes through LDS chunk by chunk as it
// lands.  lower_b: B[n][k] = 0 for k > n.
template <typename T>
__device__ __forceinline__ void cs_mm_k128(typename Mfma<T>::acc_t& acc, const T* Xl, const T* __restrict__ Bg, long ldb, bool lower_b,
                                           T* Bs, int c_lo = 0, int c_hi = NB / CS_KC, int n_lo = 0, int n_hi = NB) {
    // K chunks [c_lo, c_hi) of 32 columns, rows [n_lo, n_hi) of B (= the output columns of the waves 16 cg in [n_lo, n_hi)): the
    // two-phase hand-over runs this twice per product, once on the part of the operand that exists early
    constexpr int VE = 16 / sizeof(T);
    constexpr int VPR = CS_KC / VE, NVB = NB * VPR, UB = (NVB + CS_NTH - 1) / CS_NTH, NCH = NB / CS_KC;
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Bg), 0, (int)((127 * ldb + NB) * sizeof(T)), 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, cg = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const bool mine = 16 * cg >= n_lo && 16 * cg < n_hi;          // wave-uniform: this wave's output tile takes part
    vec_t vb[NCH][UB];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int v = tid + u * CS_NTH, row = v / VPR;
            // a lower-triangular B has nothing in chunk c for the rows above it
            if (v < NVB && c >= c_lo && c < c_hi && row >= n_lo && row < n_hi && !(lower_b && row + 1 <= c * CS_KC))
                vb[c][u] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rB, (int)(((long)row * ldb + c * CS_KC + (v % VPR) * VE) * sizeof(T)), 0, 16));
            else
#pragma unroll
                for (int e = 0; e < VE; ++e) vb[c][u][e] = (T)0;
        }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c < c_lo || c >= c_hi) continue;                      // workgroup-uniform
        __syncthreads();
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int v = tid + u * CS_NTH;
            if (v < NVB) {
                T* d = Bs + (v / VPR) * CS_CLD + (v % VPR) * VE;
#pragma unroll
                for (int e = 0; e < VE; ++e) d[e] = vb[c][u][e];
            }
        }
        __syncthreads();
        if (!mine || (lower_b && c * CS_KC > 16 * cg + 15)) continue;        // wave-uniform
#pragma unroll
        for (int ks = 0; ks < CS_KC / 4; ++ks)
            acc = Mfma<T>::run(Xl[fr * CS_XLD + c * CS_KC + 4 * ks + fk], Bs[(16 * cg + fr) * CS_CLD + 4 * ks + fk], acc);
    }
}

template <typename T, int RG>
__device__ __forceinline__ void cs_rows_body(char* smem_raw, T* __restrict__ A, long lda, int row0, int o0, int k0, bool has_next,
                                             bool crit, const T* __restrict__ inv, int* done_k, int* brow_k, int* diag_next, int ncrit,
                                             const CsWait& tmo, int* info, int direct, long long* tlog, int* early_k = nullptr,
                                             int* browe_k = nullptr) {
    constexpr int ROWS = 16 * RG, NJ = RG, WC = 16 * NJ;
#define TL(i) do { if (tlog && threadIdx.x == 0) tlog[i] = wall_clock64(); } while (0)
    constexpr int VE = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    T* Xl = reinterpret_cast<T*>(smem_raw);                      // [ROWS][CS_XLD]
    T* As = Xl + ROWS * CS_XLD;                                  // [ROWS][CS_CLD]
    T* Bs = As + ROWS * CS_CLD;                                  // [128][CS_CLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rg = RG == 2 ? wave >> 2 : 0, cg = RG == 2 ? wave & 3 : wave, fr = lane & 15;
    TL(11);                                                      // the workgroup is running
    T* Arow = A + (long)row0 * lda;                              // this workgroup's rows
    const T* Brow = A + (long)(k0 + NB) * lda;                   // block row k+1
    typename Mfma<T>::acc_t acc[NJ], cin[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[j][r] = (T)0;
            cin[j][r] = (T)0;
        }
    // this workgroup's block of column k -> LDS (its values are final except for the solve), and the block of column k+1
    for (int v = tid; v < ROWS * NB / VE; v += CS_NTH) {
        const int r = v / (NB / VE), c = (v % (NB / VE)) * VE;
        const vec_t x = *reinterpret_cast<const vec_t*>(Arow + (long)r * lda + k0 + c);
#pragma unroll
        for (int e = 0; e < VE; ++e) Xl[r * CS_XLD + c + e] = x[e];
    }
    if (has_next) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                cin[j][r] = Arow[(long)(16 * rg + Mfma<T>::row(lane, r)) * lda + k0 + NB + WC * cg + 16 * j + fr];
        // the panel's earlier columns
        cs_mm<T, RG, false>(acc, Arow + o0, lda, Xl, Brow + o0, lda, k0 - o0, false, As, Bs);
    }
    // T: rows x inv_k^T
    typename Mfma<T>::acc_t t[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[j][r] = (T)0;
    TL(4);
    // Two-phase hand-over (the eight workgroups the next leaf waits for).  The third leaf form raises *early_k once columns 0..63
    // of L_kk and rows 0..63 of its inverse are in memory -- three of its eight steps and its tail before *done_k.  With
    //     X1 = A1 inv11^T,   W = A2 - X1 L21^T,   X2 = W inv22^T          (A = [A1 A2], 64 columns each)
    // X1, its exchange, W and X1's share of the update of column k+1 run WHILE THE LEAF DOES; behind *done_k are left a 64-deep
    // solve and a 64-deep update: half the K chunks (each a barrier pair) of the single-phase form on the path to the next leaf.
    const bool two_phase = RG == 1 && direct && early_k != nullptr && has_next;
    if (two_phase) {
        constexpr int HC = NB / 2 / CS_KC, NC = NB / CS_KC;          // K chunks: of a half (2), of the whole (4)
        T* Xn = Bs + NB * CS_CLD;                                    // [16][CS_XLD] the solved rows (the 32-row workgroups' share of LDS is there)
        const T* Akk = A + (long)k0 * lda + k0;
        cs_wg_wait_wt(early_k, 1, tmo, info);
        cs_mm_k128<T>(t[0], Xl, inv, NB, true, Bs, 0, HC, 0, NB / 2);            // X1 (waves 0..3)
        if (cg < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = Mfma<T>::row(lane, r), cc = 16 * cg + fr;
                Xn[rr * CS_XLD + cc] = t[0][r];
                st_wt<T>(Arow + (long)rr * lda + k0 + cc, t[0][r]);
            }
        }
        cs_wg_signal(browe_k);                                       // (its barrier also orders the Xn stores)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[0][r] = (T)0;
        cs_mm_k128<T>(t[0], Xn, Akk, lda, false, Bs, 0, HC, NB / 2, NB);         // X1 L21^T (waves 4..7)
        if (cg >= 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Xl[Mfma<T>::row(lane, r) * CS_XLD + 16 * cg + fr] -= t[0][r];      // W over A2
        }
        cs_wg_wait_wt(browe_k, ncrit, tmo, info);
        cs_mm_k128<T>(acc[0], Xn, Brow + k0, lda, false, Bs, 0, HC, 0, NB);      // acc += X1 X1[block row k+1]^T
#pragma unroll
        for (int r = 0; r < 4; ++r) t[0][r] = (T)0;
        cs_wg_wait_wt(done_k, 1, tmo, info);
        TL(5);
        cs_mm_k128<T>(t[0], Xl, inv, NB, true, Bs, HC, NC, NB / 2, NB);          // X2 = W inv22^T (waves 4..7)
        TL(6);
        if (cg >= 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = Mfma<T>::row(lane, r), cc = 16 * cg + fr;
                Xn[rr * CS_XLD + cc] = t[0][r];
                st_wt<T>(Arow + (long)rr * lda + k0 + cc, t[0][r]);
            }
        }
        cs_wg_signal(brow_k);
        TL(7);
        cs_wg_wait_wt(brow_k, ncrit, tmo, info);
        TL(8);
        cs_mm_k128<T>(acc[0], Xn, Brow + k0, lda, false, Bs, HC, NC, 0, NB);     // acc += X2 X2[block row k+1]^T
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            T* g = Arow + (long)Mfma<T>::row(lane, r) * lda + k0 + NB + 16 * cg + fr;
            st_wt<T>(g, cin[0][r] - acc[0][r]);
        }
        TL(9);
        cs_wg_signal(diag_next);
        TL(10);
        return;
    }
    if (RG == 1 && direct) {
        cs_wg_wait_wt(done_k, 1, tmo, info);
        TL(5);
        cs_mm_k128<T>(t[0], Xl, inv, NB, true, Bs);
    } else {
        cs_wg_wait(done_k, 1, tmo, info);
        cs_mm<T, RG, true>(t, nullptr, 0, Xl, inv, NB, NB, true, As, Bs);
    }
    TL(6);
    __syncthreads();                                             // every wave has read its rows of Xl
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = 16 * rg + Mfma<T>::row(lane, r), cc = WC * cg + 16 * j + fr;
            Xl[rr * CS_XLD + cc] = t[j][r];
            T* g = Arow + (long)rr * lda + k0 + cc;
            if (crit) st_wt<T>(g, t[j][r]);
            else *g = t[j][r];
        }
    if (crit) cs_wg_signal(brow_k);
    TL(7);
    if (!has_next) return;
    if (RG == 1 && direct) {                                     // (the barrier inside also orders the Xl stores above)
        cs_wg_wait_wt(brow_k, ncrit, tmo, info);
        TL(8);
        cs_mm_k128<T>(acc[0], Xl, Brow + k0, lda, false, Bs);
    } else {
        cs_wg_wait(brow_k, ncrit, tmo, info);
        cs_mm<T, RG, true>(acc, nullptr, 0, Xl, Brow + k0, lda, NB, false, As, Bs);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            T* g = Arow + (long)(16 * rg + Mfma<T>::row(lane, r)) * lda + k0 + NB + WC * cg + 16 * j + fr;
            const T v = cin[j][r] - acc[j][r];
            if (crit) st_wt<T>(g, v);
            else *g = v;
        }
    TL(9);
    if (crit) cs_wg_signal(diag_next);
    TL(10);
#undef TL
}

// grid = 8 + (m - 128) / 32 workgroups (m = rows below tile (k, k)): the first eight own 16 rows each of block row k+1 -- what
// the next leaf waits for: half the rows, half the MFMA time on the critical path -- the others 32 rows each.  At most 77 KB of
// LDS and 128 VGPRs: a workgroup fits beside one resident 128 x 128 GEMM block of the trailing update.
// (64-row workgroups for the rows further down halve the re-reads of block row k+1 but need 119 KB: they would only ever start
// on an empty CU.)
#define CS_NCRIT 8
template <typename T>
__global__ __launch_bounds__(CS_NTH, 4) void pg_rowstep_kernel(T* __restrict__ A, long lda, int o0, int k0, int has_next,
                                                               const T* __restrict__ inv, int* done_k, int* brow_k, int* diag_next,
                                                               CsWait tmo, int* info, int direct, long long* tlog, int* early_k,
                                                               int* browe_k, int rows16, CsBatch cb) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (cb.nexp > 1) {      // blockIdx.y = expert
        const long e = blockIdx.y;
        A += e * cb.eA; inv += e * cb.eInv; info += e;
        done_k += e * cb.eF; brow_k += e * cb.eF; diag_next += e * cb.eF; tmo.tmo += e * cb.eF;
        if (early_k) early_k += e * cb.eF;
        if (browe_k) browe_k += e * cb.eF;
        tlog = nullptr;
    }
    const int w = blockIdx.x;
    if (w < CS_NCRIT)
        cs_rows_body<T, 1>(smem_raw, A, lda, k0 + NB + 16 * w, o0, k0, has_next != 0, true, inv, done_k, brow_k, diag_next, CS_NCRIT,
                           tmo, info, direct, w == 0 ? tlog : nullptr, early_k, browe_k);
    else if (rows16)   // experiment: 16-row workgroups below the critical ones too (twice as many, each half as long)
        cs_rows_body<T, 1>(smem_raw, A, lda, k0 + 2 * NB + 16 * (w - CS_NCRIT), o0, k0, has_next != 0, false, inv, done_k, brow_k,
                           diag_next, CS_NCRIT, tmo, info, rows16 == 2 ? direct : 0, nullptr);
    else
        cs_rows_body<T, 2>(smem_raw, A, lda, k0 + 2 * NB + 32 * (w - CS_NCRIT), o0, k0, has_next != 0, false, inv, done_k, brow_k,
                           diag_next, CS_NCRIT, tmo, info, direct, nullptr);
}

__global__ void pg_flagset_kernel(int* flag, int value, long eF) { __hip_atomic_store(flag + blockIdx.x * eF, value, RLX_AGENT); }

template <typename T>
int pg_rowstep(hipStream_t st, T* A, long lda, int n, int o0, int k0, int has_next, const T* inv, int* done_k, int* brow_k,
               int* diag_next, const CsWait& tmo, int* info, int* early_k, int* browe_k, int allow_tlog, const CsBatch* cbp) {
    const CsBatch cb = cbp ? *cbp : CsBatch{1, 0, 0, 0};
    const int m = n - k0 - NB;
    if (m <= 0 || m % NB) { pg_set_error("pg_rowstep: %d rows below the tile", m); return -2; }
    const size_t lds = (size_t)(32 * CS_XLD + 32 * CS_CLD + NB * CS_CLD) * sizeof(T);
    static bool attr_done = false;
    if (!attr_done) {
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pg_rowstep_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    static const int direct = getenv("PG_CS_K128") ? atoi(getenv("PG_CS_K128")) : 1;
    static const int rows16_env = getenv("PG_CS_ROWS16") ? atoi(getenv("PG_CS_ROWS16")) : 4096;   // rows at or below which every workgroup takes 16 rows
    static const int rows16_direct = getenv("PG_CS_ROWS16_DIRECT") ? atoi(getenv("PG_CS_ROWS16_DIRECT")) : 1;
    const int rows16 = (rows16_env > 0 && m <= rows16_env) ? (rows16_direct ? 2 : 1) : 0;
    hipLaunchKernelGGL(pg_rowstep_kernel<T>, dim3(8 + (m - NB) / (rows16 ? 16 : 32), cb.nexp), dim3(CS_NTH), lds, st, A, lda, o0, k0, has_next, inv, done_k,
                       brow_k, diag_next, tmo, info, direct,
                       // (the in-kernel time log lives behind the flag words of a factorisation's work buffer: never for a caller that
                       //  brings its own small flag array, pg_rowstep_raw)
                       (allow_tlog && getenv("PG_CS_TLOG")) ? reinterpret_cast<long long*>(tmo.tmo) + 512 + 48 * (k0 / NB) : nullptr, direct ? early_k : nullptr,
                       browe_k, rows16, cb);
    PG_CHECK(hipGetLastError());
    return 0;
}
template int pg_rowstep<double>(hipStream_t, double*, long, int, int, int, int, const double*, int*, int*, int*, const CsWait&, int*, int*, int*, int, const CsBatch*);
template int pg_rowstep<float>(hipStream_t, float*, long, int, int, int, int, const float*, int*, int*, int*, const CsWait&, int*, int*, int*, int, const CsBatch*);

int pg_flagset(hipStream_t st, int* flag, int value, int nexp, long eF) {
    hipLaunchKernelGGL(pg_flagset_kernel, dim3(nexp), dim3(1), 0, st, flag, value, eF);
    PG_CHECK(hipGetLastError());
    return 0;
}

// One lane waiting for a flag nobody sets, with the budget the factorisation of an n x n matrix would use: what a wait of the coupled
// chain costs when its partner never comes (tests).  out[0] = 10 ns ticks waited, out[1] = 1 if the flag came, 0 on expiry.
__global__ void pg_spin_probe_kernel(int* flag, CsWait w, long long* out) {
    const long long t0 = (long long)wall_clock64();
    const bool ok = cs_spin_ge(flag, 1, w);
    out[0] = (long long)wall_clock64() - t0;
    out[1] = ok ? 1 : 0;
}
int pg_spin_probe_launch(hipStream_t st, int* flag, const CsWait& w, long long* out) {
    hipLaunchKernelGGL(pg_spin_probe_kernel, dim3(1), dim3(1), 0, st, flag, w, out);
    PG_CHECK(hipGetLastError());
    return 0;
}
