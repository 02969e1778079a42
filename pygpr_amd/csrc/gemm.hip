// MFMA GEMM core (see gemm.h).  gfx950 only.
//
// Block tile BM x BN, K tile BKT, 256 threads = 4 waves, each wave owning a sub-tile of 16x16 MFMA tiles:
//   128x128 (wave 64x64: 16 accumulators = 128 VGPRs in fp64) -- the throughput tile of every large product;
//   64x256 and 64x128 (in-place panel solves: one workgroup owns full rows), 64x64 (4x the workgroups for launches with
//   few large tiles); 32x64 and 32x128 with BKT = 32 (the Cholesky chain's skinny products: two waves per SIMD even when
//   only one 64-row workgroup per CU exists).
// Operands are staged global -> registers -> LDS (double buffered, one barrier per K tile); LDS rows are padded so that the
// fragment read `lane -> (row l&15, k l>>4)` is bank-conflict free:
//   K-contiguous operand : LDS [R][BKT+2]   (row stride 18 elements == 2 mod 32)
//   M/N-contiguous one   : LDS [BKT][R+16]  (row stride == 16 mod 32)
// One 16x16x4 fp64 MFMA is 2048 flop in 64 cycles per SIMD (78.6 TFLOP/s chip peak), so a 16-deep K tile of the 128x128
// block is 4096 MFMA cycles per wave against 8 ds_read_b64 per k-step: the loop is MFMA-bound by construction.
#include "gemm.h"
#include <cstdlib>
#include <cstdio>

#define BK 16

template <typename T, int R, bool KCONT, int BKT = BK, int NTH = 256> struct Stage {
    static constexpr int VE = 16 / sizeof(T);
    static constexpr int NV = (R * BKT / VE) / NTH;
    static_assert(NV >= 1, "fewer 16-byte vectors in the stage than threads");
    static constexpr int LDS_ELEMS = KCONT ? R * (BKT + 2) : BKT * (R + 16);
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    typedef T half_t __attribute__((ext_vector_type(VE / 2)));
    vec_t v[NV];
    unsigned boff[NV];   // this lane's BYTE offsets inside a K tile, relative to the tile's origin (loop invariant)

    // Addressing (round 4): a wave-uniform 64-bit origin (scalar registers, advanced by a scalar add per K tile) plus a 32-bit
    // per-lane byte offset -- the global_load saddr + voffset form.  With a 64-bit pointer per lane and vector (rounds 1-3) the fp64
    // 128 x 128 block needed 133 VGPRs: one base pointer was spilled and RELOADED from scratch at the top of every K tile, behind
    // an `s_waitcnt vmcnt(0)` in front of the tile's loads.  (The offsets stay below 2^32 for every leading dimension the library
    // can meet: 128 rows x ld x 8 bytes, ld < 4 M elements.)
    __device__ __forceinline__ void init(long ld, int tid) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int idx = tid + NTH * q;
            if (KCONT) {
                const int row = idx / (BKT / VE), kv = idx % (BKT / VE);
                boff[q] = (unsigned)(((long)row * ld + kv * VE) * (long)sizeof(T));
            } else {
                const int kr = idx / (R / VE), rv = idx % (R / VE);
                boff[q] = (unsigned)(((long)kr * ld + rv * VE) * (long)sizeof(T));
            }
        }
    }
    // origin of the K tile at k0 for the tile rows / columns starting at r0 (wave-uniform)
    static __device__ __forceinline__ const T* origin(const T* __restrict__ g, long ld, int r0, int k0) {
        return KCONT ? g + (long)r0 * ld + k0 : g + (long)k0 * ld + r0;
    }
    __device__ __forceinline__ void load(const T* __restrict__ org) {
        const char* b = reinterpret_cast<const char*>(org);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = *reinterpret_cast<const vec_t*>(b + (size_t)boff[q]);
    }
    __device__ __forceinline__ void store(T* s, int tid) const {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int idx = tid + NTH * q;
            if (KCONT) {
                const int row = idx / (BKT / VE), kv = idx % (BKT / VE);
                T* d = s + row * (BKT + 2) + kv * VE;   // 8-byte aligned for fp32, 16 for fp64
                half_t lo, hi;
#pragma unroll
                for (int e = 0; e < VE / 2; ++e) { lo[e] = v[q][e]; hi[e] = v[q][e + VE / 2]; }
                if (sizeof(T) == 8) {
                    *reinterpret_cast<vec_t*>(d) = v[q];
                } else {
                    *reinterpret_cast<half_t*>(d) = lo;
                    *reinterpret_cast<half_t*>(d + VE / 2) = hi;
                }
            } else {
                const int kr = idx / (R / VE), rv = idx % (R / VE);
                *reinterpret_cast<vec_t*>(s + kr * (R + 16) + rv * VE) = v[q];
            }
        }
    }
    // fragment element for MFMA lane l: (row r0 + (l&15), k kk*4 + (l>>4))
    static __device__ __forceinline__ T frag(const T* s, int r, int k) {
        return KCONT ? s[r * (BKT + 2) + k] : s[k * (R + 16) + r];
    }
};

// Tile -> (ti, tj) for enumeration index wg of a launch with `nwg` tiles in this order (see the comments inside).
template <typename T, int BM, int BN>
__device__ __forceinline__ void gemm_decode(const GemmP<T>& p, int wg, int& ti, int& tj) {
    // Equal-work launches walk the tiles in bands of GR tile rows, column by column inside a band: the 64 tiles an XCD
    // has in flight then form an 8 x 8 patch that re-uses 8 A panels and 8 B panels from its L2 (a row-major walk
    // re-uses one A panel and misses on every B panel; PMC: -10 % L2 fetch over an evaluation).  Launches with K ranges
    // keep the row-major walk: tiles of one row start at the same k and stay in lockstep on their shared panel, tiles
    // of different rows do not, and patches made of them fetched MORE (measured: lauum 59 -> 80 GB, 8 % slower).
    constexpr int GR = 8;
    const bool grouped = !(p.klo | p.khi | p.noxcd);
    // tri with N < M: a trapezoid -- the lower triangle of the leading N x N block, then the full tile rows below it (the Cholesky's
    // trailing update restricted to a column range, and the deferred block's column panels: linalg.hip)
    int tmT = p.M / BM;                    // tile rows of the triangular part
    if (p.tri && p.N / BN < tmT) {
        const int tnA = p.N / BN, t0 = tnA * (tnA + 1) / 2;
        if (wg >= t0) {                    // bands of GR full rows, column by column inside a band
            const int w0 = wg - t0, band = w0 / (GR * tnA), r0 = band * GR;
            const int rows = min(GR, tmT - tnA - r0), w = w0 - band * GR * tnA;
            ti = tnA + r0 + w % rows;
            tj = w / rows;
            return;
        }
        tmT = tnA;
    }
    if (p.tri && grouped) {
        // band g = tile rows [8g, 8g+8): 64 g + 36 tiles, the bands before it hold 32 g (g - 1) + 36 g
        int g = (int)((sqrtf(1024.0f + 128.0f * (float)wg) - 32.0f) / 64.0f);
        while (g > 0 && 32L * g * (g - 1) + 36L * g > wg) --g;
        while (32L * (g + 1) * g + 36L * (g + 1) <= wg) ++g;
        int w = wg - (int)(32L * g * (g - 1) + 36L * g);
        const int r0 = GR * g;
        if (r0 + GR > tmT) {               // ragged last band: row-major inside it
            ti = r0;
            while (w >= ti + 1) { w -= ti + 1; ++ti; }
            tj = w;
        } else if (w < GR * r0) {          // full columns left of the diagonal patch
            tj = w / GR; ti = r0 + w % GR;
        } else {                           // the diagonal 8 x 8 triangle, column by column
            w -= GR * r0;
            int c = 0;
            while (w >= GR - c) { w -= GR - c; ++c; }
            tj = r0 + c; ti = r0 + c + w;
        }
    } else if (p.tri) {
        ti = (int)((sqrtf(8.0f * (float)wg + 1.0f) - 1.0f) * 0.5f);
        while ((long)ti * (ti + 1) / 2 > wg) --ti;
        while ((long)(ti + 1) * (ti + 2) / 2 <= wg) ++ti;
        tj = wg - ti * (ti + 1) / 2;
        // Workgroup b runs on XCD b mod 8.  Rotating the row's columns by its first index mod 8 gives that XCD the tile COLUMNS congruent
        // to it (up to the wrap at the row's end): of the column panels a row reads, each L2 holds an eighth instead of all of them
        // (round 4; with the reverse K walk of p.krev the tiles in flight are in step on those panels).
        if (p.krev & 1) tj = (tj + (int)(((long)ti * (ti + 1) / 2) & 7)) % (ti + 1);
    } else if (grouped) {
        const int tn = p.N / BN, tmr = p.M / BM;
        const int band = wg / (GR * tn), r0 = band * GR;
        const int rows = min(GR, tmr - r0), w = wg - band * GR * tn;
        ti = r0 + w % rows;
        tj = w / rows;
    } else if (p.khi == 2 && p.N / BN > 1) {
        // K grows with the tile COLUMN (a product against a lower-triangular B^T): column-major, the long columns first -- the tiles
        // of a column share their K range and B panel and stay in lockstep, and the launch does not end on its longest tiles
        const int tmr = p.M / BM;
        tj = p.N / BN - 1 - wg / tmr;
        ti = wg % tmr;
    } else {
        const int tn = p.N / BN;
        ti = wg / tn;
        tj = wg % tn;
        if (p.khi == 1) ti = p.M / BM - 1 - ti;   // K grows with the tile row: launch the long rows first
    }
}

// XCD-aware tile order: blocks b, b+8, .. share one XCD's L2; give each XCD a contiguous run of
// tiles (bijective for any grid size) so neighbouring tiles re-use operand panels from L2.
// Tiles with unequal K ranges (triangular operands) keep launch order instead: longest first and
// round-robin over the XCDs, which balances the work.
template <typename T> __device__ __forceinline__ int gemm_xcd_order(const GemmP<T>& p, int b, int nwg) {
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    return (p.klo | p.khi | p.noxcd) ? b : (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
}

template <typename T, bool TA, bool TB, int BM, int BN, int EPI, int BKT, int NW>
__device__ __forceinline__ void gemm_tile(const GemmP<T>& p, int ti, int tj, char* smem_raw) {
    constexpr int NTH = 64 * NW;
    typedef Stage<T, BM, !TA, BKT, NTH> SA;   // A: K-contiguous when not transposed
    typedef Stage<T, BN, TB, BKT, NTH> SB;    // B: K-contiguous when transposed
    typedef typename Mfma<T>::acc_t acc_t;
    // wave tile; 4 waves cover BM x BN: 128x128 -> 64x64, 64x256 -> 64x64, 64x64 -> 32x32, 64x128 -> 32x64,
    // 32x64 -> 16x32, 32x128 -> 32x32, 32x32 -> 16x16
    // (NW == 8: eight waves of 64 x 32 on the 128 x 128 block -- half the accumulators per wave, twice the waves per SIMD;
    //  eight waves of 16 x 32 on a 64 x 64 block: the quarter tiles that end a mixed launch, pg_gemm_mixed_kernel)
    constexpr int WTM = (NW == 8 && BM == 64) ? 16 : ((BM == 128 || BN == 256) ? 64 : ((BM == 32 && BN <= 64) ? 16 : 32));
    constexpr int WTN = (NW == 8) ? 32 : ((BN == 32) ? 16 : ((BN == 64 || (BM == 32 && BN == 128)) ? 32 : 64));
    constexpr int MIM = WTM / 16, MIN = WTN / 16;             // MFMA tiles per wave tile
    constexpr int WN_ = BN / WTN;                             // waves along N
    static_assert((BM / WTM) * (BN / WTN) == NW, "the waves must tile the block");
    const int ze = (int)blockIdx.y / p.batch;                      // which of the batched problems (experts)
    T* As = reinterpret_cast<T*>(smem_raw);
    T* Bs = As + 2 * SA::LDS_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN_, wn = wave % WN_;
    const long zb = (int)blockIdx.y % p.batch;
    const T* __restrict__ A = p.A + zb * p.sA + ze * p.eA;
    const T* __restrict__ B = p.B + zb * p.sB + ze * p.eB;
    T* __restrict__ C = p.C ? p.C + zb * p.sC + ze * p.eC : nullptr;

    const int m0 = ti * BM, n0 = tj * BN;
    int kbeg = 0, kend = p.K;
    int kbw = 0, kew = p.K;   // this wave's own useful K range (16-granular skip inside diagonal tiles)
    if (p.klo == 1) { kbeg = m0; kbw = m0 + wm * WTM; }
    if (p.klo == 2) { kbeg = n0; kbw = n0 + wn * WTN; }
    if (p.khi == 1) { kend = min(p.K, m0 + BM); kew = min(p.K, m0 + (wm + 1) * WTM); }
    if (p.khi == 2) { kend = min(p.K, n0 + BN); kew = min(p.K, n0 + (wn + 1) * WTN); }

    acc_t acc[MIM][MIN];
#pragma unroll
    for (int i = 0; i < MIM; ++i)
#pragma unroll
        for (int j = 0; j < MIN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (T)0;

    SA sa;
    SB sb;
    sa.init(p.lda, tid);
    sb.init(p.ldb, tid);
    const int nk = (kend - kbeg) / BKT;
    // K ranges that START with the tile row / column (klo: L^-T L^-1, the triangular inverse's first products) are walked from the END
    // downwards when the launch asks for it (p.krev): every tile then starts at the same k, and the tiles in flight together -- launched
    // longest first, so of similar length -- stay in step on the row slabs they share through L2 instead of running a fixed 128 (i' - i)
    // rows apart for their whole life.  Only the order of an element's k tiles changes (results agree to rounding).
    const bool rev = p.klo != 0 && p.krev != 0;
    const int kfirst = rev ? kend - BKT : kbeg, kstep = rev ? -BKT : BKT;
    if (nk > 0) {
        sa.load(SA::origin(A, p.lda, m0, kfirst));
        sb.load(SB::origin(B, p.ldb, n0, kfirst));
        sa.store(As, tid);
        sb.store(Bs, tid);
    }
    __syncthreads();

    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int k0 = kfirst + kt * kstep;
        if (kt + 1 < nk) {
            sa.load(SA::origin(A, p.lda, m0, k0 + kstep));
            sb.load(SB::origin(B, p.ldb, n0, k0 + kstep));
        }
        if (k0 + BKT > kbw && k0 < kew) {   // wave-uniform
            const T* as = As + cur * SA::LDS_ELEMS;
            const T* bs = Bs + cur * SB::LDS_ELEMS;
            // Fragments one k-step ahead (round 4) in the forms with an M/N-contiguous operand: its reads of consecutive k-steps lie 4
            // LDS rows apart and do not pair into one ds_read2, so the compiler left every k-step's eight MFMAs behind that step's own
            // three reads -- four exposed LDS latencies per K tile in TN / TT / NN against two in NT.  (NT keeps the plain loop: it is
            // at 128 VGPRs, and the second fragment set put a scratch reload into its K loop: 70.1 -> 68.9 TFLOP/s.)
            constexpr bool PF = TA || !TB;
            if constexpr (PF) {
                T a[2][MIM], bq[2][MIN];
#pragma unroll
                for (int i = 0; i < MIM; ++i) a[0][i] = SA::frag(as, wm * WTM + i * 16 + fr, fk);
#pragma unroll
                for (int j = 0; j < MIN; ++j) bq[0][j] = SB::frag(bs, wn * WTN + j * 16 + fr, fk);
#pragma unroll
                for (int kk = 0; kk < BKT / 4; ++kk) {
                    const int c = kk & 1, nx = c ^ 1;
                    if (kk + 1 < BKT / 4) {
#pragma unroll
                        for (int i = 0; i < MIM; ++i) a[nx][i] = SA::frag(as, wm * WTM + i * 16 + fr, (kk + 1) * 4 + fk);
#pragma unroll
                        for (int j = 0; j < MIN; ++j) bq[nx][j] = SB::frag(bs, wn * WTN + j * 16 + fr, (kk + 1) * 4 + fk);
                    }
#pragma unroll
                    for (int i = 0; i < MIM; ++i)
#pragma unroll
                        for (int j = 0; j < MIN; ++j) acc[i][j] = Mfma<T>::run(a[c][i], bq[c][j], acc[i][j]);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < BKT / 4; ++kk) {
                    T a[MIM], bq[MIN];
#pragma unroll
                    for (int i = 0; i < MIM; ++i) a[i] = SA::frag(as, wm * WTM + i * 16 + fr, kk * 4 + fk);
#pragma unroll
                    for (int j = 0; j < MIN; ++j) bq[j] = SB::frag(bs, wn * WTN + j * 16 + fr, kk * 4 + fk);
#pragma unroll
                    for (int i = 0; i < MIM; ++i)
#pragma unroll
                        for (int j = 0; j < MIN; ++j) acc[i][j] = Mfma<T>::run(a[i], bq[j], acc[i][j]);
                }
            }
        }
        if (kt + 1 < nk) {
            sa.store(As + (cur ^ 1) * SA::LDS_ELEMS, tid);
            sb.store(Bs + (cur ^ 1) * SB::LDS_ELEMS, tid);
        }
        __syncthreads();
    }

    if (EPI == 0) {
        const T alpha = p.alpha, beta = p.beta;
#pragma unroll
        for (int i = 0; i < MIM; ++i)
#pragma unroll
            for (int j = 0; j < MIN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * WTM + i * 16 + Mfma<T>::row(lane, r);
                    const int col = n0 + wn * WTN + j * 16 + fr;
                    T* c = C + (long)row * p.ldc + col;
                    T v = alpha * acc[i][j][r];
                    if (p.atomic_c) {
                        // beta == 1 (every update of the factorisation): C += alpha A B^T as a no-return fp64 add performed in
                        // L2.  One add per element, so the same result bit for bit -- but the tile is not read into the CU first:
                        // the epilogue's serial load -> add -> store chain (64 elements per lane, no registers left to batch the
                        // loads) was 10 % of a K = 1024 tile.  K = 1024 SYRK 63.8 -> 65.9 TFLOP/s, K = 512 53.5 -> 56.5,
                        // factorisation at n = 16384 29.2 -> 28.3 ms.
                        unsafeAtomicAdd(c, v);
                        continue;
                    }
                    if (beta != (T)0) v += beta * *c;
                    *c = v;
                }
    } else {
        // column sums of squares of this wave's 64 rows -> part[(m0/64 + wm)][col]
        static_assert(EPI == 0 || WTM == 64, "the column-sum epilogue assumes 64-row wave tiles");
#pragma unroll
        for (int j = 0; j < MIN; ++j) {
            T s = (T)0;
#pragma unroll
            for (int i = 0; i < MIM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += acc[i][j][r] * acc[i][j][r];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16) {
                const int col = n0 + wn * WTN + j * 16 + lane;
                p.part[(long)(m0 / 64 + wm) * p.ldp + col + zb * p.sC + (long)ze * p.eC] = s;   // (experts together: eC strides the partial sums)
            }
        }
    }
}

template <typename T, bool TA, bool TB, int BM, int BN, int EPI, int BKT = BK, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void pg_gemm_kernel(GemmP<T> p) {
    const int ze = (int)blockIdx.y / p.batch;                      // which of the batched problems (experts)
    if (p.info && p.info[(long)ze * p.einfo] != 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int ti, tj;
    gemm_decode<T, BM, BN>(p, gemm_xcd_order(p, blockIdx.x, gridDim.x), ti, tj);
    gemm_tile<T, TA, TB, BM, BN, EPI, BKT, NW>(p, ti, tj, smem_raw);
}

// Mixed launch (round 4): the first `tmain` tiles as 128 x 128 blocks, the REST as 64 x 64 quarter tiles of the same eight-wave
// workgroups.  A launch of equally long tiles on S workgroup slots takes ceil(tiles / S) rounds -- the lower tiles of an
// h = 8192, K = 8192 update are 2080 = 4.06 rounds and took five (8.9 ms against 7.75 for 2016 tiles, tools/probe_tri_rounds.py);
// with tmain = a whole number of rounds the remainder ends in quarter-length tiles instead.  One workgroup per output element as
// before: same numbers bit for bit (the k order of an element does not depend on the tile shape).
template <typename T>
__global__ __launch_bounds__(512, 4) void pg_gemm_mixed_kernel(GemmP<T> p, int tmain) {
    if (p.info && p.info[0] != 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int b = blockIdx.x;
    int ti, tj;
    if (b < tmain) {
        gemm_decode<T, 128, 128>(p, gemm_xcd_order(p, b, tmain), ti, tj);
        gemm_tile<T, false, true, 128, 128, 0, BK, 8>(p, ti, tj, smem_raw);
    } else {
        const int s = b - tmain, q = s & 3;
        gemm_decode<T, 128, 128>(p, tmain + (s >> 2), ti, tj);
        if (p.tri && ti == tj && q == 1) return;               // the quarter above the diagonal
        // a 32-deep K tile: one such workgroup per CU is bound by the latency of its one-tile-ahead prefetch, not by its MFMAs
        // (16-deep: 1.0 ms per quarter tile at K = 8192, the length of a FULL tile's 0.45 ms share twice over)
        gemm_tile<T, false, true, 64, 64, 0, 32, 8>(p, 2 * ti + (q >> 1), 2 * tj + (q & 1), smem_raw);
    }
}

template <typename T, bool TA, bool TB, int BM, int BN, int EPI, int BKT = BK, int NW = 4>
static int launch(hipStream_t st, const GemmP<T>& p) {
    typedef Stage<T, BM, !TA, BKT, 64 * NW> SA;
    typedef Stage<T, BN, TB, BKT, 64 * NW> SB;
    const size_t lds = 2 * (SA::LDS_ELEMS + SB::LDS_ELEMS) * sizeof(T);
    static bool attr_done = false;
    auto kern = pg_gemm_kernel<T, TA, TB, BM, BN, EPI, BKT, NW>;
    if (!attr_done) {
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    if (p.M % BM || p.N % BN || p.K % BKT || (p.tri && BM != BN)) {
        pg_set_error("pg_gemm: shape %d x %d x %d not tile aligned (%d x %d)", p.M, p.N, p.K, BM, BN);
        return -2;
    }
    const int tm = p.M / BM, tn = p.N / BN;
    if (p.tri && tn > tm) { pg_set_error("pg_gemm: tri with N = %d > M = %d", p.N, p.M); return -2; }
    const long tiles = p.tri ? (long)tn * (tn + 1) / 2 + (long)(tm - tn) * tn : (long)tm * tn;   // (tn < tm: a trapezoid)
    if (tiles == 0 || p.batch == 0 || p.nexp == 0) return 0;
    if (p.batch > 65535) { pg_set_error("pg_gemm: batch=%d exceeds the grid's y limit", p.batch); return -2; }
    // grid.y = batch * nexp is limited to 65535: many small experts go in chunks of whole experts (each launch addresses its experts
    // from 0, so the operand and status pointers move with the chunk)
    const int per = std::max(1, 65535 / p.batch);
    for (int e0 = 0; e0 < p.nexp; e0 += per) {
        GemmP<T> q = p;
        q.nexp = std::min(per, p.nexp - e0);
        q.A = p.A + (long)e0 * p.eA; q.B = p.B + (long)e0 * p.eB;
        if (p.C) q.C = p.C + (long)e0 * p.eC;
        if (p.part) q.part = p.part + (long)e0 * p.eC;
        if (p.info) q.info = p.info + (long)e0 * p.einfo;
        dim3 grid((unsigned)tiles, (unsigned)(q.batch * q.nexp), 1);
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, q);
        PG_CHECK(hipGetLastError());
    }
    return 0;
}

template <typename T> static int launch_mixed(hipStream_t st, const GemmP<T>& p, int tmain) {
    typedef Stage<T, 128, true, BK, 512> S128;
    typedef Stage<T, 64, true, 32, 512> S64;
    static_assert(S64::LDS_ELEMS <= S128::LDS_ELEMS, "the quarter tiles' stages must fit the block's LDS");
    const size_t lds = 4 * S128::LDS_ELEMS * sizeof(T);
    static bool attr_done = false;
    auto kern = pg_gemm_mixed_kernel<T>;
    if (!attr_done) {
        PG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const long tm = p.M / 128, tn = p.N / 128, tiles = p.tri ? tm * (tm + 1) / 2 : tm * tn;
    dim3 grid((unsigned)(tmain + 4 * (tiles - tmain)), 1, 1);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p, tmain);
    PG_CHECK(hipGetLastError());
    return 0;
}

double pg_gemm_flops(int variant, int M, int N, int K, int tri, int klo, int khi, int batch) {
    const int BM = (variant == GEMM_NT_32x64 || variant == GEMM_NT_32x128 || variant == GEMM_NT_32x32) ? 32
                   : ((variant == GEMM_NT_RP || variant == GEMM_NT_64 || variant == GEMM_NT_64x128 || variant == GEMM_TT_64 || variant == GEMM_TN_64) ? 64 : 128);
    const int BN = (variant == GEMM_NT_RP) ? 256
                   : ((variant == GEMM_NT_64 || variant == GEMM_NT_32x64 || variant == GEMM_TT_64 || variant == GEMM_TN_64) ? 64
                      : (variant == GEMM_NT_32x32 ? 32 : 128));
    const int tm = M / BM, tn = N / BN;
    double f = 0;
    for (int ti = 0; ti < tm; ++ti)
        for (int tj = 0; tj < (tri ? std::min(ti + 1, tn) : tn); ++tj) {
            int kb = 0, ke = K;
            if (klo == 1) kb = ti * BM;
            if (klo == 2) kb = tj * BN;
            if (khi == 1) ke = std::min(K, (ti + 1) * BM);
            if (khi == 2) ke = std::min(K, (tj + 1) * BN);
            if (ke > kb) f += 2.0 * BM * BN * (double)(ke - kb);
        }
    return f * batch;
}

template <typename T> int pg_gemm(pg_ctx* ctx, hipStream_t st, int variant_in, const GemmP<T>& p_in) {
    int variant = variant_in;
    static const int noxcd = getenv("PG_NOXCD") ? atoi(getenv("PG_NOXCD")) : 0;
    // 128 x 128 blocks: eight waves of 64 x 32 (default) or the round-1 form, four waves of 64 x 64 (PG_GEMM_W4=1).  Same
    // numbers bit for bit; four waves per SIMD instead of two cover the staging and barrier stalls: uniform 8192^3 69.1 ->
    // 71.8 TFLOP/s, K = 1024 SYRK 60.1 -> 64.0, L^-T L^-1 64.9 -> 67.5 (16 waves of 32 x 32 are slower: 59-67).
    // fp32 keeps four waves (its 64 x 64 wave tile needs half the registers: already four waves per SIMD; eight were 3 % slower).
    static const int w4env = getenv("PG_GEMM_W4") ? atoi(getenv("PG_GEMM_W4")) : 0;
    static const int ss4 = getenv("PG_SS_W4") ? atoi(getenv("PG_SS_W4")) : 1;   // the column-sum epilogue: four waves (8.63 vs 8.85 ms, config-2 predict)
    const bool w4 = w4env || sizeof(T) == 4 || ((variant == GEMM_NN_128_SS || variant == GEMM_NT_128_SS) && ss4);
    GemmP<T> p = p_in;
    p.noxcd = noxcd;
    if (p.nexp < 1) p.nexp = 1;
    static const int atomic_c = getenv("PG_ATOMIC_C") ? atoi(getenv("PG_ATOMIC_C")) : 1;   // 0: read-modify-write epilogue
    p.atomic_c = (atomic_c && !(ctx && ctx->no_atomic_c) && sizeof(T) == 8 && p.beta == (T)1 && p.C && p.part == nullptr) ? 1 : 0;
    const bool prof = ctx && ctx->prof_on;
    if (prof) PG_CHECK(hipEventRecord(ctx->ev[6], st));
    int rc;
    // equally long 128 x 128 tiles that do not fill a whole number of rounds of the stream's workgroup slots: mixed launch
    static const int mixed_env = getenv("PG_GEMM_MIXED") ? atoi(getenv("PG_GEMM_MIXED")) : 1;
    if (mixed_env && variant == GEMM_NT_128 && !w4 && sizeof(T) == 8 && ctx && ctx->ncu > 0 && !p.klo && !p.khi && !p.noxcd && p.batch == 1 &&
        p.nexp == 1 && p.part == nullptr && p.M % 128 == 0 && p.N % 128 == 0 && p.K % 32 == 0 && (!p.tri || p.M == p.N)) {
        const long tm = p.M / 128, tn = p.N / 128, tiles = p.tri ? tm * (tm + 1) / 2 : tm * tn;
        const long slots = 2L * ((ctx->upd && st == ctx->upd) ? ctx->upd_cus : ctx->ncu);
        const long rem = tiles % slots;
        // quarter tiles run at about 0.3 of a full tile's length each: worth it while four times the remainder stays within ~2.5 rounds
        if (tiles > slots && rem > 0 && 4 * rem <= (5 * slots) / 2) {
            if constexpr (sizeof(T) == 8) {
                if ((rc = launch_mixed<T>(st, p, (int)(tiles - rem)))) return rc;
                variant = -1;
            }
        }
    }
    switch (variant) {
        case -1: rc = 0; break;
        case GEMM_NT_128: rc = w4 ? launch<T, false, true, 128, 128, 0>(st, p) : launch<T, false, true, 128, 128, 0, BK, 8>(st, p); break;
        case GEMM_NT_RP: rc = launch<T, false, true, 64, 256, 0>(st, p); break;
        case GEMM_NN_128: rc = w4 ? launch<T, false, false, 128, 128, 0>(st, p) : launch<T, false, false, 128, 128, 0, BK, 8>(st, p); break;
        case GEMM_TN_128: rc = w4 ? launch<T, true, false, 128, 128, 0>(st, p) : launch<T, true, false, 128, 128, 0, BK, 8>(st, p); break;
        case GEMM_NT_128_SS: rc = w4 ? launch<T, false, true, 128, 128, 1>(st, p) : launch<T, false, true, 128, 128, 1, BK, 8>(st, p); break;
        case GEMM_NN_128_SS: rc = w4 ? launch<T, false, false, 128, 128, 1>(st, p) : launch<T, false, false, 128, 128, 1, BK, 8>(st, p); break;
        case GEMM_TT_128: rc = w4 ? launch<T, true, true, 128, 128, 0>(st, p) : launch<T, true, true, 128, 128, 0, BK, 8>(st, p); break;
        case GEMM_NT_64: rc = launch<T, false, true, 64, 64, 0>(st, p); break;
        case GEMM_NT_64x128: rc = launch<T, false, true, 64, 128, 0>(st, p); break;
        case GEMM_NT_32x64: rc = launch<T, false, true, 32, 64, 0, 32>(st, p); break;
        // fp64: a 16-deep K tile keeps the block's LDS at 46 KB (three per CU); with 32 it is 87 KB and one per CU, which
        // capped the chain's panel solve at a third of its CUs' rate
        case GEMM_NT_32x128: rc = launch<T, false, true, 32, 128, 0, (sizeof(T) == 8 ? 16 : 32)>(st, p); break;
        case GEMM_TT_64: rc = launch<T, true, true, 64, 64, 0>(st, p); break;
        case GEMM_TN_64: rc = launch<T, true, false, 64, 64, 0>(st, p); break;
        case GEMM_NT_32x32: rc = launch<T, false, true, 32, 32, 0, 64>(st, p); break;
        default: pg_set_error("pg_gemm: unknown variant %d", variant); return -2;
    }
    if (rc) return rc;
    variant = variant_in;
    {   // PG_GEMM_LOG=<file>: one line per launch (variant, waves, shape, flops) in launch order, for tools/trace_util.py
        static FILE* logf = getenv("PG_GEMM_LOG") ? fopen(getenv("PG_GEMM_LOG"), "w") : nullptr;
        if (logf) {
            fprintf(logf, "%d %d %d %d %d %d %d %.0f\n", variant, (int)sizeof(T), w4 ? 4 : 8, p.M, p.N, p.K, p.batch,
                    pg_gemm_flops(variant, p.M, p.N, p.K, p.tri, p.klo, p.khi, p.batch * p.nexp));
            fflush(logf);
        }
    }
    if (prof) {
        PG_CHECK(hipEventRecord(ctx->ev[7], st));
        PG_CHECK(hipEventSynchronize(ctx->ev[7]));
        float ms = 0;
        PG_CHECK(hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]));
        ctx->prof_ms += ms;
        ctx->prof_flops += pg_gemm_flops(variant, p.M, p.N, p.K, p.tri, p.klo, p.khi, p.batch * p.nexp);
        ctx->prof_launches += 1;
    }
    return 0;
}

template int pg_gemm<double>(pg_ctx*, hipStream_t, int, const GemmP<double>&);
template int pg_gemm<float>(pg_ctx*, hipStream_t, int, const GemmP<float>&);
