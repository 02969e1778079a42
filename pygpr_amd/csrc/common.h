// Shared device/host helpers for libpygpr_hip (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "pygpr_hip.h"

#define PG_TILE 128      // GEMM block tile and Cholesky leaf size
#define PG_RESERVED_CUS 32   // default; PG_RESERVED_CUS in the environment overrides
#define PG_BG_CUS 4096         // CUs the background stream may use, clamped to all non-reserved ones (measured best; PG_BG_CUS overrides)
#define PG_PAD 256       // every matrix dimension handed to the O(n^3) kernels is a multiple of this


typedef double pg_d4 __attribute__((ext_vector_type(4)));
typedef float pg_f4 __attribute__((ext_vector_type(4)));

struct pg_ctx {
    hipStream_t aux;          // look-ahead / panel stream (highest priority, non-blocking)
    hipStream_t rows;         // rows stream of the flag-coupled chain (chainstep.hip; highest priority, non-blocking)
    hipStream_t bg;           // background stream (same CU mask as upd): L^-1 of the leading half during potrf's tail
    hipStream_t upd;          // trailing-update stream: CU mask leaves PG_RESERVED_CUS compute units to the panel chain
                              // (the 128x128 leaf needs a whole CU's LDS and starves beside a chip-filling SYRK)
    hipEvent_t ev[8];
    hipEvent_t* pool;         // events for cross-stream dependencies, grown on demand
    int npool;
    int lookahead;            // 0 disables the two-stream Cholesky (default 1)
    int nbo;                  // outer panel of the Cholesky; 0 = chosen from n (pg_set_outer_panel / PG_NBO)
    int rec_min;              // the fused factor-and-invert call splits recursively from this n on (pg_set_recursive_split / PG_REC_MIN; 0: never)
    int coupled;              // the flag-coupled chain may be used (pg_set_coupled_chain; cleared by pg_create when kernels of the
                              // panel and rows streams do not run concurrently here, e.g. under a counter-collecting profiler)
    int last_coupled;         // panels the last factorisation ran on the flag-coupled chain (chainstep.hip); tests / diagnostics
    int defer;                // deferred trailing block in the coupled factorisation (pg_set_deferred_block / PG_DEFER; default 0)
    int last_deferred;        // column panels of the last factorisation whose updates by the first half came as deferred deep-K products (linalg.hip)
    int panel_mode;           // how the rows below an outer panel ride its 128-column steps (linalg.hip, PG_PANEL_MODE)
    int side_pending;         // side-stream work (pg_alpha_nlml_async) that the next reader of its outputs must wait for: ev[5]
    hipStream_t side_owner;   // the caller stream that work was forked from: only a join on THAT stream clears side_pending
    int* tmo_host;            // pinned host word a timed-out wait of the coupled chain sets (chainstep.h); polled at every entry point
    int* tmo_dev;             // its device address
    long long spin_ticks;     // budget of one wait of the coupled chain in 10 ns ticks (< 0: forced expiry, test hook; 0: scaled to the call, pg_potrf_t)
    int timeouts;             // coupled-chain time-outs seen on this handle (each switched the handle to the classic chain)
    int tmo_off;              // the chain is off BECAUSE of a time-out (not by the caller's choice or a failed probe): re-armed automatically
    int calls_since_tmo;      // factorisations enqueued on the classic chain since then
    int rearm_after;          // ... after this many of them the handle probes again and re-arms (pg_set_rearm_after / PG_CS_REARM; 0: never)
    int rearms;               // automatic re-arms so far
    int rearms_seen;          // re-arms already answered by a back-off
    int rearm_cur;            // the current distance: doubled (up to 4096) by every time-out that follows a re-arm, so that a GPU shared for
                              // good with another process' resident kernels costs one wait budget ever more rarely, not every rearm_after calls
    int calls_since_rearm;    // factorisations since the last automatic re-arm: a re-armed chain that survives rearm_cur of them has
                              // earned the short distance back (the back-off decays)
    int* probe_words;         // pinned words of the queue probe, allocated once per handle (pg_create's probe and every re-arm use them)
    int chain_epoch;          // counts the factorisations that took the coupled chain; an expiry reports its call's number
    int counted_epoch;        // the last epoch whose time-out was counted
    int no_atomic_c;          // set for the duration of an entry point whose C operand is not plain device memory
    int ncu, upd_cus;         // compute units of the device / those the update stream's mask leaves it (workgroup slots of a launch: 2 per CU)
    int prof_on;              // profiling of the GEMM core (bench roofline leg)
    double prof_flops;
    double prof_ms;
    long prof_launches;
};

void pg_set_error(const char* fmt, ...);

#define PG_CHECK(expr)                                                              \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            pg_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,               \
                         hipGetErrorString(_e));                                    \
            return -100;                                                            \
        }                                                                           \
    } while (0)

// MFMA wrapper: one 16x16x4 step, D = A(16x4) B(4x16) + C.
// lane l supplies A[l&15][l>>4] and B[l>>4][l&15] for both dtypes
// (cdna_hip_programming.md section 3; f64 C/D map differs from f32).
template <typename T> struct Mfma;
template <> struct Mfma<double> {
    typedef pg_d4 acc_t;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // row of accumulator register r inside the 16x16 tile
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mfma<float> {
    typedef pg_f4 acc_t;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
