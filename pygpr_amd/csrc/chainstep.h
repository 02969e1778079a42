#pragma once
#include "common.h"
#define PG_CS_NCRIT 8   // workgroups that own block row k+1 (and so publish tile (k+1, k+1)) in a rows kernel
// Flag-coupled chain step (chainstep.hip): the rows below tile (k, k) of the outer panel that starts at column o0 -- the next block
// column's update by the panel's earlier columns, then (after *done_k) the solve against inv_k, then the last 128 columns of the
// update.  has_next = 0: the panel's last step (solve only).  n, k0, o0 multiples of 128.
// How a resident kernel of the chain waits: every spin is bounded by WALL time (wall_clock64, 100 MHz), not by an iteration count.
// On expiry the sticky device word *tmo ends every later spin of the call at once, *tmo_host (pinned host memory, read by the
// library at its next entry point) switches the handle to the classic chain, and the waiter reports *info = -1.
//   ticks > 0: budget per wait in 10 ns ticks (default 2 s: pg_create, PG_CS_SPIN_US, pg_set_spin_budget)
//   ticks < 0: every wait expires at once, whatever its flag holds -- the deterministic test hook of the fall-back path
struct CsWait {
    int* tmo;
    int* tmo_host;
    long long ticks;
    int epoch;        // which factorisation of the handle this is (> 0): what an expiry writes to *tmo_host, so that the host counts
                      // one time-out per call however many of its waits expire and whenever it polls
};
#define PG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void cs_expire(const CsWait& w) {
    __hip_atomic_store(w.tmo, 1, PG_RLX_AGENT);
    if (w.tmo_host) __hip_atomic_store(w.tmo_host, w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool cs_spin_ge(int* flag, int want, const CsWait& w) {   // ONE lane
    if (w.ticks < 0) {
        cs_expire(w);
        return false;
    }
    long long t0 = 0;
    for (unsigned it = 0;; ++it) {
        if (__hip_atomic_load(flag, PG_RLX_AGENT) >= want) return true;
        if ((it & 31u) == 31u) {
            if (__hip_atomic_load(w.tmo, PG_RLX_AGENT) != 0) return false;
            const long long now = (long long)wall_clock64();
            if (it == 31u) t0 = now;
            else if (now - t0 > w.ticks) {
                cs_expire(w);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// Experts together on the coupled chain (round 4): the kernels take a second grid dimension, one expert each, with its own matrix,
// diagonal-block inverses, flag words (they live in each expert's work buffer) and status word.
struct CsBatch {
    int nexp;
    long eA, eInv;    // strides of the matrices and of the inverse blocks (elements)
    long eF;          // stride of the flag arrays (ints)
};
template <typename T>
int pg_rowstep(hipStream_t st, T* A, long lda, int n, int o0, int k0, int has_next, const T* inv, int* done_k, int* brow_k,
               int* diag_next, const CsWait& wt, int* info, int* early_k = nullptr, int* browe_k = nullptr, int allow_tlog = 1,
               const CsBatch* cb = nullptr);   // early_k: two-phase hand-over
int pg_flagset(hipStream_t st, int* flag, int value, int nexp = 1, long eF = 0);
int pg_spin_probe_launch(hipStream_t st, int* flag, const CsWait& w, long long* out);   // tests: one bounded wait on a flag nobody sets
