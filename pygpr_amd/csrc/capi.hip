// extern "C" surface of libpygpr_hip (declared in include/pygpr_hip.h): argument checks, dtype dispatch.
#include "gemm.h"
#include "kbuild.h"
#include "leaf.h"
#include "chainstep.h"
#include "linalg.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

// Live handles.  A handle owns priority / CU-masked streams and events; if they are still alive when the HIP runtime
// tears itself down at process exit, the runtime's finalisers touch freed queue state (round 1: SIGSEGV in
// __cxa_finalize under rocprofv3, after the profiler had written its output).  pg_create therefore registers a C atexit
// handler the first time it runs -- i.e. after the HIP runtime registered its own, so it runs BEFORE the runtime's --
// that destroys whatever the caller did not pg_destroy.
static std::mutex g_live_mu;
static std::vector<pg_ctx*> g_live;
static bool g_atexit_registered = false;

static void release_ctx(pg_ctx* h) {
    (void)hipStreamDestroy(h->aux);
    if (h->rows) (void)hipStreamDestroy(h->rows);
    if (h->upd) (void)hipStreamDestroy(h->upd);
    if (h->bg) (void)hipStreamDestroy(h->bg);
    for (int i = 0; i < 8; ++i) (void)hipEventDestroy(h->ev[i]);
    for (int i = 0; i < h->npool; ++i) (void)hipEventDestroy(h->pool[i]);
    free(h->pool);
    if (h->tmo_host) (void)hipHostFree(h->tmo_host);
    if (h->probe_words) (void)hipHostFree(h->probe_words);
    delete h;
}

static void destroy_live_handles() {
    std::vector<pg_ctx*> left;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        left.swap(g_live);
    }
    if (left.empty()) return;
    (void)hipDeviceSynchronize();
    for (pg_ctx* h : left) release_ctx(h);
}

void pg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define NEED(cond, what)                                         \
    do {                                                         \
        if (!(cond)) { pg_set_error("%s: %s", __func__, what); return -1; } \
    } while (0)
#define DISPATCH(dtype, CALL_D, CALL_F)                                      \
    do {                                                                     \
        if ((dtype) == PG_F64) return CALL_D;                                \
        if ((dtype) == PG_F32) return CALL_F;                                \
        pg_set_error("%s: unknown dtype %d", __func__, (int)(dtype));        \
        return -1;                                                           \
    } while (0)

// A timed-out wait of the coupled chain (chainstep.h) sets the handle's pinned host word: from the next entry point on the handle
// factorises on the classic chain (no flags, no resident kernels), so that repeating the failed call -- which is what
// pg_build_potrf_trtri_checked and the Python layer do -- ends in a correct factor.  pg_set_coupled_chain(h, 1) probes and re-arms.
static void poll_timeout(pg_ctx* h) {
    if (h && h->tmo_host) {
        const int v = *(volatile int*)h->tmo_host;     // epoch of the factorisation whose wait expired (chainstep.h)
        if (v) {
            *(volatile int*)h->tmo_host = 0;
            if (h->coupled) {
                h->tmo_off = 1; h->calls_since_tmo = 0;
                if (h->rearms > h->rearms_seen) { h->rearm_cur = std::min(4096, std::max(1, h->rearm_cur) * 2); h->rearms_seen = h->rearms; }   // timed out again after a re-arm: back off
            }
            h->coupled = 0;
            if (v != h->counted_epoch) {               // one count per call, however many waits expired and whenever we looked
                h->counted_epoch = v;
                h->timeouts += 1;
            }
        }
    }
}

// pg_alpha_nlml_async leaves work on the handle's side stream; every entry point that may read its outputs (everything that
// takes a stream except pg_lauum, which is meant to overlap with it) first makes its stream wait for that work.  Only a join on
// the stream the work was forked from retires the pending mark: a call on another stream in between waits too, but the owner's
// next reader still does (round 2 cleared the mark at the first join of ANY stream).
static int join_side(pg_ctx* h, void* stream) {
    poll_timeout(h);
    if (h && h->side_pending) {
        PG_CHECK(hipStreamWaitEvent(ST(stream), h->ev[5], 0));
        if (ST(stream) == h->side_owner) h->side_pending = 0;
    }
    return 0;
}
#define JOIN(h, stream)                         \
    do {                                        \
        int _j = join_side((h), (stream));      \
        if (_j) return _j;                      \
    } while (0)

// The beta == 1 epilogue of the GEMM core adds into C with no-return fp64 atomics performed at the memory side.  On fine-grained,
// host-pinned or managed allocations such atomics may be dropped or crawl, so the entry points that take a caller-owned C check
// the pointer once and keep the read-modify-write epilogue for anything that is not plain device memory.
static int plain_device_memory(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    // (fine-grained device allocations -- hipExtMallocWithFlags(hipDeviceMallocFinegrained) -- also report hipMemoryTypeDevice: they are
    //  exactly the memory where memory-side fp64 atomics may be dropped)
    return a.type == hipMemoryTypeDevice && !a.isManaged && !(a.allocationFlags & hipDeviceMallocFinegrained);
}
struct AtomicGuard {
    pg_ctx* h;
    AtomicGuard(pg_ctx* h_, const void* c) : h(h_) { if (h) h->no_atomic_c = plain_device_memory(c) ? 0 : 1; }
    ~AtomicGuard() { if (h) h->no_atomic_c = 0; }
};

template <typename T>
static int gemm_raw_t(pg_handle h, int variant, int M, int N, int K, double alpha, const void* A, long lda, const void* B,
                      long ldb, double beta, void* C, long ldc, int tri, int klo, int khi, void* stream) {
    GemmP<T> p;
    p.A = (const T*)A; p.B = (const T*)B; p.C = (T*)C;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.alpha = (T)alpha; p.beta = (T)beta;
    p.tri = tri; p.klo = klo & 3; p.khi = khi; p.krev = (klo >> 2) & 1;   // (klo bit 2: reverse walk, measurement only)
    p.sA = p.sB = p.sC = 0; p.batch = 1;
    p.nexp = 1; p.eA = p.eB = p.eC = 0; p.einfo = 0;
    p.part = nullptr; p.ldp = 0; p.info = nullptr; p.noxcd = 0;
    // PG_RAW_STREAM=upd|bg (measurement only): run the product on one of the handle's CU-masked streams instead
    static const char* rs = getenv("PG_RAW_STREAM");
    hipStream_t on = ST(stream);
    if (rs && rs[0] == 'u' && h->upd) on = h->upd;
    if (rs && rs[0] == 'b' && h->bg) on = h->bg;
    if (on == ST(stream)) return pg_gemm<T>(h, on, variant, p);
    PG_CHECK(hipEventRecord(h->ev[2], ST(stream)));      // (ev[4] / ev[5] belong to pg_alpha_nlml_async's side-stream fork)
    PG_CHECK(hipStreamWaitEvent(on, h->ev[2], 0));
    const int rc = pg_gemm<T>(h, on, variant, p);
    PG_CHECK(hipEventRecord(h->ev[3], on));
    PG_CHECK(hipStreamWaitEvent(ST(stream), h->ev[3], 0));
    return rc;
}

// Do kernels of the panel stream and the rows stream run at the same time here?  The coupled chain needs that: its leaf is
// resident while the rows kernel that publishes its tile is still queued on the other stream.  A counter-collecting profiler
// (rocprofv3 --pmc) or a serialising debug setting runs one kernel at a time; the chain's bounded waits would then expire and the
// factorisation report info = -1.  Probe once per handle: a waiter on the panel stream, enqueued FIRST, then the setter on the rows
// stream.  Concurrent queues: the waiter sees the flag within microseconds.  One kernel at a time: it gives up after ~2 ms.
__global__ void pg_probe_wait_kernel(int* flag, int* seen) {
    int ok = 0;
    for (int it = 0; it < 4096 && !ok; ++it) {
        ok = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        __builtin_amdgcn_s_sleep(32);
    }
    __hip_atomic_store(seen, ok ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void pg_probe_set_kernel(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

static int probe_concurrent_queues(pg_ctx* c) {
    // (the words live as long as the handle: a re-arm inside an enqueueing entry point then costs the probe's two stream
    //  synchronisations, not a pinned allocation and its device-wide release as well)
    if (!c->probe_words && hipHostMalloc(reinterpret_cast<void**>(&c->probe_words), 64, hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        c->probe_words = nullptr;
        return 0;
    }
    int* words = c->probe_words;
    words[0] = 0;
    words[1] = 0;
    int* dwords = nullptr;
    int ok = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dwords), words, 0) == hipSuccess) {
        hipLaunchKernelGGL(pg_probe_wait_kernel, dim3(1), dim3(1), 0, c->aux, dwords, dwords + 1);
        hipLaunchKernelGGL(pg_probe_set_kernel, dim3(1), dim3(1), 0, c->rows, dwords);
        if (hipStreamSynchronize(c->aux) == hipSuccess && hipStreamSynchronize(c->rows) == hipSuccess) ok = words[1] == 1;
    }
    (void)hipGetLastError();
    return ok;
}

// A handle whose chain was switched off by a time-out counts its factorisations; after rearm_after of them it probes the queues
// again (2 ms at most, once) and takes the coupled chain back: the cause on record (profiles/r03_bench_n2_gloo_rehearsal.json: another
// process's resident kernels on the same GPU) is transient, the downgrade should be too.
static void maybe_rearm(pg_ctx* h) {
    if (h && h->coupled && h->rearms > 0 && h->rearm_cur > h->rearm_after && ++h->calls_since_rearm > h->rearm_cur) {
        h->rearm_cur = h->rearm_after;      // the re-armed chain ran cleanly for a whole back-off distance: forget the back-off
        h->calls_since_rearm = 0;
    }
    if (!h || !h->tmo_off || h->coupled || h->rearm_after <= 0) return;
    if (h->rearm_cur < h->rearm_after) h->rearm_cur = h->rearm_after;
    if (++h->calls_since_tmo <= h->rearm_cur) return;
    h->calls_since_tmo = 0;
    if (h->rows && h->spin_ticks >= 0 && probe_concurrent_queues(h)) {
        h->coupled = 1;
        h->tmo_off = 0;
        h->rearms += 1;
        h->calls_since_rearm = 0;
    }
}

template <typename T>
static int build_potrf_t(pg_handle h, hipStream_t st, const pg_covspec* spec, const double* hp, const void* X, long ldx, int n, int d,
                         double jitter, void* A, long lda, int n_pad, void* inv_diag, int* info, void* Minv, long ldm) {
    BuildReq<T> br;
    br.spec = spec; br.hp = hp; br.X = (const T*)X; br.ldx = ldx; br.n_real = n; br.d = d; br.jitter = jitter;
    return pg_potrf_t<T>(h, st, n_pad, (T*)A, lda, (T*)inv_diag, info, (T*)Minv, ldm, &br);
}

template <typename T>
static int potrf_batched_t(pg_handle h, hipStream_t st, const pg_covspec* spec, const double* hp, long hp_stride, const void* X, long ldx,
                           long x_stride, int n, int d, double jitter, void* A, long lda, long a_stride, int n_pad, void* inv_diag,
                           long inv_stride, int* info, void* Minv, long ldm, long m_stride, int nexp) {
    ExpBatch eb;
    eb.nexp = nexp; eb.eA = a_stride; eb.eInv = inv_stride; eb.eM = m_stride; eb.eX = x_stride; eb.ehp = hp_stride;
    BuildReq<T> br;
    br.spec = spec; br.hp = hp; br.X = (const T*)X; br.ldx = ldx; br.n_real = n; br.d = d; br.jitter = jitter;
    return pg_potrf_t<T>(h, st, n_pad, (T*)A, lda, (T*)inv_diag, info, (T*)Minv, ldm, X ? &br : nullptr, &eb);
}

extern "C" {

int pg_version(void) { return 100; }
const char* pg_last_error(void) { return g_err; }

int pg_create(pg_handle* h) {
    NEED(h, "null handle pointer");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        pg_set_error("pg_create: no HIP device visible");
        return -101;
    }
    pg_ctx* c = new pg_ctx();
    memset(c, 0, sizeof(*c));
    int prio_lo = 0, prio_hi = 0;
    PG_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    PG_CHECK(hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, prio_hi));
    c->rows = nullptr;
    c->last_coupled = 0;
    c->defer = getenv("PG_DEFER") ? atoi(getenv("PG_DEFER")) : 0;
    c->lookahead = 1;
    {
        const char* e = getenv("PG_NBO");
        c->nbo = e ? atoi(e) : 0;
        if (c->nbo < 128 || c->nbo % 128 || c->nbo > 2048) c->nbo = 0;
        const char* r = getenv("PG_REC_MIN");
        c->rec_min = r ? atoi(r) : 16384;
        if (c->rec_min < 0 || (c->rec_min > 0 && (c->rec_min < 512 || c->rec_min % 512))) c->rec_min = 16384;
        const char* v = getenv("PG_PANEL_MODE");
        c->panel_mode = v ? atoi(v) : 0;
    }
    {   // update stream on all but the last PG_RESERVED_CUS compute units; without it look-ahead stays off
        hipDeviceProp_t prop;
        int dev = 0;
        PG_CHECK(hipGetDevice(&dev));
        PG_CHECK(hipGetDeviceProperties(&prop, dev));
        const int ncu = prop.multiProcessorCount;
        const int words = (ncu + 31) / 32;
        uint32_t mask[64];
        for (int i = 0; i < 64; ++i) mask[i] = 0;
        const char* env = getenv("PG_RESERVED_CUS");
        int reserved = env ? atoi(env) : PG_RESERVED_CUS;
        if (reserved < 1 || reserved > ncu / 2) reserved = PG_RESERVED_CUS;
        c->ncu = ncu; c->upd_cus = ncu - reserved;
        for (int cu = 0; cu < ncu - reserved; ++cu) mask[cu / 32] |= (1u << (cu % 32));   // (a strided reservation was measured: 9 % slower)
        if (ncu <= reserved * 2 || words > 64 ||
            hipExtStreamCreateWithCUMask(&c->upd, (uint32_t)words, mask) != hipSuccess) {
            c->upd = nullptr;
            c->lookahead = 0;
            (void)hipGetLastError();
        }
        // The background stream of the fused factor-and-invert call (PG_BG_STREAM=1, replaces the rows stream) is an experiment:
        // its overlap returned 0.5 ms of 8.7 at n = 8192 and nothing at 16384, and a handle with TWO CU-masked streams is
        // fragile -- any fifth hardware queue in the process (a second stream of the caller's is enough) then costs the
        // look-ahead 20-60 %, which the current panel / rows / update set does not show with up to seven queues.
        c->bg = nullptr;
        const char* envbg = getenv("PG_BG_STREAM");
        if (envbg && atoi(envbg)) {
            const char* envb = getenv("PG_BG_CUS");
            int bgcus = envb ? atoi(envb) : PG_BG_CUS;
            if (bgcus < 8 || bgcus > ncu - reserved) bgcus = ncu - reserved;
            for (int i = 0; i < 64; ++i) mask[i] = 0;
            for (int cu = 0; cu < bgcus; ++cu) mask[cu / 32] |= (1u << (cu % 32));
            if (c->upd && hipExtStreamCreateWithCUMask(&c->bg, (uint32_t)words, mask) != hipSuccess) {
                c->bg = nullptr;
                (void)hipGetLastError();
            }
        }
    }
    {   // rows stream of the flag-coupled chain (chainstep.hip): non-blocking, same priority as the panel stream
        const char* envr = getenv("PG_ROWS_STREAM");
        c->rows = nullptr;
        if (!(envr && !atoi(envr)) && !c->bg && c->upd &&
            hipStreamCreateWithPriority(&c->rows, hipStreamNonBlocking, prio_hi) != hipSuccess) {
            c->rows = nullptr;
            (void)hipGetLastError();
        }
    }
    c->coupled = (c->rows && probe_concurrent_queues(c)) ? 1 : 0;
    {   // pinned word through which a timed-out wait of the coupled chain reaches the host (poll_timeout), and the wait budget
        c->tmo_host = nullptr;
        c->tmo_dev = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&c->tmo_host), 64, hipHostMallocMapped) == hipSuccess) {
            c->tmo_host[0] = 0;
            if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->tmo_dev), c->tmo_host, 0) != hipSuccess) c->tmo_dev = nullptr;
        }
        (void)hipGetLastError();
        // budget of one wait: by default scaled to the call (pg_potrf_t: 20x the classic-chain estimate of the coupled region, at least
        // 50 ms -- round 3's flat 2 s was 70x a whole N = 16384 factorisation, and in a lock-step all-reduce one rank's stall is every
        // rank's); PG_CS_SPIN_US / pg_set_spin_budget fix it
        const char* e = getenv("PG_CS_SPIN_US");
        const long long us = e ? atoll(e) : 0;
        c->spin_ticks = us < 0 ? -1 : us * 100;
        c->timeouts = 0;
        const char* ra = getenv("PG_CS_REARM");
        c->rearm_after = ra ? std::max(0, atoi(ra)) : 8;
        c->rearm_cur = c->rearm_after;
    }
    for (int i = 0; i < 8; ++i) PG_CHECK(hipEventCreate(&c->ev[i]));
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.push_back(c);
        if (!g_atexit_registered) {
            atexit(destroy_live_handles);
            g_atexit_registered = true;
        }
    }
    *h = c;
    return 0;
}

int pg_destroy(pg_handle h) {
    if (!h) return 0;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        auto it = std::find(g_live.begin(), g_live.end(), h);
        if (it == g_live.end()) return 0;      // already released (second pg_destroy, or after the exit handler ran)
        g_live.erase(it);
    }
    release_ctx(h);
    return 0;
}

static int check_spec(const pg_covspec* s, const char* fn, bool allow_sqdist = false) {
    if (!s || s->ncomp < 0 || s->ncomp > PG_MAX_COMP || s->nnoise < 0 || s->nnoise > PG_MAX_COMP) {
        pg_set_error("%s: bad covariance spec", fn);
        return -1;
    }
    for (int c = 0; c < s->ncomp; ++c)
        if (s->kind[c] != PG_KIND_RBF && s->kind[c] != PG_KIND_MATERN52 && !(allow_sqdist && s->kind[c] == PG_KIND_SQDIST)) {
            pg_set_error("%s: unknown kernel kind %d", fn, s->kind[c]);
            return -1;
        }
    return 0;
}

int pg_kernel_build(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* Xr, long ldr, int nr,
                    const void* Xc, long ldc, int nc, int d, int lower_only, int accumulate, double jitter, void* K,
                    long ldk, int rows_pad, int cols_pad, void* stream) {
    JOIN(h, stream);
    NEED(h && hp && Xr && K, "null pointer");
    if (check_spec(spec, __func__, true)) return -1;
    const int sym = (Xc == nullptr);
    if (sym) { Xc = Xr; ldc = ldr; nc = nr; }
    NEED(nr >= 0 && nc >= 0 && rows_pad >= nr && cols_pad >= nc && ldk >= cols_pad, "inconsistent sizes");
    NEED(!sym || rows_pad == cols_pad, "symmetric build needs a square padded shape");
    NEED(ldk % (dtype == PG_F64 ? 2 : 4) == 0, "ldk must keep rows 16-byte aligned");
    DISPATCH(dtype,
             pg_kbuild<double>(ST(stream), *spec, hp, (const double*)Xr, ldr, nr, (const double*)Xc, ldc, nc, d, sym,
                               lower_only, accumulate, jitter, (double*)K, ldk, rows_pad, cols_pad),
             pg_kbuild<float>(ST(stream), *spec, hp, (const float*)Xr, ldr, nr, (const float*)Xc, ldc, nc, d, sym,
                              lower_only, accumulate, jitter, (float*)K, ldk, rows_pad, cols_pad));
}

int pg_kernel_build_batched(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, long hp_stride, const void* Xr, long ldr,
                            long xr_stride, int nr, const void* Xc, long ldc, long xc_stride, int nc, int d, int lower_only, double jitter,
                            void* K, long ldk, long k_stride, int rows_pad, int cols_pad, int nexp, void* stream) {
    JOIN(h, stream);
    NEED(h && hp && Xr && K, "null pointer");
    if (check_spec(spec, __func__, true)) return -1;
    NEED(nexp >= 1 && nexp <= 65535, "1 <= nexp <= 65535");
    const int sym = (Xc == nullptr);
    if (sym) { Xc = Xr; ldc = ldr; nc = nr; xc_stride = xr_stride; }
    NEED(nr >= 0 && nc >= 0 && rows_pad >= nr && cols_pad >= nc && ldk >= cols_pad, "inconsistent sizes");
    NEED(!sym || rows_pad == cols_pad, "symmetric build needs a square padded shape");
    NEED(nexp == 1 || k_stride >= (long)rows_pad * ldk, "experts' matrices overlap");
    NEED(ldk % (dtype == PG_F64 ? 2 : 4) == 0, "ldk must keep rows 16-byte aligned");
    DISPATCH(dtype,
             pg_kbuild<double>(ST(stream), *spec, hp, (const double*)Xr, ldr, nr, (const double*)Xc, ldc, nc, d, sym, lower_only, 0, jitter,
                               (double*)K, ldk, rows_pad, cols_pad, 0, 0, nexp, xc_stride, hp_stride, k_stride, xr_stride),
             pg_kbuild<float>(ST(stream), *spec, hp, (const float*)Xr, ldr, nr, (const float*)Xc, ldc, nc, d, sym, lower_only, 0, jitter,
                              (float*)K, ldk, rows_pad, cols_pad, 0, 0, nexp, xc_stride, hp_stride, k_stride, xr_stride));
}

int pg_kernel_grad_build(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* X, long ldx,
                         int n, int d, void* dK, void* stream) {
    JOIN(h, stream);
    NEED(h && hp && X && dK, "null pointer");
    if (check_spec(spec, __func__)) return -1;
    DISPATCH(dtype, pg_kgrad<double>(ST(stream), *spec, hp, (const double*)X, ldx, n, d, (double*)dK),
             pg_kgrad<float>(ST(stream), *spec, hp, (const float*)X, ldx, n, d, (float*)dK));
}

int pg_build_potrf_trtri(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* X, long ldx, int n, int d,
                         double jitter, void* A, long lda, int n_pad, void* inv_diag, int* info, void* Minv, long ldm,
                         void* stream) {
    JOIN(h, stream);
    maybe_rearm(h);
    NEED(h && hp && X && A && inv_diag && info, "null pointer");
    if (check_spec(spec, __func__, true)) return -1;
    NEED(n >= 0 && n_pad >= n && lda >= n_pad, "inconsistent sizes");
    NEED(lda % (dtype == PG_F64 ? 2 : 4) == 0, "lda must keep rows 16-byte aligned");
    AtomicGuard ag(h, A);
    DISPATCH(dtype, build_potrf_t<double>(h, ST(stream), spec, hp, X, ldx, n, d, jitter, A, lda, n_pad, inv_diag, info, Minv, ldm),
             build_potrf_t<float>(h, ST(stream), spec, hp, X, ldx, n, d, jitter, A, lda, n_pad, inv_diag, info, Minv, ldm));
}

int pg_build_potrf_trtri_batched(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, long hp_stride, const void* X, long ldx,
                                 long x_stride, int n, int d, double jitter, void* A, long lda, long a_stride, int n_pad, void* inv_diag,
                                 long inv_stride, int* info, void* Minv, long ldm, long m_stride, int nexp, void* stream) {
    JOIN(h, stream);
    NEED(h && A && inv_diag && info, "null pointer");
    NEED(!X || hp, "a folded build needs hp");
    if (X && check_spec(spec, __func__, true)) return -1;
    NEED(nexp >= 1 && nexp <= 65535, "1 <= nexp <= 65535");
    NEED(n >= 0 && n_pad >= n && lda >= n_pad, "inconsistent sizes");
    NEED(a_stride >= (long)n_pad * lda || nexp == 1, "experts' matrices overlap");
    NEED(inv_stride >= pg_potrf_worksize_impl(n_pad) || nexp == 1, "experts' workspaces overlap (stride < pg_potrf_worksize)");
    NEED(!Minv || (ldm >= n_pad && (m_stride >= (long)n_pad * ldm || nexp == 1)), "experts' inverses overlap");
    NEED(lda % (dtype == PG_F64 ? 2 : 4) == 0, "lda must keep rows 16-byte aligned");
    maybe_rearm(h);      // a batched-only workload counts towards the re-arm like the single-matrix calls (one per call)
    AtomicGuard ag(h, A);
    DISPATCH(dtype,
             potrf_batched_t<double>(h, ST(stream), spec, hp, hp_stride, X, ldx, x_stride, n, d, jitter, A, lda, a_stride, n_pad, inv_diag,
                                     inv_stride, info, Minv, ldm, m_stride, nexp),
             potrf_batched_t<float>(h, ST(stream), spec, hp, hp_stride, X, ldx, x_stride, n, d, jitter, A, lda, a_stride, n_pad, inv_diag,
                                    inv_stride, info, Minv, ldm, m_stride, nexp));
}

long pg_potrf_worksize(int dtype, int n) { (void)dtype; return pg_potrf_worksize_impl(n); }

int pg_potrf(pg_handle h, int dtype, int n, void* A, long lda, void* inv_diag, int* info, void* stream) {
    JOIN(h, stream);
    maybe_rearm(h);
    NEED(h && A && inv_diag && info, "null pointer");
    AtomicGuard ag(h, A);
    NEED(lda >= n, "lda < n");
    DISPATCH(dtype, pg_potrf_t<double>(h, ST(stream), n, (double*)A, lda, (double*)inv_diag, info, nullptr, 0),
             pg_potrf_t<float>(h, ST(stream), n, (float*)A, lda, (float*)inv_diag, info, nullptr, 0));
}

int pg_potrf_trtri(pg_handle h, int dtype, int n, void* A, long lda, void* inv_diag, int* info, void* Minv, long ldm,
                   void* stream) {
    JOIN(h, stream);
    maybe_rearm(h);
    NEED(h && A && inv_diag && info && Minv, "null pointer");
    NEED(lda >= n && ldm >= n && A != Minv, "bad leading dimension / aliasing");
    AtomicGuard ag(h, A);
    DISPATCH(dtype, pg_potrf_t<double>(h, ST(stream), n, (double*)A, lda, (double*)inv_diag, info, (double*)Minv, ldm),
             pg_potrf_t<float>(h, ST(stream), n, (float*)A, lda, (float*)inv_diag, info, (float*)Minv, ldm));
}

long pg_potrs_vec_worksize(int dtype, int n) { (void)dtype; return pg_potrs_vec_worksize_impl(n); }

int pg_potrs_vec(pg_handle h, int dtype, int n, const void* L, long ldl, const void* inv_diag, const void* y, void* x,
                 void* work, void* stream) {
    JOIN(h, stream);
    NEED(h && L && inv_diag && y && x && work, "null pointer");
    DISPATCH(dtype,
             pg_potrs_vec_t<double>(h, ST(stream), n, (const double*)L, ldl, (const double*)inv_diag, (const double*)y,
                                    (double*)x, (double*)work),
             pg_potrs_vec_t<float>(h, ST(stream), n, (const float*)L, ldl, (const float*)inv_diag, (const float*)y,
                                   (float*)x, (float*)work));
}

int pg_trtri(pg_handle h, int dtype, int n, const void* L, long ldl, const void* inv_diag, void* Minv, long ldm,
             void* stream) {
    JOIN(h, stream);
    NEED(h && L && inv_diag && Minv, "null pointer");
    NEED(L != Minv, "pg_trtri is out of place");
    DISPATCH(dtype,
             pg_trtri_t<double>(h, ST(stream), n, (const double*)L, ldl, (const double*)inv_diag, (double*)Minv, ldm),
             pg_trtri_t<float>(h, ST(stream), n, (const float*)L, ldl, (const float*)inv_diag, (float*)Minv, ldm));
}

long pg_potrs_worksize(int dtype, int n, int nrhs, int have_minv) {
    (void)dtype;
    return (have_minv ? 0L : (long)n * n) + (long)n * nrhs;
}

static int potrs_any(pg_handle h, int dtype, int n, int nrhs, const void* L, long ldl, const void* inv_diag, const void* Minv, long ldm,
                     const void* B, long ldb, void* X, long ldx, void* work, int both, void* stream) {
    JOIN(h, stream);
    NEED(h && B && X && work && (Minv || (L && inv_diag)), "null pointer");
    NEED(ldb >= nrhs && ldx >= nrhs && (Minv ? ldm >= n : ldl >= n), "leading dimension too small");
    NEED(B != X, "out of place: X must not alias B");
    DISPATCH(dtype,
             pg_potrs_t<double>(h, ST(stream), n, nrhs, (const double*)L, ldl, (const double*)inv_diag, (const double*)Minv, ldm,
                                (const double*)B, ldb, (double*)X, ldx, (double*)work, both),
             pg_potrs_t<float>(h, ST(stream), n, nrhs, (const float*)L, ldl, (const float*)inv_diag, (const float*)Minv, ldm, (const float*)B,
                               ldb, (float*)X, ldx, (float*)work, both));
}
int pg_potrs(pg_handle h, int dtype, int n, int nrhs, const void* L, long ldl, const void* inv_diag, const void* Minv, long ldm, const void* B,
             long ldb, void* X, long ldx, void* work, void* stream) {
    return potrs_any(h, dtype, n, nrhs, L, ldl, inv_diag, Minv, ldm, B, ldb, X, ldx, work, 1, stream);
}
int pg_trsm_lower(pg_handle h, int dtype, int n, int nrhs, const void* L, long ldl, const void* inv_diag, const void* Minv, long ldm,
                  const void* B, long ldb, void* X, long ldx, void* work, void* stream) {
    return potrs_any(h, dtype, n, nrhs, L, ldl, inv_diag, Minv, ldm, B, ldb, X, ldx, work, 0, stream);
}

int pg_lauum(pg_handle h, int dtype, int n, const void* Minv, long ldm, void* Kinv, long ldk, void* stream) {
    NEED(h && Minv && Kinv, "null pointer");
    NEED(Minv != Kinv, "pg_lauum is out of place");
    DISPATCH(dtype, pg_lauum_t<double>(h, ST(stream), n, (const double*)Minv, ldm, (double*)Kinv, ldk),
             pg_lauum_t<float>(h, ST(stream), n, (const float*)Minv, ldm, (float*)Kinv, ldk));
}

int pg_potri(pg_handle h, int dtype, int n, const void* L, long ldl, const void* inv_diag, void* Kinv, long ldk, void* work,
             void* stream) {
    JOIN(h, stream);
    NEED(h && L && inv_diag && Kinv && work, "null pointer");
    NEED(work != L && work != Kinv, "pg_potri: work must not alias L or Kinv");
    NEED(ldl >= n && ldk >= n, "leading dimension < n");
    int rc = pg_trtri(h, dtype, n, L, ldl, inv_diag, work, (long)n, stream);
    if (rc) return rc;
    return pg_lauum(h, dtype, n, work, (long)n, Kinv, ldk, stream);
}

int pg_logdet(pg_handle h, int dtype, int n, const void* L, long ldl, double* out, void* stream) {
    JOIN(h, stream);
    NEED(h && L && out, "null pointer");
    NEED(n > 0 && ldl >= n, "bad size");
    DISPATCH(dtype, pg_logdet_t<double>(ST(stream), n, (const double*)L, ldl, out),
             pg_logdet_t<float>(ST(stream), n, (const float*)L, ldl, out));
}

int pg_trmv(pg_handle h, int dtype, int n, const void* Minv, long ldm, int trans, const void* x, void* y, void* work,
            void* stream) {
    JOIN(h, stream);
    NEED(h && Minv && x && y, "null pointer");
    NEED(!trans || work, "transposed product needs a workspace");
    NEED(x != y, "pg_trmv is out of place");
    DISPATCH(dtype,
             pg_trmv_t<double>(h, ST(stream), n, (const double*)Minv, ldm, trans, (const double*)x, (double*)y, (double*)work),
             pg_trmv_t<float>(h, ST(stream), n, (const float*)Minv, ldm, trans, (const float*)x, (float*)y, (float*)work));
}

int pg_alpha_batched(pg_handle h, int dtype, int n, const void* Minv, long ldm, long m_stride, const void* y, long y_stride, void* u,
                     long u_stride, void* alpha, long alpha_stride, void* work, long work_stride, int nexp, void* stream) {
    JOIN(h, stream);
    NEED(h && Minv && y && u && alpha && work, "null pointer");
    NEED(ldm >= n && nexp >= 1 && nexp <= 65535, "bad size");
    DISPATCH(dtype,
             pg_alpha_batched_t<double>(ST(stream), n, (const double*)Minv, ldm, m_stride, (const double*)y, y_stride, (double*)u, u_stride,
                                        (double*)alpha, alpha_stride, (double*)work, work_stride, nexp),
             pg_alpha_batched_t<float>(ST(stream), n, (const float*)Minv, ldm, m_stride, (const float*)y, y_stride, (float*)u, u_stride,
                                       (float*)alpha, alpha_stride, (float*)work, work_stride, nexp));
}

int pg_alpha_nlml_batched(pg_handle h, int dtype, int n_real, int n, const void* Minv, long ldm, long m_stride, const void* y, long y_stride,
                          void* u, long u_stride, void* alpha, long alpha_stride, void* work, long work_stride, double* out, long out_stride,
                          int nexp, void* stream) {
    JOIN(h, stream);
    NEED(h && Minv && y && u && alpha && work && out, "null pointer");
    NEED(n_real > 0 && n_real <= n && ldm >= n && nexp >= 1 && nexp <= 65535, "bad size");
    DISPATCH(dtype,
             pg_alpha_batched_t<double>(ST(stream), n, (const double*)Minv, ldm, m_stride, (const double*)y, y_stride, (double*)u, u_stride,
                                        (double*)alpha, alpha_stride, (double*)work, work_stride, nexp, n_real, out, out_stride),
             pg_alpha_batched_t<float>(ST(stream), n, (const float*)Minv, ldm, m_stride, (const float*)y, y_stride, (float*)u, u_stride,
                                       (float*)alpha, alpha_stride, (float*)work, work_stride, nexp, n_real, out, out_stride));
}

int pg_lauum_batched(pg_handle h, int dtype, int n, const void* Minv, long ldm, long m_stride, void* Kinv, long ldk, long k_stride, int nexp,
                     void* stream) {
    JOIN(h, stream);
    NEED(h && Minv && Kinv, "null pointer");
    NEED(Minv != Kinv, "pg_lauum_batched is out of place");
    NEED(nexp >= 1 && nexp <= 65535 && ldm >= n && ldk >= n, "bad size");
    NEED(nexp == 1 || (m_stride >= (long)n * ldm && k_stride >= (long)n * ldk), "experts' matrices overlap");
    ExpBatch eb;
    eb.nexp = nexp; eb.eA = k_stride; eb.eInv = 0; eb.eM = m_stride; eb.eX = 0; eb.ehp = 0;
    DISPATCH(dtype, pg_lauum_t<double>(h, ST(stream), n, (const double*)Minv, ldm, (double*)Kinv, ldk, &eb),
             pg_lauum_t<float>(h, ST(stream), n, (const float*)Minv, ldm, (float*)Kinv, ldk, &eb));
}

int pg_nlml_grad_batched(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, long hp_stride, const void* X, long ldx, long x_stride,
                         int n, int d, const void* Kinv, long ldk, long k_stride, const void* alpha, long alpha_stride, double* grad,
                         long grad_stride, int nhp, double* work, long lwork, int nexp, void* stream) {
    JOIN(h, stream);
    NEED(h && hp && X && Kinv && alpha && grad && work, "null pointer");
    if (check_spec(spec, __func__)) return -1;
    DISPATCH(dtype,
             pg_nlml_grad_t<double>(ST(stream), *spec, hp, (const double*)X, ldx, n, d, (const double*)Kinv, ldk, (const double*)alpha, grad,
                                    nhp, work, lwork, nexp, hp_stride, x_stride, k_stride, alpha_stride, grad_stride),
             pg_nlml_grad_t<float>(ST(stream), *spec, hp, (const float*)X, ldx, n, d, (const float*)Kinv, ldk, (const float*)alpha, grad, nhp,
                                   work, lwork, nexp, hp_stride, x_stride, k_stride, alpha_stride, grad_stride));
}

int pg_alpha_nlml_async(pg_handle h, int dtype, int n_real, int n, const void* L, long ldl, const void* Minv, long ldm, const void* y,
                        void* u, void* alpha, void* work, double* out, void* stream) {
    JOIN(h, stream);
    NEED(h && L && Minv && y && u && alpha && work && out, "null pointer");
    NEED(n_real > 0 && n_real <= n && ldl >= n && ldm >= n, "bad size");
    DISPATCH(dtype,
             pg_alpha_nlml_async_t<double>(h, ST(stream), n_real, n, (const double*)L, ldl, (const double*)Minv, ldm, (const double*)y,
                                           (double*)u, (double*)alpha, (double*)work, out),
             pg_alpha_nlml_async_t<float>(h, ST(stream), n_real, n, (const float*)L, ldl, (const float*)Minv, ldm, (const float*)y,
                                          (float*)u, (float*)alpha, (float*)work, out));
}

int pg_nlml_value(pg_handle h, int dtype, int n, const void* L, long ldl, const void* y, const void* alpha, double* out,
                  void* stream) {
    JOIN(h, stream);
    NEED(h && L && y && alpha && out, "null pointer");
    DISPATCH(dtype,
             pg_nlml_value_t<double>(ST(stream), n, (const double*)L, ldl, (const double*)y, (const double*)alpha, out),
             pg_nlml_value_t<float>(ST(stream), n, (const float*)L, ldl, (const float*)y, (const float*)alpha, out));
}

long pg_nlml_grad_worksize(int n, int nhp) { return pg_nlml_grad_worksize_impl(n, nhp); }

int pg_nlml_grad(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* X, long ldx, int n, int d,
                 const void* Kinv, long ldk, const void* alpha, double* grad, int nhp, double* work, long lwork,
                 void* stream) {
    JOIN(h, stream);
    NEED(h && hp && X && Kinv && alpha && grad && work, "null pointer");
    if (check_spec(spec, __func__)) return -1;
    DISPATCH(dtype,
             pg_nlml_grad_t<double>(ST(stream), *spec, hp, (const double*)X, ldx, n, d, (const double*)Kinv, ldk,
                                    (const double*)alpha, grad, nhp, work, lwork),
             pg_nlml_grad_t<float>(ST(stream), *spec, hp, (const float*)X, ldx, n, d, (const float*)Kinv, ldk,
                                   (const float*)alpha, grad, nhp, work, lwork));
}

int pg_predict_mean_q(pg_handle h, int dtype, int n_pad, int m_pad, const void* Ks, long ldks, const void* Minv, long ldm,
                      const void* alpha, void* mean, void* q, double kss, void* work, void* stream) {
    JOIN(h, stream);
    NEED(h && Ks && alpha && mean && work, "null pointer");
    NEED(!q || Minv, "variance needs Minv");
    DISPATCH(dtype,
             pg_predict_mean_q_t<double>(h, ST(stream), n_pad, m_pad, (const double*)Ks, ldks, (const double*)Minv, ldm,
                                         (const double*)alpha, (double*)mean, (double*)q, kss, (double*)work),
             pg_predict_mean_q_t<float>(h, ST(stream), n_pad, m_pad, (const float*)Ks, ldks, (const float*)Minv, ldm,
                                        (const float*)alpha, (float*)mean, (float*)q, kss, (float*)work));
}

int pg_predict_mean_q_kt(pg_handle h, int dtype, int n_pad, int m_pad, const void* Kt, long ldkt, const void* Minv, long ldm,
                         const void* alpha, void* mean, void* q, double kss, void* work, void* stream) {
    JOIN(h, stream);
    NEED(h && Kt && alpha && mean && work, "null pointer");
    NEED(!q || Minv, "variance needs Minv");
    DISPATCH(dtype,
             pg_predict_mean_q_kt_t<double>(h, ST(stream), n_pad, m_pad, (const double*)Kt, ldkt, (const double*)Minv, ldm,
                                            (const double*)alpha, (double*)mean, (double*)q, kss, (double*)work),
             pg_predict_mean_q_kt_t<float>(h, ST(stream), n_pad, m_pad, (const float*)Kt, ldkt, (const float*)Minv, ldm,
                                           (const float*)alpha, (float*)mean, (float*)q, kss, (float*)work));
}

int pg_predict_mean_q_kt_batched(pg_handle h, int dtype, int n_pad, int m_pad, const void* Kt, long ldkt, long kt_stride, const void* Minv,
                                 long ldm, long m_stride, const void* alpha, long alpha_stride, void* mean, long mean_stride, void* var,
                                 long var_stride, const pg_covspec* spec, const double* hp, long hp_stride, void* work, long work_stride,
                                 int nexp, void* stream) {
    JOIN(h, stream);
    NEED(h && Kt && alpha && mean && work, "null pointer");
    NEED(!var || (Minv && hp), "variance needs Minv and hp");
    if (var && check_spec(spec, __func__)) return -1;
    NEED(nexp >= 1 && nexp <= 65535, "1 <= nexp <= 65535");
    NEED(ldkt >= n_pad && (!Minv || ldm >= n_pad), "bad leading dimension");
    NEED(nexp == 1 || (mean_stride >= m_pad && (!var || (var_stride >= m_pad && work_stride >= (long)(n_pad / 64) * m_pad))),
         "experts' outputs / workspaces overlap");
    static const pg_covspec none = {};
    DISPATCH(dtype,
             pg_predict_mean_q_kt_batched_t<double>(h, ST(stream), n_pad, m_pad, (const double*)Kt, ldkt, kt_stride, (const double*)Minv, ldm,
                                                    m_stride, (const double*)alpha, alpha_stride, (double*)mean, mean_stride, (double*)var,
                                                    var_stride, spec ? *spec : none, hp, hp_stride, (double*)work, work_stride, nexp),
             pg_predict_mean_q_kt_batched_t<float>(h, ST(stream), n_pad, m_pad, (const float*)Kt, ldkt, kt_stride, (const float*)Minv, ldm,
                                                   m_stride, (const float*)alpha, alpha_stride, (float*)mean, mean_stride, (float*)var,
                                                   var_stride, spec ? *spec : none, hp, hp_stride, (float*)work, work_stride, nexp));
}

int pg_trmm_lower(pg_handle h, int dtype, int n_pad, int m_pad, const void* Minv, long ldm, const void* Ks, long ldks,
                  void* V, long ldv, void* stream) {
    JOIN(h, stream);
    NEED(h && Minv && Ks && V, "null pointer");
    DISPATCH(dtype,
             pg_trmm_lower_t<double>(h, ST(stream), n_pad, m_pad, (const double*)Minv, ldm, (const double*)Ks, ldks,
                                     (double*)V, ldv),
             pg_trmm_lower_t<float>(h, ST(stream), n_pad, m_pad, (const float*)Minv, ldm, (const float*)Ks, ldks,
                                    (float*)V, ldv));
}

int pg_syrk_tn_sub(pg_handle h, int dtype, int m_pad, int n_pad, const void* V, long ldv, void* C, long ldc, int lower_only,
                   void* stream) {
    JOIN(h, stream);
    NEED(h && V && C, "null pointer");
    AtomicGuard ag(h, C);
    DISPATCH(dtype, pg_syrk_tn_sub_t<double>(h, ST(stream), m_pad, n_pad, (const double*)V, ldv, (double*)C, ldc, lower_only),
             pg_syrk_tn_sub_t<float>(h, ST(stream), m_pad, n_pad, (const float*)V, ldv, (float*)C, ldc, lower_only));
}

int pg_trmm_lower_kt_batched(pg_handle h, int dtype, int n_pad, int m_pad, const void* Minv, long ldm, long m_stride, const void* Kt, long ldkt,
                             long kt_stride, void* Vt, long ldvt, long vt_stride, int nexp, void* stream) {
    JOIN(h, stream);
    NEED(h && Minv && Kt && Vt, "null pointer");
    NEED(nexp >= 1 && nexp <= 65535, "1 <= nexp <= 65535");
    NEED(ldm >= n_pad && ldkt >= n_pad && ldvt >= n_pad && Kt != Vt, "bad leading dimension / aliasing");
    NEED(nexp == 1 || vt_stride >= (long)m_pad * ldvt, "experts' outputs overlap");
    DISPATCH(dtype,
             pg_trmm_lower_kt_t<double>(h, ST(stream), n_pad, m_pad, (const double*)Minv, ldm, m_stride, (const double*)Kt, ldkt, kt_stride,
                                        (double*)Vt, ldvt, vt_stride, nexp),
             pg_trmm_lower_kt_t<float>(h, ST(stream), n_pad, m_pad, (const float*)Minv, ldm, m_stride, (const float*)Kt, ldkt, kt_stride,
                                       (float*)Vt, ldvt, vt_stride, nexp));
}

int pg_syrk_nt_sub_batched(pg_handle h, int dtype, int m_pad, int n_pad, const void* Vt, long ldvt, long vt_stride, void* C, long ldc,
                           long c_stride, int nexp, int lower_only, void* stream) {
    JOIN(h, stream);
    NEED(h && Vt && C, "null pointer");
    NEED(nexp >= 1 && nexp <= 65535, "1 <= nexp <= 65535");
    NEED(ldvt >= n_pad && ldc >= m_pad, "bad leading dimension");
    NEED(nexp == 1 || (vt_stride >= (long)m_pad * ldvt && c_stride >= (long)m_pad * ldc), "experts' matrices overlap");
    AtomicGuard ag(h, C);
    DISPATCH(dtype,
             pg_syrk_nt_sub_t<double>(h, ST(stream), m_pad, n_pad, (const double*)Vt, ldvt, vt_stride, (double*)C, ldc, c_stride, nexp, lower_only),
             pg_syrk_nt_sub_t<float>(h, ST(stream), m_pad, n_pad, (const float*)Vt, ldvt, vt_stride, (float*)C, ldc, c_stride, nexp, lower_only));
}

int pg_grbcm_local_terms(pg_handle h, int dtype, int m, const void* mean_c, const void* var_c, const void* var_g,
                         int is_first, int accumulate, double* out, long ldo, double* beta_out, double* prec_out,
                         void* stream) {
    JOIN(h, stream);
    NEED(h && mean_c && var_c && var_g && out, "null pointer");
    NEED(ldo >= m, "ldo < m");
    DISPATCH(dtype,
             pg_grbcm_terms_t<double>(ST(stream), m, (const double*)mean_c, (const double*)var_c, (const double*)var_g,
                                      is_first, accumulate, out, ldo, beta_out, prec_out),
             pg_grbcm_terms_t<float>(ST(stream), m, (const float*)mean_c, (const float*)var_c, (const float*)var_g,
                                     is_first, accumulate, out, ldo, beta_out, prec_out));
}

int pg_grbcm_local_terms_batched(pg_handle h, int dtype, int m, const void* mean_l, long mean_stride, const void* var_l, long var_stride,
                                 const void* var_g, int nexp, int first, int accumulate, double* out, long ldo, double* beta_out,
                                 double* prec_out, long ldb, void* stream) {
    JOIN(h, stream);
    NEED(h && mean_l && var_l && var_g && out, "null pointer");
    NEED(ldo >= m && nexp >= 1 && (nexp == 1 || (!beta_out && !prec_out) || ldb >= m), "bad size");
    DISPATCH(dtype,
             pg_grbcm_terms_batched_t<double>(ST(stream), m, (const double*)mean_l, mean_stride, (const double*)var_l, var_stride,
                                              (const double*)var_g, nexp, first, accumulate, out, ldo, beta_out, prec_out, ldb),
             pg_grbcm_terms_batched_t<float>(ST(stream), m, (const float*)mean_l, mean_stride, (const float*)var_l, var_stride,
                                             (const float*)var_g, nexp, first, accumulate, out, ldo, beta_out, prec_out, ldb));
}

int pg_grbcm_finish(pg_handle h, int dtype, int m, const double* sums, long lds, const void* mean_g, const void* var_g,
                    void* mean, void* var, double* beta0, double* prec0, void* stream) {
    JOIN(h, stream);
    NEED(h && sums && mean_g && var_g && mean && var, "null pointer");
    DISPATCH(dtype,
             pg_grbcm_finish_t<double>(ST(stream), m, sums, lds, (const double*)mean_g, (const double*)var_g, (double*)mean,
                                       (double*)var, beta0, prec0),
             pg_grbcm_finish_t<float>(ST(stream), m, sums, lds, (const float*)mean_g, (const float*)var_g, (float*)mean,
                                      (float*)var, beta0, prec0));
}

int pg_grbcm_weighted_prec(pg_handle h, int dtype, int m, int m_pad, const void* P, long ldp, const double* beta, void* acc,
                           long lda, int accumulate, void* stream) {
    JOIN(h, stream);
    NEED(h && P && beta && acc, "null pointer");
    NEED(m_pad >= m && ldp >= m && lda >= m_pad, "inconsistent sizes");
    DISPATCH(dtype, pg_weighted_prec_t<double>(ST(stream), m, m_pad, (const double*)P, ldp, beta, (double*)acc, lda, accumulate),
             pg_weighted_prec_t<float>(ST(stream), m, m_pad, (const float*)P, ldp, beta, (float*)acc, lda, accumulate));
}

int pg_symmetrize(pg_handle h, int dtype, int n, void* A, long lda, void* stream) {
    JOIN(h, stream);
    NEED(h && A, "null pointer");
    DISPATCH(dtype, pg_symmetrize_t<double>(ST(stream), n, (double*)A, lda), pg_symmetrize_t<float>(ST(stream), n, (float*)A, lda));
}

int pg_grbcm_finish_full(pg_handle h, int dtype, int m, const double* sums, long lds, const void* mean_g, const void* var_g,
                         const void* cov, long ldc, void* mean, void* stream) {
    JOIN(h, stream);
    NEED(h && sums && mean_g && var_g && cov && mean, "null pointer");
    DISPATCH(dtype,
             pg_grbcm_finish_full_t<double>(ST(stream), m, sums, lds, (const double*)mean_g, (const double*)var_g,
                                            (const double*)cov, ldc, (double*)mean),
             pg_grbcm_finish_full_t<float>(ST(stream), m, sums, lds, (const float*)mean_g, (const float*)var_g, (const float*)cov,
                                           ldc, (float*)mean));
}

int pg_sqdist_argmin(pg_handle h, int dtype, const void* X, long ldx, int n, const void* C, long ldc, int m, int d, void* D,
                     long ldd, int* idx, void* stream) {
    JOIN(h, stream);
    NEED(h && X && C && (D || idx), "null pointer");
    DISPATCH(dtype,
             pg_centres<double>(ST(stream), (const double*)X, ldx, n, (const double*)C, ldc, m, d, (double*)D, ldd, idx),
             pg_centres<float>(ST(stream), (const float*)X, ldx, n, (const float*)C, ldc, m, d, (float*)D, ldd, idx));
}

int pg_tril(pg_handle h, int dtype, int n, void* A, long lda, void* stream) {
    JOIN(h, stream);
    NEED(h && A, "null pointer");
    DISPATCH(dtype, pg_tril_t<double>(ST(stream), n, (double*)A, lda), pg_tril_t<float>(ST(stream), n, (float*)A, lda));
}

int pg_set_lookahead(pg_handle h, int on) {
    NEED(h, "null handle");
    h->lookahead = (on && h->upd) ? 1 : 0;
    return 0;
}

int pg_set_outer_panel(pg_handle h, int columns) {
    NEED(h, "null handle");
    NEED(columns == 0 || (columns >= 128 && columns % 128 == 0 && columns <= 2048), "outer panel must be 0 (automatic) or a multiple of 128 up to 2048");
    h->nbo = columns;
    return 0;
}

int pg_set_recursive_split(pg_handle h, int min_n) {
    NEED(h, "null handle");
    NEED(min_n == 0 || (min_n >= 512 && min_n % 512 == 0), "min_n must be 0 (never) or a multiple of 512");
    h->rec_min = min_n;
    return 0;
}

int pg_profile(pg_handle h, int on) {
    NEED(h, "null handle");
    if (on) { h->prof_flops = 0; h->prof_ms = 0; h->prof_launches = 0; }
    h->prof_on = on;
    return 0;
}
int pg_set_coupled_chain(pg_handle h, int on) {
    NEED(h, "null handle");
    if (on < 0) {       // off for now, as after a time-out: the rows stream stays and the handle re-arms itself (pg_set_rearm_after)
        if (h->coupled || h->tmo_off == 0) { h->tmo_off = h->rows ? 1 : 0; h->calls_since_tmo = 0; }
        h->coupled = 0;
        return 0;
    }
    h->tmo_off = 0;
    h->rearm_cur = h->rearm_after;
    h->rearms_seen = h->rearms;      // an explicit switch forgets the back-off's history: the next time-out starts from rearm_after again
    h->calls_since_rearm = 0;
    if (!on) {
        // off also RELEASES the rows stream (the handle then owns two streams)
        if (h->rows) {
            PG_CHECK(hipStreamSynchronize(h->rows));
            PG_CHECK(hipStreamDestroy(h->rows));
            h->rows = nullptr;
        }
        h->coupled = 0;
        return 0;
    }
    if (!h->rows && h->upd && !h->bg) {
        int prio_lo = 0, prio_hi = 0;
        PG_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        if (hipStreamCreateWithPriority(&h->rows, hipStreamNonBlocking, prio_hi) != hipSuccess) {
            h->rows = nullptr;
            (void)hipGetLastError();
        }
    }
    h->coupled = (h->rows && probe_concurrent_queues(h)) ? 1 : 0;
    return 0;
}
int pg_coupled_chain(pg_handle h) { return h ? h->coupled : -1; }
int pg_last_coupled_panels(pg_handle h) {
    if (!h) return -1;
    return h->last_coupled;
}
int pg_set_deferred_block(pg_handle h, int on) {
    NEED(h, "null handle");
    h->defer = on ? 1 : 0;
    return 0;
}
int pg_last_deferred_panels(pg_handle h) {
    if (!h) return -1;
    return h->last_deferred;
}
int pg_profile_read(pg_handle h, double* flops, double* ms, long* launches) {
    NEED(h, "null handle");
    if (flops) *flops = h->prof_flops;
    if (ms) *ms = h->prof_ms;
    if (launches) *launches = h->prof_launches;
    return 0;
}

int pg_set_spin_budget(pg_handle h, long microseconds) {
    NEED(h, "null handle");
    h->spin_ticks = microseconds < 0 ? -1 : (long long)microseconds * 100;      // 0: scaled to the call
    return 0;
}
int pg_set_rearm_after(pg_handle h, int calls) {
    NEED(h, "null handle");
    NEED(calls >= 0, "calls must be >= 0 (0: never re-arm automatically)");
    h->rearm_after = calls;
    h->rearm_cur = calls;
    h->rearms_seen = h->rearms;
    h->calls_since_rearm = 0;
    return 0;
}
int pg_chain_rearms(pg_handle h) { return h ? h->rearms : -1; }
long pg_wait_budget_us(pg_handle h, int n) {
    if (!h) return 0;
    const long long t = pg_wait_ticks(h, n);
    return t < 0 ? -1 : (long)(t / 100);
}
// scratch: 4 ints + 2 long longs of device memory (32 bytes, 8-byte aligned).  The handle's state is not touched (no pinned word, no
// switch to the classic chain): the probe only measures the wait.
int pg_spin_probe(pg_handle h, int n, void* scratch, void* stream) {
    NEED(h && scratch, "null pointer");
    PG_CHECK(hipMemsetAsync(scratch, 0, 32, ST(stream)));
    int* w = reinterpret_cast<int*>(scratch);
    const CsWait cw = {w + 1, nullptr, pg_wait_ticks(h, n), 1};
    return pg_spin_probe_launch(ST(stream), w, cw, reinterpret_cast<long long*>(w + 4));
}
int pg_chain_timeouts(pg_handle h) {
    if (!h) return -1;
    poll_timeout(h);
    return h->timeouts;
}

// Blocking form of pg_build_potrf_trtri: waits for the factorisation and, when a wait of the coupled chain expired (info = -1),
// repeats the whole call -- covariance build included, the first attempt's A is garbage -- on the classic chain before it returns.
int pg_build_potrf_trtri_checked(pg_handle h, int dtype, const pg_covspec* spec, const double* hp, const void* X, long ldx, int n,
                                 int d, double jitter, void* A, long lda, int n_pad, void* inv_diag, int* info, void* Minv, long ldm,
                                 void* stream, int* info_host) {
    NEED(info_host, "null pointer");
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int rc = pg_build_potrf_trtri(h, dtype, spec, hp, X, ldx, n, d, jitter, A, lda, n_pad, inv_diag, info, Minv, ldm, stream);
        if (rc) return rc;
        PG_CHECK(hipMemcpyAsync(info_host, info, sizeof(int), hipMemcpyDeviceToHost, ST(stream)));
        PG_CHECK(hipStreamSynchronize(ST(stream)));
        if (*info_host >= 0) return 0;
        poll_timeout(h);
        if (h->coupled) { h->tmo_off = 1; h->calls_since_tmo = 0; }
        h->coupled = 0;        // whatever the pinned word said: the repeat must not take the coupled chain
    }
    return 0;
}

int pg_leaf_raw(pg_handle h, int dtype, void* A, long lda, void* inv, long ldi, int* info, int ablate, void* stream) {
    JOIN(h, stream);
    NEED(h && A && info, "null pointer");
    DISPATCH(dtype, pg_leaf<double>(ST(stream), (double*)A, lda, (double*)inv, ldi, info, 0, ablate),
             pg_leaf<float>(ST(stream), (float*)A, lda, (float*)inv, ldi, info, 0, ablate));
}

int pg_rowstep_raw(pg_handle h, int dtype, int n, void* A, long lda, int o0, int k0, const void* inv, int* flags, int* info, void* stream) {
    JOIN(h, stream);
    NEED(h && A && inv && flags && info, "null pointer");
    NEED(n % 128 == 0 && k0 % 128 == 0 && o0 % 128 == 0 && o0 <= k0 && k0 + 128 < n, "bad shape");
    // every flag the kernel waits for is already up (64 >= any count): the kernel alone, nothing to wait for
    PG_CHECK(hipMemsetAsync(flags, 0x40, 8 * sizeof(int), ST(stream)));
    PG_CHECK(hipMemsetAsync(flags + 6, 0, 2 * sizeof(int), ST(stream)));   // [6]: time-out word, [7]: spare
    const CsWait cw = {flags + 6, nullptr, 200000000LL, 1};
    DISPATCH(dtype, pg_rowstep<double>(ST(stream), (double*)A, lda, n, o0, k0, 1, (const double*)inv, flags, flags + 1, flags + 2, cw, info, flags + 3, flags + 4, 0),
             pg_rowstep<float>(ST(stream), (float*)A, lda, n, o0, k0, 1, (const float*)inv, flags, flags + 1, flags + 2, cw, info, flags + 3, flags + 4, 0));
}

int pg_gemm_raw(pg_handle h, int dtype, int variant, int M, int N, int K, double alpha, const void* A, long lda,
                const void* B, long ldb, double beta, void* C, long ldc, int tri, int klo, int khi, void* stream) {
    JOIN(h, stream);
    NEED(h && A && B && C, "null pointer");
    NEED(variant == GEMM_NT_128 || variant == GEMM_NT_RP || variant == GEMM_NN_128 || variant == GEMM_TN_128 ||
             variant == GEMM_TT_128 || variant == GEMM_NT_64 || variant == GEMM_NT_64x128 || variant == GEMM_NT_32x64 ||
             variant == GEMM_NT_32x128 || variant == GEMM_TT_64 || variant == GEMM_NT_32x32 || variant == GEMM_TN_64,
         "variant not exposed");
    AtomicGuard ag(h, C);
    DISPATCH(dtype, gemm_raw_t<double>(h, variant, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, tri, klo, khi, stream),
             gemm_raw_t<float>(h, variant, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, tri, klo, khi, stream));
}

}  // extern "C"
