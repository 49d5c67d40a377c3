// Context, workspace arena and the primitive entry points of the C-ABI (include/ctm_hip.h).
#include "ctm_common.h"
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <dlfcn.h>
#include <mutex>
#include <atomic>

int g_ctm_sync_launch = [] { const char* e = getenv("CTM_SYNC_LAUNCH"); return (e && *e && *e != '0') ? 1 : 0; }();

namespace {

// Diagnostic, off unless the environment holds CTM_ABORT_BACKTRACE=1 (tests/conftest.py sets it): on SIGABRT / SIGSEGV write the
// NATIVE stack of the failing thread to fd 2, then hand the signal to whoever owned it before (Python's faulthandler under pytest,
// else the default action).  A Python traceback ends at the ctypes call; this says which frame below it aborted.
struct sigaction g_prev_abrt, g_prev_segv;

constexpr int LAUNCH_RING = 32;
const char* g_ring[LAUNCH_RING];
std::atomic<unsigned> g_ring_pos{0};

void print_launch_ring() {
    if (!g_ctm_sync_launch) return;
    static const char head[] = "ctm_hip: last kernel launches of the library (oldest first; CTM_SYNC_LAUNCH=1, so the last one is the one in flight):\n";
    (void)!write(2, head, sizeof(head) - 1);
    const unsigned pos = g_ring_pos.load();
    for (unsigned i = pos >= LAUNCH_RING ? pos - LAUNCH_RING : 0; i < pos; ++i) {
        const char* s = g_ring[i % LAUNCH_RING];
        if (!s) continue;
        (void)!write(2, "  ", 2); (void)!write(2, s, strlen(s)); (void)!write(2, "\n", 1);
    }
}

void fatal_backtrace(int sig, siginfo_t*, void*) {
    print_launch_ring();
    static const char head[] = "ctm_hip: fatal signal -- native stack of the failing thread:\n";
    (void)!write(2, head, sizeof(head) - 1);
    void* frames[96];
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    sigaction(sig, sig == SIGABRT ? &g_prev_abrt : &g_prev_segv, nullptr);
    raise(sig);
}

void install_fatal_backtrace() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = getenv("CTM_ABORT_BACKTRACE");
        if (!e || !*e || *e == '0') return;
        void* warm[2];
        (void)backtrace(warm, 2);                  // loads the unwinder now: its first call allocates
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = fatal_backtrace;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER;
        sigemptyset(&sa.sa_mask);
        sigaction(SIGABRT, &sa, &g_prev_abrt);
        sigaction(SIGSEGV, &sa, &g_prev_segv);
    });
}

}  // namespace

void ctm_note_launch(const char* kernel_name) { g_ring[g_ring_pos.fetch_add(1) % LAUNCH_RING] = kernel_name; }

// CTM_ARENA_GUARD=1 (diagnostic, with tools/guard/guard_malloc.cpp preloaded): every arena allocation is its own hipMalloc of exactly
// the requested size (16-byte granular), so a kernel that runs past the end of a workspace buffer faults instead of landing in its
// neighbour on the stack; the blocks of a scope are freed (after a stream sync) when the scope ends.
int g_ctm_arena_guard = [] { const char* e = getenv("CTM_ARENA_GUARD"); return (e && *e && *e != '0') ? 1 : 0; }();

void arena_guard_release(ctm_ctx* ctx, size_t keep) {
    (void)hipStreamSynchronize(ctx->stream);
    Arena& a = ctx->arena;
    while (a.guard_blocks.size() > keep) { (void)hipFree(a.guard_blocks.back()); a.guard_blocks.pop_back(); }
}

int arena_alloc(ctm_ctx* ctx, size_t bytes, void** out) {
    Arena& a = ctx->arena;
    if (g_ctm_arena_guard) {
        void* p = nullptr;
        const size_t b = bytes ? (bytes + 15) & ~(size_t)15 : 16;
        if (hipMalloc(&p, b) != hipSuccess) {
            (void)hipGetLastError();
            ctx->set_error("arena (guard mode): hipMalloc of " + std::to_string(b) + " bytes failed");
            return CTM_ERR_NOMEM;
        }
        a.guard_blocks.push_back(p);
        *out = p;
        return CTM_OK;
    }
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    if (a.cur >= 0 && a.top + bytes <= a.slabs[a.cur].cap) {
        *out = a.slabs[a.cur].base + a.top;
        a.top += bytes;
        return CTM_OK;
    }
    // move to the next slab (reuse it if large enough, else replace it)
    int nxt = a.cur + 1;
    if (nxt < (int)a.slabs.size() && a.slabs[nxt].cap < bytes) {
        // later slabs are unused at this point (stack discipline): free them all and regrow
        (void)hipStreamSynchronize(ctx->stream);
        for (int i = (int)a.slabs.size() - 1; i >= nxt; --i) {
            (void)hipFree(a.slabs[i].base); a.total -= a.slabs[i].cap; a.slabs.pop_back();
        }
    }
    if (nxt >= (int)a.slabs.size()) {
        size_t cap = bytes;
        const size_t min_cap = (size_t)64 << 20;
        if (cap < min_cap) cap = min_cap;
        if (!a.slabs.empty() && cap < a.slabs.back().cap) cap = a.slabs.back().cap;
        Slab s;
        hipError_t e = hipMalloc((void**)&s.base, cap);
        if (e != hipSuccess && cap > bytes) {
            (void)hipGetLastError();          // the failed attempt must not surface as the "launch error" of the next kernel
            cap = bytes; e = hipMalloc((void**)&s.base, cap);
        }
        if (e != hipSuccess) (void)hipGetLastError();
        if (e != hipSuccess) {
            ctx->set_error("arena: hipMalloc of " + std::to_string(cap) + " bytes failed: " + hipGetErrorString(e));
            return CTM_ERR_NOMEM;
        }
        s.cap = cap;
        a.slabs.push_back(s);
        a.total += cap;
        if (a.total > a.high) a.high = a.total;
    }
    a.cur = nxt;
    a.top = bytes;
    *out = a.slabs[a.cur].base;
    return CTM_OK;
}

extern "C" {

const char* ctm_version(void) { return "ctm_hip 0.1 (gfx950, f64)"; }

int ctm_create(ctm_ctx** out, void* hip_stream, int dtype) {
    if (!out) return CTM_ERR_BADARG;
    *out = nullptr;
    if (dtype != CTM_F64 && dtype != CTM_C128) return CTM_ERR_UNSUPPORTED;
    install_fatal_backtrace();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return CTM_ERR_HIP;
    ctm_ctx* c = new ctm_ctx();
    c->cplx = (dtype == CTM_C128);
    (void)hipGetDevice(&c->device);
    if (hip_stream) c->stream = (hipStream_t)hip_stream;
    else { if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return CTM_ERR_HIP; } c->own_stream = true; }
    if (hipMalloc((void**)&c->d_scratch, 64 * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_scratch, 64 * sizeof(double)) != hipSuccess) { delete c; return CTM_ERR_NOMEM; }
    *out = c;
    return CTM_OK;
}

int ctm_destroy(ctm_ctx* ctx) {
    if (!ctx) return CTM_OK;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->arena.slabs) (void)hipFree(s.base);
    for (auto& e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->tile_cnt) (void)hipFree(ctx->tile_cnt);
    if (ctx->comm_own_buf) (void)hipFree(ctx->comm_own_buf);
    if (ctx->h_scratch) (void)hipHostFree(ctx->h_scratch);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CTM_OK;
}

int ctm_trim(ctm_ctx* ctx) {
    return ctm_entry(ctx, "ctm_trim", [&]() -> int {
    // between calls the arena stack is empty: give every slab back to the device (it regrows on demand)
    if (!ctx) return CTM_OK;
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->arena.cur >= 0 || ctx->arena.top != 0) { ctx->set_error("trim: called inside an engine call"); return CTM_ERR_BADARG; }
    for (auto& sl : ctx->arena.slabs) (void)hipFree(sl.base);
    ctx->arena.slabs.clear();
    ctx->arena.total = 0;
    return CTM_OK;
    });
}

const char* ctm_last_error(ctm_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int ctm_sync(ctm_ctx* ctx) {
    return ctm_entry_nolock(ctx, "ctm_sync", [&]() -> int {
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
    });
}

int ctm_set_comm(ctm_ctx* ctx, void* rccl_comm, int rank, int nranks) {
    return ctm_entry_nolock(ctx, "ctm_set_comm", [&]() -> int {
    if (!ctx) return CTM_ERR_BADARG;
    if (nranks < 1 || nranks > 2 || rank < 0 || rank >= nranks) { ctx->set_error("set_comm: nranks must be 1 or 2 and rank inside [0, nranks)"); return CTM_ERR_BADARG; }
    ctx->comm_host_allgather = nullptr; ctx->comm_user = nullptr; ctx->comm_send = nullptr; ctx->comm_recv = nullptr; ctx->comm_cap = 0;
    ctx->comm_nccl_allgather = nullptr;
    if (rccl_comm) {
        // the library does not link librccl: resolve ncclAllGather in the process (torch's bundled librccl when torch.distributed uses the
        // "nccl" backend, or a system librccl)
        void* h = nullptr;
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (!h) h = dlopen(name, RTLD_NOW);
            if (h) break;
        }
        void* fn = h ? dlsym(h, "ncclAllGather") : dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!fn) { ctx->set_error("set_comm: ncclAllGather could not be resolved (librccl.so not loadable in this process)"); return CTM_ERR_UNSUPPORTED; }
        ctx->comm_nccl_allgather = fn;
    } else if (nranks > 1) { ctx->set_error("set_comm: a group of more than one rank needs a communicator (or ctm_set_comm_ops)"); return CTM_ERR_BADARG; }
    ctx->comm = rccl_comm; ctx->comm_rank = rank; ctx->comm_nranks = nranks;
    return CTM_OK;
    });
}

int ctm_set_comm_ops(ctm_ctx* ctx, ctm_allgather_fn allgather, void* user, double* send_buf, double* recv_buf, long long capacity_doubles,
                     int rank, int nranks) {
    return ctm_entry_nolock(ctx, "ctm_set_comm_ops", [&]() -> int {
    if (!ctx) return CTM_ERR_BADARG;
    if (nranks < 1 || nranks > 2 || rank < 0 || rank >= nranks) { ctx->set_error("set_comm_ops: nranks must be 1 or 2 and rank inside [0, nranks)"); return CTM_ERR_BADARG; }
    if (allgather && (!send_buf || !recv_buf || capacity_doubles < 1)) { ctx->set_error("set_comm_ops: staging buffers missing"); return CTM_ERR_BADARG; }
    ctx->comm = nullptr; ctx->comm_nccl_allgather = nullptr;
    ctx->comm_host_allgather = allgather; ctx->comm_user = user;
    ctx->comm_send = allgather ? send_buf : nullptr; ctx->comm_recv = allgather ? recv_buf : nullptr; ctx->comm_cap = allgather ? capacity_doubles : 0;
    ctx->comm_rank = allgather ? rank : 0; ctx->comm_nranks = allgather ? nranks : 1;
    return CTM_OK;
    });
}

int ctm_set_option(ctm_ctx* ctx, const char* key, double value) {
    return ctm_entry_nolock(ctx, "ctm_set_option", [&]() -> int {
    // 37 keys (tests/test_gpu_options.py names every one).  Everything else that used to be settable is a constant of the build: the
    // tuning parameters keep their measured values as member defaults of ctm_ctx (ctm_common.h), the variants that lost their
    // measurement (one-launch Jacobi sweep, rotation lists, the device-side lock around chip-filling launches, the LDS-staged two-layer
    // kernel, 1 / 4 blocks per thread in the 64 x 64 eigensolver) are gone.
    const std::string k(key ? key : "");
    // -- accuracy / semantics
    if (k == "jacobi_tol") ctx->jacobi_tol = value;
    else if (k == "jacobi_max_sweeps") ctx->jacobi_max_sweeps = (int)value;
    else if (k == "svd_null_tol") ctx->svd_null_tol = value;
    else if (k == "rank_tol") ctx->rank_tol = value;
    else if (k == "si_tol") ctx->si_tol = value;
    else if (k == "svd_abs_accuracy") ctx->svd_abs_accuracy = (int)value;
    else if (k == "svd_polar") ctx->svd_polar = (int)value;
    // -- which solver / kernel route
    else if (k == "si_enable") ctx->si_enable = value != 0.0;
    else if (k == "si_min_n") ctx->si_min_n = (int)value;
    else if (k == "si_max_iter") ctx->si_max_iter = (int)value;
    else if (k == "lz_enable") ctx->lz_enable = value != 0.0;
    else if (k == "lz_min_k") ctx->lz_min_k = (int)value;
    else if (k == "lz_block") ctx->lz_block = (int)value;
    else if (k == "lz_block_c") ctx->lz_block_c = (int)value;
    else if (k == "lz_async") ctx->lz_async = value != 0.0;
    else if (k == "lz_local_project") ctx->lz_local_project = value != 0.0;
    else if (k == "lz_verify_op") ctx->lz_verify_op = value != 0.0;
    else if (k == "jacobi_cross_only") ctx->jacobi_cross_only = (int)value;
    else if (k == "eigh_warm") ctx->eigh_warm = (int)value;
    else if (k == "eigh_orth_iter") ctx->eigh_orth_iter = (int)value;
    else if (k == "eigh_orth_double") ctx->eigh_orth_double = (int)value;
    else if (k == "proj_from_krylov") ctx->proj_from_krylov = value != 0.0;
    else if (k == "use_layer2") ctx->use_layer2 = value != 0.0;
    else if (k == "gemm_fast") ctx->gemm_fast = value != 0.0;
    else if (k == "xgemm_stack_rows") ctx->xgemm_stack_rows = value != 0.0;
    // -- stationary fast path of the implicit-operator truncation (ctm_args.projector_warm_tol)
    else if (k == "warm_accept_tol") ctx->warm_accept_tol = value;
    else if (k == "ritz_warm") ctx->ritz_warm = (int)value;
    else if (k == "lz_two_pass") ctx->lz_two_pass = (int)value;
    else if (k == "sign_follow") ctx->sign_follow = (int)value;
    else if (k == "warm_try_factor") ctx->warm_try_factor = value;
    else if (k == "warm_accept_max_run") ctx->warm_accept_max_run = (int)value;
    // -- row-block GEMM: the knobs tests/test_gpu_gemm_rows.py needs to reach every epilogue
    else if (k == "rows_fused_reduce") ctx->rows_fused_reduce = value != 0.0;
    else if (k == "rows_kernel_min_m") ctx->rows_kernel_min_m = (int)value;
    else if (k == "rows_kernel_min_m_kc") ctx->rows_kernel_min_m_kc = (int)value;
    else if (k == "rows_min_klen") ctx->rows_min_klen = (int)value;
    else if (k == "rows_target_wgs") ctx->rows_target_wgs = (int)value;
    // -- instrumentation
    else if (k == "timing_min_flops") ctx->timing_min_flops = value;
    else if (k == "profile") ctx->profile = value != 0.0;
    else if (k == "jacobi_verbose") ctx->jacobi_verbose = (int)value;
    else if (k == "gemm_timing") {
        gemm_timing_drain(ctx);
        ctx->gemm_timing = value != 0.0;
        if (ctx->gemm_timing) { gemm_timing_base(ctx); ctx->intervals.clear(); }
        for (int i = 0; i < 5; ++i) { ctx->k_ms[i] = 0; ctx->k_flops[i] = 0; ctx->k_calls[i] = 0; }
    }
    else { ctx->set_error("unknown option " + k); return CTM_ERR_BADARG; }
    return CTM_OK;
    });
}

int ctm_get_stat(ctm_ctx* ctx, const char* key, double* value) {
    return ctm_entry_nolock(ctx, "ctm_get_stat", [&]() -> int {
    const std::string k(key ? key : "");
    if (k == "last_sweeps") *value = ctx->last_sweeps;
    else if (k == "last_offnorm") *value = ctx->last_offnorm;
    else if (k == "total_sweeps") *value = (double)ctx->total_sweeps;
    else if (k == "jacobi_calls") *value = (double)ctx->jacobi_calls;
    else if (k == "si_hits") *value = (double)ctx->si_hits;
    else if (k == "si_fallbacks") *value = (double)ctx->si_fallbacks;
    else if (k == "si_last_iters") *value = (double)ctx->si_last_iters;
    else if (k == "si_total_iters") *value = (double)ctx->si_total_iters;
    else if (k == "si_last_rank") *value = (double)ctx->si_last_rank;
    else if (k == "si_warm_skips") *value = (double)ctx->si_warm_skips;
    else if (k == "corner_cache_hits") *value = (double)ctx->corner_cache_hits;
    else if (k == "si_warm_starts") *value = (double)ctx->si_warm_starts;
    else if (k == "eigh_warm_hits") *value = (double)ctx->eigh_warm_hits;
    else if (k == "eigh_warm_rejects") *value = (double)ctx->eigh_warm_rejects;
    else if (k == "eigh_orth_hits") *value = (double)ctx->eigh_orth_hits;
    else if (k == "eigh_orth_fails") *value = (double)ctx->eigh_orth_fails;
    else if (k == "eigh_orth_doubled") *value = (double)ctx->eigh_orth_doubled;
    else if (k == "svd_polar_completions") *value = (double)ctx->svd_polar_completions;
    else if (k == "svd_eig_completions") *value = (double)ctx->svd_eig_completions;
    else if (k == "svd_polar_solves") *value = (double)ctx->svd_polar_solves;
    else if (k == "lz_hits") *value = (double)ctx->lz_hits;
    else if (k == "warm_accepts") *value = (double)ctx->warm_accepts;
    else if (k == "warm_rejects") *value = (double)ctx->warm_rejects;
    else if (k == "warm_last_dist") *value = ctx->warm_last_dist;
    else if (k == "lz_total_steps") *value = (double)ctx->lz_total_steps;
    else if (k == "lz_total_rows") *value = (double)ctx->lz_total_rows;
    else if (k == "lz_extractions") *value = (double)ctx->lz_extractions;
    else if (k == "ritz_warm_starts") *value = (double)ctx->ritz_warm_starts;
    else if (k == "comm_calls") *value = (double)ctx->comm_calls;
    else if (k == "comm_doubles") *value = ctx->comm_doubles;
    else if (k == "ritz_sweeps") *value = (double)ctx->ritz_sweeps;
    else if (k == "lz_last_est") *value = ctx->lz_last_est;
    else if (k == "lz_async_fallbacks") *value = (double)ctx->lz_async_fallbacks;
    else if (k == "lz_third_passes") *value = (double)ctx->lz_third_passes;
    else if (k == "lz_last_steps") *value = (double)ctx->lz_last_steps;
    else if (k == "gemm_flops") *value = ctx->gemm_flops;
    else if (k == "gemm_calls") *value = (double)ctx->gemm_calls;
    else if (k == "layer2_flops") *value = ctx->layer2_flops;
    else if (k == "layer2_calls") *value = (double)ctx->layer2_calls;
    else if (k == "arena_high") *value = (double)ctx->arena.high;
    else if (k == "arena_total") *value = (double)ctx->arena.total;          // bytes this context holds right now
    else if (k == "absorb_bytes") *value = ctx->absorb_bytes;
    else if (k == "absorb_calls") *value = (double)ctx->absorb_calls;
    else if (k.rfind("k_", 0) == 0 && k.size() >= 5) {      // k_ms0, k_ms1, k_flops0, k_flops1, k_calls0, k_calls1
        gemm_timing_drain(ctx);
        const int i = k.back() - '0';
        if (i < 0 || i > 4) { ctx->set_error("unknown stat " + k); return CTM_ERR_BADARG; }
        if (k.compare(0, 4, "k_ms") == 0) *value = ctx->k_ms[i];
        else if (k.compare(0, 7, "k_flops") == 0) *value = ctx->k_flops[i];
        else if (k.compare(0, 7, "k_calls") == 0) *value = (double)ctx->k_calls[i];
        else { ctx->set_error("unknown stat " + k); return CTM_ERR_BADARG; }
    }
    else { ctx->set_error("unknown stat " + k); return CTM_ERR_BADARG; }
    return CTM_OK;
    });
}

int ctm_gemm_intervals(ctm_ctx* ctx, double* out, long long capacity, long long* count) {
    return ctm_entry_nolock(ctx, "ctm_gemm_intervals", [&]() -> int {
    gemm_timing_drain(ctx);
    const long long n = (long long)ctx->intervals.size();
    if (count) *count = n / 4;
    if (out) for (long long i = 0; i < n && i < capacity; ++i) out[i] = ctx->intervals[i];
    return CTM_OK;
    });
}

int ctm_timers(ctm_ctx* ctx, double* out8, int reset) {
    return ctm_entry_nolock(ctx, "ctm_timers", [&]() -> int {
    gemm_timing_drain(ctx);       // event-timed phases are accumulated when their events are read
    for (int i = 0; i < CTM_T_COUNT; ++i) { if (out8) out8[i] = ctx->timers[i]; if (reset) ctx->timers[i] = 0.0; }
    if (reset) { ctx->gemm_flops = 0; ctx->gemm_calls = 0; ctx->layer2_flops = 0; ctx->layer2_calls = 0; ctx->total_sweeps = 0; ctx->jacobi_calls = 0; ctx->si_hits = 0; ctx->si_fallbacks = 0; ctx->si_total_iters = 0; ctx->si_warm_starts = 0; ctx->eigh_warm_hits = 0; ctx->eigh_warm_rejects = 0; ctx->eigh_orth_hits = 0; ctx->eigh_orth_fails = 0; ctx->eigh_orth_doubled = 0; ctx->svd_polar_completions = 0; ctx->svd_eig_completions = 0; ctx->svd_polar_solves = 0; ctx->lz_hits = 0; ctx->lz_total_steps = 0; ctx->lz_total_rows = 0; ctx->lz_extractions = 0; ctx->ritz_warm_starts = 0; ctx->ritz_sweeps = 0; ctx->absorb_bytes = 0; ctx->absorb_calls = 0; }
    return CTM_OK;
    });
}

int ctm_gemm(ctm_ctx* ctx, int transA, int transB, int M, int N, int K, double alpha, const double* A, long long lda,
             const double* B, long long ldb, double beta, double* C, long long ldc) {
    return ctm_entry(ctx, "ctm_gemm", [&]() -> int {
    if (!ctx->cplx) {
        GemmDesc d;
        d.M = M; d.N = N; d.K = K; d.alpha = alpha; d.beta = beta;
        d.A = A; if (transA) { d.sam = 1; d.sak = lda; } else { d.sam = lda; d.sak = 1; }
        d.B = B; if (transB) { d.sbk = 1; d.sbn = ldb; } else { d.sbk = ldb; d.sbn = 1; }
        d.C = C; d.ldc = ldc;
        return gemm_f64(ctx, d);
    }
    // complex128: trans = 0 'N', 1 'T', 2 'C' (conjugate transpose); dense operands (ld == row length), alpha real, beta == 0
    const long long ra = transA ? K : M, ca = transA ? M : K, rb = transB ? N : K, cb = transB ? K : N;
    if (lda != ca || ldb != cb || ldc != N || beta != 0.0) { ctx->set_error("gemm(c128): dense operands and beta == 0 only"); return CTM_ERR_UNSUPPORTED; }
    ArenaScope scope(ctx);
    double *a, *b, *c;
    const size_t na = (size_t)ra * ca, nb = (size_t)rb * cb, nc = (size_t)M * N;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * na, (void**)&a));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * nb, (void**)&b));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * nc, (void**)&c));
    CTM_TRY(deinterleave_c128(ctx, A, a, a + na, na));
    CTM_TRY(deinterleave_c128(ctx, B, b, b + nb, nb));
    XM xa{a, a + na, lda, transA != 0, transA == 2}, xb{b, b + nb, ldb, transB != 0, transB == 2};
    CTM_TRY(xgemm(ctx, M, N, K, xa, xb, c, c + nc, N));
    if (alpha != 1.0) {
        double* s = ctx->d_scratch + 9;
        const double inv = 1.0 / alpha;
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(s, &inv, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        CTM_TRY(div_by_device_scalar(ctx, c, 2 * nc, s, 0));
    }
    return interleave_c128(ctx, c, c + nc, C, nc);
    });
}

int ctm_permute(ctm_ctx* ctx, const double* in, double* out, int nd, const long long* dims, const int* perm) {
    return ctm_entry(ctx, "ctm_permute", [&]() -> int {
    if (!ctx->cplx) return permute_f64(ctx, in, out, nd, dims, perm);
    if (nd + 1 > CTM_MAXD) { ctx->set_error("permute(c128): rank"); return CTM_ERR_UNSUPPORTED; }
    long long d2[CTM_MAXD]; int p2[CTM_MAXD];
    for (int i = 0; i < nd; ++i) { d2[i] = dims[i]; p2[i] = perm[i]; }
    d2[nd] = 2; p2[nd] = nd;                       // the (re,im) pair travels as an innermost axis
    return permute_f64(ctx, in, out, nd + 1, d2, p2);
    });
}

int ctm_normalize_inf(ctm_ctx* ctx, double* x, long long n) {
    return ctm_entry(ctx, "ctm_normalize_inf", [&]() -> int {
    double* s = ctx->d_scratch + 8;
    if (!ctx->cplx) {
        CTM_TRY(absmax_f64(ctx, x, (size_t)n, s));
        return div_by_device_scalar(ctx, x, (size_t)n, s, 0);
    }
    ArenaScope scope(ctx);
    double* pl;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)n, (void**)&pl));
    CTM_TRY(deinterleave_c128(ctx, x, pl, pl + n, (size_t)n));
    CTM_TRY(absmax_c128(ctx, pl, pl + n, (size_t)n, s));
    return div_by_device_scalar(ctx, x, 2 * (size_t)n, s, 0);
    });
}

}  // extern "C"
