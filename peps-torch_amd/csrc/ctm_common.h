// Internal (non-ABI) declarations shared by the HIP translation units of libctm_hip.so.
// gfx950 / CDNA4 only.  The public C-ABI is include/ctm_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <chrono>
#include <atomic>
#include <thread>
#include <new>

#include "../../include/ctm_hip.h"

#define CTM_HIP_CHECK(ctx, expr)                                                        \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            (ctx)->set_error(std::string(#expr) + ": " + hipGetErrorString(_e));        \
            return CTM_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

// Diagnostics (environment, read once): CTM_SYNC_LAUNCH=1 waits for the stream after EVERY kernel launch and keeps the names of the last
// launches in a process-wide ring that the fatal-signal handler of ctm_runtime.hip prints -- a GPU memory fault (raised
// asynchronously by the HSA runtime, normally long after the offending launch) then names its kernel.
extern int g_ctm_sync_launch;
void ctm_note_launch(const char* kernel_name);

// kernel launch on the context's stream with the launch error checked (a failed launch must not come back as CTM_OK)
#define CTM_LAUNCH(ctx, kernel, grid, block, shmem, ...)                                              \
    do {                                                                                              \
        if (g_ctm_sync_launch) ctm_note_launch(#kernel);                                              \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);                   \
        hipError_t _le = hipGetLastError();                                                           \
        if (_le == hipSuccess && g_ctm_sync_launch) _le = hipStreamSynchronize((ctx)->stream);        \
        if (_le != hipSuccess) {                                                                      \
            (ctx)->set_error(std::string("launch of " #kernel ": ") + hipGetErrorString(_le));        \
            return CTM_ERR_HIP;                                                                       \
        }                                                                                             \
    } while (0)

constexpr int CTM_TILE_COUNTERS = 8192;

#define CTM_TRY(expr)                        \
    do {                                     \
        int _s = (expr);                     \
        if (_s != CTM_OK) return _s;         \
    } while (0)

// ---- device workspace arena: stack allocator over a chain of slabs, grown on demand ---------
struct Slab { char* base = nullptr; size_t cap = 0; };
struct Arena {
    std::vector<Slab> slabs;
    int cur = -1;                  // current slab
    size_t top = 0;                // offset in the current slab
    size_t high = 0, total = 0;
    std::vector<void*> guard_blocks;   // CTM_ARENA_GUARD=1 (diagnostic): every allocation its own hipMalloc, freed when its scope ends
};
extern int g_ctm_arena_guard;
void arena_guard_release(struct ctm_ctx* ctx, size_t keep);

// per-phase timers (seconds, host wall time with stream sync when profiling is enabled)
enum { CTM_T_CORNERS = 0, CTM_T_HALVES, CTM_T_SVD, CTM_T_PROJ, CTM_T_ABSORB, CTM_T_NORM, CTM_T_RDM, CTM_T_EIG, CTM_T_COUNT };

struct ctm_ctx {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int device = 0;
    Arena arena;
    std::string last_error;
    // jacobi state
    int jacobi_block = 32;
    int jacobi_max_sweeps = 30;
    double jacobi_tol = 1e-14;
    int si_tau_both = 1;               // Rayleigh-Ritz of the subspace iteration: guard-guard pairs only are measured against the k-th row norm
    int jacobi_gram_kmin = 256, jacobi_gram_kmin_short = 64;   // pair Gram GEMMs: K is split over workgroups down to this many columns each (rows longer / not longer than 2048)
    int jacobi_tau_relax = 1;          // rows below the k-th largest norm are measured against it (development switch)
    double svd_null_tol = 1e-11;       // full decomposition: right vectors of s_i <= svd_null_tol s_0 are completed orthonormally (svd_full)
    int jacobi_inner_sweeps = 2;        // inner sweeps of the LDS eigensolver per visit of a pair (2 or 3 pairs per round)
    int jacobi_inner_sweeps_many = 1;   // ... when a round has >= 4 pairs (dense small SVDs, full-block Rayleigh-Ritz): measured faster
    int jacobi_cross_only = 1;          // many-panel block Jacobi: only the first round of a sweep solves the full 64 x 64 pair problems, the others rotate cross pairs only
    int jacobi_verbose = 0;
    // leading-k block power iteration (svd_iter): enabled for n >= si_min_n, residual tolerance relative to s_0
    bool si_enable = true;
    int si_min_n = 256, si_max_iter = 40, si_last_iters = 0, si_rr_sweeps = 40, si_last_rank = 0;
    bool si_block32 = true;       // warm start with numerical rank <= 24: 32-row block
    int si_warm_skip_calls = 12;  // upper bound on the calls of a unit that start cold after its warm probe went to the Krylov solver
    long si_warm_skips = 0;
    long corner_cache_hits = 0;
    double si_tol = 2e-14;
    double rank_tol = 5e-13;             // numerical-rank threshold of the leading-k solvers (relative to s_0)
    long si_hits = 0, si_fallbacks = 0, si_total_iters = 0, si_warm_starts = 0;
    long svd_polar_completions = 0, svd_eig_completions = 0;     // rank-deficient full SVDs: V completed by the polar iteration / by eigenvectors
    int svd_polar = 1, svd_polar_min_n = 96;   // full real SVD with vectors (differentiable route): polar decomposition + shifted symmetric Jacobi
    long svd_polar_solves = 0;
    int lz_abs_accuracy = 0;           // experiment: absolute criterion for the Ritz extraction of the block Krylov solver
    bool force_abs = false;
    int svd_abs_accuracy = 1;          // full SVD with vectors (differentiable route): row pairs orthogonalised to tol * s_0 absolute (see tau_floor)
    int eigh_warm = 1;                 // symmetric problems with a warm basis: Rayleigh-Ritz in the warm subspace + deflated probe first
    long eigh_warm_hits = 0, eigh_warm_rejects = 0, eigh_probe_calls = 0;
    int eigh_probe_orth_once = 1;      // ... the probe block is orthonormalised once (before its last application) instead of after each of the first two
    int eigh_warm_early_reject = 1;    // ... a subspace whose residual |Y - (Y Q^T) Q|_F already exceeds the threshold is refused before its Rayleigh-Ritz
    int eigh_orth_iter = 1;            // refused warm restart (the matrix moved): symmetric orthogonal iteration with Cholesky-QR steps and ONE Rayleigh-Ritz
    int eigh_orth_max = 32;            // ... applications before it gives up (the regular block iteration runs then; an application + Cholesky-QR
                                       //     step costs a quarter of a half step of that iteration, so a slowly contracting block stays here)
    int eigh_orth_extra_blocks = 0;    // ... additional 64-row blocks of guard rows
    int eigh_orth_double = 2;          // ... two applications per Cholesky-QR step (1), shifted: Q (A^2 - c^2/2) (2; default since round 5: the whole GPU suite, 386 tests, ran with it)
    double eigh_orth_double_min_ratio = 1e-3;   // ... only while |theta_kk| / |theta_0| of the previous look is above this (the kept block's condition number is its inverse square)
    int eigh_orth_predict = 1;         // ... its looks (Rayleigh-Ritz + residual test) are placed where the residual is predicted to pass
    double eigh_orth_quad_exit = 1e-9; // ... early exit of its small Jacobi eigensolver (see lz_quad_exit; the residual test certifies what it returns)
    long eigh_orth_hits = 0, eigh_orth_fails = 0, eigh_orth_doubled = 0;      // (doubled: Cholesky-QR steps that followed two applications)
                                       // (its contraction rate / back-off state is kept per warm workspace, see OrthState in eigh.hip)
    // block Golub-Kahan-Lanczos for spectra that do not collapse inside a small block (svd_lanczos)
    bool lz_enable = true; int lz_min_k = 48; double lz_switch_steps = 6.0; double lz_last_resid = 1.0; long lz_hits = 0, lz_total_steps = 0;
    int lz_first = 0;                   // > 0: first Ritz extraction after this many block steps (development); 0: policy of svd_lanczos
    int lz_stride = 0;                  // > 0: fixed distance between Ritz extractions (development); 0: predicted from the residual estimate
    double lz_first_factor = 3.25;      // cold default: first extraction when the basis holds this many times k rows
    double lz_first_factor32 = 2.5;     // ... with 32-row blocks
    int lz_block = 0;                   // rows per block of the real block Krylov recurrence: 64, 32, or 0 = 32 for k > lz_block32_min_k
    int lz_block32_min_k = 0;           // (32-row blocks for every k: D = 6 chi = 128 full rank 0.49-0.57 -> 0.39 s/sweep, D = 8 chi = 256 3.26 -> 3.12)
    int lz_block_c = 32;                // complex rows per block of the complex recurrence (64 or 32; D = 8 chi = 384: 34.3 -> 31.6 s per full-rank sweep)
    long lz_total_rows = 0;             // basis rows over all accepted solves (steps x block)
    bool lz_verify_op = false;          // additionally check both relations of the Ritz triplets with operator applications (debug / tests)
    long lz_extractions = 0; double lz_last_est = 0.0; int lz_last_steps = 0;
    // Ritz extraction of the block Krylov solvers started from the accumulated rotations of the unit's previous extraction ("ritz_warm", svd_full: rot)
    int ritz_warm = 1; long ritz_warm_starts = 0, ritz_sweeps = 0;
    // orientation of the returned singular vectors of a warm-started unit follows its previous decomposition (fix_signs_rows_kernel: ref) also
    // without the stationary fast path: the largest-element rule re-gauges legs of the environment by signs from sweep to sweep, which makes
    // the operator of the next sweep a DIFFERENT matrix (measured D = 4 chi = 64: |dM| / |M| = 5e-3 ... 7e-2 with | |M| - |M'| | = 5e-5) and
    // its Krylov basis / Ritz matrix unrelated to the previous one
    int sign_follow = 1;
    double jacobi_quad_exit = 0.0;      // (internal) jacobi_rows stops after a sweep that FOUND <= this measure (quadratic regime)
    double si_quad_exit = 0.0;          // ... optionally during the Rayleigh-Ritz of the subspace iteration (off: measured no gain -- the sweep it saves finds every
                                        // pair below tolerance, and such a sweep costs ~10 us per round: the eigensolver exits early, the apply GEMMs are skipped)
    double lz_quad_exit = 1e-9;         // ... during the Ritz extraction of the block Krylov solver, whose triplets are verified afterwards
    bool lz_async = true;               // block Krylov recurrence issued without host synchronisations (status words checked at the extraction)
    bool lz_force_sync = false;         // (internal) the current solve is being repeated on the synchronous path
    long lz_async_fallbacks = 0, lz_third_passes = 0;
    int lz_two_pass = 1;                // orthonormalise_block_async with two passes for units whose previous solve needed no third (svd_lanczos)
    bool lz_local_project = true;       // first Gram-Schmidt pass of a block step against the previous block only, second against all
    int last_sweeps = 0;
    long total_sweeps = 0, jacobi_calls = 0;
    double last_offnorm = 0;
    std::map<int, int*> rr_tables;       // nblocks -> device round-robin pair table
    double* d_scratch = nullptr;         // small device scalars (64 doubles)
    double* h_scratch = nullptr;         // pinned host mirror
    // timers
    bool profile = false;
    double timers[CTM_T_COUNT] = {0};
    // GEMM instrumentation (flop count of all GEMM launches)
    double gemm_flops = 0;
    long gemm_calls = 0;
    double layer2_flops = 0;
    long layer2_calls = 0;
    double absorb_bytes = 0;            // algorithmic HBM bytes of the absorb calls (operands read once, results written once)
    long absorb_calls = 0;
    bool use_layer2 = true;
    bool gemm_fast = true;
    bool einsum_in_relayout = true, z_spectators_first = true;   // layout of the fused two-layer kernel's input (contract.hip)
    bool proj_from_krylov = true; // projectors from the half-way products the block Krylov solver stored (no corner passes after the truncation)
    bool chain_as_strips = true;  // projector columns (<= 64) kept as rows through the two corner passes
    int splitk_reduce_vec = 1;
    bool gemm_log = false;        // debug: print every GEMM shape to stderr
    bool gemm_strip = true;       // streaming kernel for <= 64 rows times a big operand
    int strip_target_wgs = 512;   // K slices x column tiles of the strip kernel: fewer slices = fewer partials (measured 512 <= 1024, 256)
    int rows_kernel_min_m = 1, rows_kernel_min_m_kc = 1;   // LDS-tiled row-block kernel from this many rows (n-contiguous / k-contiguous big operand)
    int rows_min_klen = 576;      // ... lower bound on the K slice: mid-size operands (n = 4608 = 36 column tiles) otherwise run 18 slices of 256 k that are all
                                  //     prologue, epilogue and an 18-slab combine (D = 6 chi = 128 sweep +8-12 %; n >= 12288 keeps its slice count).  Round 3 saw one
                                  //     full test run with 576 end in a core dump and suspected this option; round 4 cleared it (shape sweep through every epilogue,
                                  //     tests/test_gpu_gemm_rows.py, ks = 1 included; the whole suite under AddressSanitizer) -- the crash sits elsewhere
    int rows_min_klen_hbm = 576;      // ... the same bound for <= 32-row blocks (HBM-bound)
    bool rows_deep_prefetch = true;   // ... two K tiles in flight per workgroup when at most two workgroups share a CU (mid-size operands)
    bool rows_quantise = true;    // ... its slice count is rounded down so that the last round of workgroups over the 256 CUs is nearly full
    int rows_target_wgs = 512;    // its workgroup count (column tiles x K slices): two per CU.  (768 until round 4: alone the same speed; with four units streaming
                                  //     four corners 256 ... 640 all give 2.95-2.99 s per full-rank D = 8 sweep against 3.06 with 768: fewer, longer slices, fewer slabs to combine)
    // units of a move overlap their latency-bound stages with each other and with ONE corner pass at a time, instead of four
    double timing_min_flops = 5e9;      // event pairs only around the chip-filling launches (corner passes, corner builds, absorb GEMMs): pairs around the ~170 projection GEMMs of a unit (7e8 flop each) cost the full-rank sweep 1.6 %
    bool rows_fused_reduce = true;      // its K-slice partials are summed inside the launch by the last workgroup of a column tile
    unsigned* tile_cnt = nullptr;       // per-column-tile arrival counters of that combine (zero between launches)
    bool xgemm_stack_rows = true; // complex row blocks (<= 64 rows, planes contiguous): two real products on the stacked 2M rows instead of four
    bool gemm_split_rem = true;   // split a 128 q + r (r <= 64) dimension into a vectorised part and a strip
    int splitk_max_tiles = 256, splitk_target_wgs = 1024;   // split-K of skinny GEMMs: when few output tiles, how many workgroups to aim for
    bool layer2_cplx = true;            // fused kernel for complex128 operands too
    // optional per-launch HIP-event timing of the GEMM kernels on ctx->stream (bench roofline):
    // kind 0 = 128x128 tile kernel, kind 1 = 64x64 tile kernel
    bool gemm_timing = false;
    std::vector<hipEvent_t> ev_pool;
    struct PendingEv { int e0, e1, kind; double flops; };
    std::vector<PendingEv> ev_pending;
    int ev_next = 0;
    std::vector<double> intervals;       // (kind, start_ms, end_ms, flops) per timed GEMM launch, process-wide clock
    // per kernel class: 0 = 128-tile GEMMs, 1 = other GEMMs, 2 = fused two-layer kernel (layer2.hip), 3 = streaming strip kernel
    // (k_flops[3] holds ALGORITHMIC BYTES: that kernel is HBM-bound)
    // 4 = the same row-block products with 33..64 rows: MFMA-bound (16 flop per byte of the big operand), k_flops[4] holds flops
    double k_ms[5] = {0, 0, 0, 0, 0}, k_flops[5] = {0, 0, 0, 0, 0};
    long k_calls[5] = {0, 0, 0, 0, 0};
    void* comm = nullptr; int comm_rank = 0, comm_nranks = 1;   // rank group sharing one unit (ctm_set_comm; column split: include/ctm_hip.h)
    // all-gather of the group: RCCL on this context's stream (comm_nccl_allgather = ncclAllGather resolved with dlopen) or a host callback
    void* comm_nccl_allgather = nullptr;
    int (*comm_host_allgather)(void*, long long) = nullptr; void* comm_user = nullptr;
    double* comm_send = nullptr; double* comm_recv = nullptr; long long comm_cap = 0;      // host-callback staging buffers (caller-owned)
    double* comm_own_buf = nullptr; long long comm_own_cap = 0;                            // RCCL staging buffers (library-owned: send | recv)
    long comm_calls = 0; double comm_doubles = 0.0;
    bool cplx = false;                   // CTM_C128 context: every tensor pointer of the C-ABI is interleaved complex128
    alignas(8) unsigned char orth_cur_storage[64] = {0};      // adaptive state of the symmetric orthogonal iteration for the workspace of the CURRENT call (eigh.hip: OrthState)
    // stationary fast path of the implicit-operator truncation (svd_leading.hip: svd_stationary).  0 = off: every truncation is solved to resid_tol
    double warm_accept_tol = 0.0;        // accept one Rayleigh-Ritz half step from the previous basis when its residual is <= this x s_0
    double warm_try_factor = 1e-4;       // ... tried when the unit's normalised singular values moved by at most this x warm_accept_tol between its last two solves (a
                                         // LOWER bound on the movement of the operator: measured on signed D = 6 chi = 128, the residual of the previous triplets passes
                                         // 1e-9 s_0 once the values are stationary to ~1e-13; a refused attempt costs a fifth of a solve and backs off x2)
    int warm_accept_max_run = 32;        // ... at most this many accepted calls of a unit between two full solves (0: no limit)
    long warm_accepts = 0, warm_rejects = 0;
    double warm_last_dist = 0.0;
    // one call at a time: a context owns ONE arena stack and ONE stream, so two threads inside it at once corrupt both silently.
    // EntryGuard (below) makes that a loud CTM_ERR_BUSY instead (same-thread nesting -- an entry implemented by another -- is fine)
    std::atomic<int> busy{0};
    std::atomic<std::thread::id> owner{std::thread::id()};      // read by a contending thread while the owning one writes it: atomic, not a plain member
    int depth = 0;                                              // touched by the owning thread only (after ownership is established)
    void set_error(const std::string& s) { last_error = s; }
};

struct EntryGuard {
    ctm_ctx* c; bool ok;
    explicit EntryGuard(ctm_ctx* ctx) : c(ctx), ok(true) {
        const std::thread::id me = std::this_thread::get_id();
        int expect = 0;
        if (c->busy.compare_exchange_strong(expect, 1, std::memory_order_acquire)) { c->owner.store(me, std::memory_order_relaxed); c->depth = 1; }
        else if (c->owner.load(std::memory_order_relaxed) == me) ++c->depth;      // only the owner itself can read its own id here
        else ok = false;
    }
    ~EntryGuard() { if (ok && --c->depth == 0) { c->owner.store(std::thread::id(), std::memory_order_relaxed); c->busy.store(0, std::memory_order_release); } }
};

// Body of every compute entry of the C-ABI: refuses concurrent use of one context, and no C++ exception crosses the boundary
// (std::bad_alloc -> CTM_ERR_NOMEM, anything else -> CTM_ERR_HIP with its message in ctm_last_error).
template <class F>
int ctm_entry_nolock(ctm_ctx* ctx, const char* name, F&& body) {      // options, statistics, stream sync: callable from any thread
    if (!ctx) return CTM_ERR_BADARG;
    try { return body(); }
    catch (const std::bad_alloc&) { ctx->set_error(std::string(name) + ": host allocation failed (std::bad_alloc)"); return CTM_ERR_NOMEM; }
    catch (const std::exception& e) { ctx->set_error(std::string(name) + ": C++ exception: " + e.what()); return CTM_ERR_HIP; }
    catch (...) { ctx->set_error(std::string(name) + ": unknown C++ exception"); return CTM_ERR_HIP; }
}

template <class F>
int ctm_entry(ctm_ctx* ctx, const char* name, F&& body) {
    if (!ctx) return CTM_ERR_BADARG;
    EntryGuard g(ctx);
    if (!g.ok) return CTM_ERR_BUSY;       // (last_error belongs to the thread that is inside: not touched)
    try { return body(); }
    catch (const std::bad_alloc&) { ctx->set_error(std::string(name) + ": host allocation failed (std::bad_alloc)"); return CTM_ERR_NOMEM; }
    catch (const std::exception& e) { ctx->set_error(std::string(name) + ": C++ exception: " + e.what()); return CTM_ERR_HIP; }
    catch (...) { ctx->set_error(std::string(name) + ": unknown C++ exception"); return CTM_ERR_HIP; }
}

// arena API (ctm_runtime.hip)
int arena_alloc(ctm_ctx* ctx, size_t bytes, void** out);
struct ArenaScope {
    ctm_ctx* c; int cur; size_t top; size_t nguard;
    explicit ArenaScope(ctm_ctx* ctx) : c(ctx), cur(ctx->arena.cur), top(ctx->arena.top), nguard(ctx->arena.guard_blocks.size()) {}
    ~ArenaScope() { c->arena.cur = cur; c->arena.top = top; if (c->arena.guard_blocks.size() > nguard) arena_guard_release(c, nguard); }
};

// event pair around a launch on ctx->stream while "gemm_timing" is on: begin returns the slot (or -1), end files it under `kind`
int timing_begin(ctm_ctx* ctx);
void timing_end(ctm_ctx* ctx, int e0, int kind, double flops);
constexpr int CTM_KIND_PHASE0 = 16;      // timing kinds >= this are phases: kind - CTM_KIND_PHASE0 indexes ctm_ctx::timers

// Phase time of the engine's stream.  "profile": host wall time with stream syncs on both sides (development).  "gemm_timing":
// a HIP-event pair on the stream, no synchronisation -- the phase's device time, accumulated when the events are drained.
struct PhaseTimer {
    ctm_ctx* c; int id; int e0 = -1; std::chrono::high_resolution_clock::time_point t0;
    PhaseTimer(ctm_ctx* ctx, int i) : c(ctx), id(i) {
        if (c->profile) { (void)hipStreamSynchronize(c->stream); t0 = std::chrono::high_resolution_clock::now(); }
        else if (c->gemm_timing) e0 = timing_begin(c);
    }
    ~PhaseTimer() {
        if (c->profile) {
            (void)hipStreamSynchronize(c->stream);
            c->timers[id] += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        } else if (e0 >= 0) timing_end(c, e0, CTM_KIND_PHASE0 + id, 0.0);
    }
};


// ---- GEMM (gemm_f64.hip) -----------------------------------------------------------------
// C(m,n) = alpha * sum_k A(m,k) B(k,n) + beta * C(m,n),  element addresses
//   A(m,k) = A + segA(m) + k*sak      with segA(m) = (m < splitA ? a0 + m*sam : a1 + (m-splitA)*sam)
//   B(k,n) = B + segB(.) ...          the segmented dim of B is K (splitB_dim=1) or N (splitB_dim=2)
//   C(m,n) = C + segC(m) + n          (row-major, unit column stride)
// batched over `batch` entries of offsets (device array of GemmOff) or regular strides.
struct GemmOff { long long a0, a1, b0, b1, c0, c1; int klen; int pad_; };   // klen > 0: this batch entry contracts only klen values of K

struct GemmDesc {
    int M = 0, N = 0, K = 0;
    const double* A = nullptr; long long sam = 0, sak = 0;
    const double* B = nullptr; long long sbk = 0, sbn = 0;
    double* C = nullptr; long long ldc = 0;
    double alpha = 1.0, beta = 0.0;
    int batch = 1;
    long long strideA = 0, strideB = 0, strideC = 0;   // regular batching (used when offs == nullptr)
    const GemmOff* offs = nullptr;                      // device pointer, optional
    int splitA = 1 << 30, splitB = 1 << 30, splitC = 1 << 30;
    int splitB_dim = 0;                                  // 0 none, 1 = K, 2 = N
    // optional fused column scale of the output: C(m,n) *= colscale[n]
    const double* colscale = nullptr;
    // optional per-batch skip flags (device): batch z is skipped when skip_flags[z] == 0
    const int* skip_flags = nullptr;
    // optional device word: the whole product (and its split-K reduction) is skipped when *skip_all == 0 (generic kernel only)
    const int* skip_all = nullptr;
};
int gemm_f64(ctm_ctx* ctx, const GemmDesc& d);
void gemm_timing_drain(ctm_ctx* ctx);
void gemm_timing_base(ctm_ctx* ctx);
// ---- elementwise / layout kernels (tensor_ops.hip) -----------------------------------------
#define CTM_MAXD 8
int permute_f64(ctm_ctx* ctx, const double* in, double* out, int nd, const long long* dims, const int* perm);
int absmax_f64(ctm_ctx* ctx, const double* x, size_t n, double* d_out);           // d_out: device scalar
int div_by_device_scalar(ctm_ctx* ctx, double* x, size_t n, const double* d_s, int use_abs);
int fill_f64(ctm_ctx* ctx, double* x, size_t n, double v);
int set_identity(ctm_ctx* ctx, double* x, int n, long long ld);
int copy2d(ctm_ctx* ctx, const double* src, long long lds, double* dst, long long ldd, int rows, int cols);
int row_norms(ctm_ctx* ctx, const double* x, int rows, int cols, long long ld, double* d_out);
int row_dots(ctm_ctx* ctx, const double* x, const double* y, int rows, int cols, long long ld, double* d_out);
int gather_rows(ctm_ctx* ctx, const double* src, long long lds, const int* d_idx, int nrows, int cols, double* dst,
                long long ldd, const double* d_rowscale);
int symmetrize_lower(ctm_ctx* ctx, const double* a, double* out, int n, double shift);
int add_transposed01(ctm_ctx* ctx, double* t, int d0, int d2);   // t[i,j,s] = 0.5 (t[i,j,s] + t[j,i,s])
int hermitize_lower_c128(ctm_ctx* ctx, const double* ar, const double* ai, double* outr, double* outi, int n, double shift);
int add_conj_transposed01_c128(ctm_ctx* ctx, double* tr, double* ti, int d0, int d2);   // t = 0.5 (t + conj(t)^T(0,1)), planar
int tril_correction(ctm_ctx* ctx, double* E, int k);             // E -> I - strict_lower(E) - diag(E)/2 with E=G-I
int diag_to_matrix(ctm_ctx* ctx, const double* d, double* out, int n);
int trace_partial(ctm_ctx* ctx, const double* in, double* out, long long n2, int p);  // out[ab] = sum_i in[ab,i,i]
// complex128: the C-ABI carries torch's interleaved (re,im) layout, the engine computes on two planes
int deinterleave_c128(ctm_ctx* ctx, const double* z, double* re, double* im, size_t n);
int interleave_c128(ctm_ctx* ctx, const double* re, const double* im /* nullptr: zero */, double* z, size_t n);
int absmax_c128(ctm_ctx* ctx, const double* re, const double* im, size_t n, double* d_out);   // max |z|
int row_norms_c128(ctm_ctx* ctx, const double* re, const double* im, int rows, int cols, long long ld, double* d_out);
int tril_correction_c128(ctm_ctx* ctx, double* Er, double* Ei, int k);
int norm2_f64(ctm_ctx* ctx, const double* x, size_t n, double* tmp, double* d_out);   // sqrt(sum x^2), deterministic order

// op(X) of a (possibly complex, planar) matrix: t = stored transposed, c = conjugated; im == nullptr for real data
struct XM { const double* re; const double* im; long long ld; bool t; bool c; };
// C (M x N, planar) = op(A) (M x K) op(B) (K x N) [* diag(colscale)] ; one real GEMM, or four for complex operands
int xgemm(ctm_ctx* ctx, int M, int N, int K, const XM& A, const XM& B, double* Cre, double* Cim, long long ldc,
          const double* colscale = nullptr);

// linear operator for the leading-k decomposition: either an explicit n x n matrix M, or the implicit product
// M = R^T Rt with R = opA(cA) opB(cB), Rt = opC(cC) opD(cD) of four n x n enlarged corners (never formed).
struct MatOp {
    int n = 0;
    const double* M = nullptr;
    const double* Mi = nullptr;                                      // imaginary plane (complex128)
    const double* c[4] = {nullptr, nullptr, nullptr, nullptr};
    const double* ci[4] = {nullptr, nullptr, nullptr, nullptr};      // imaginary planes of the corners (complex128)
    bool t[4] = {false, false, false, false};
    int mid[2] = {0, 0};      // inner dimensions of R = opA(cA) opB(cB) (n x mid0 x n) and Rt (n x mid1 x n); 0 means n
    // optional warm start (in/out): k x n row basis (planar for complex128) of the right singular vectors of a nearby
    // operator; rows the caller does not have are zero.  Overwritten with this decomposition's right row factor.
    double* warm = nullptr;
    double* warm_hdr = nullptr;   // optional n-double header row of the warm workspace: [0] = calls left to skip the warm start
    // optional by-products for the projectors of an implicit operator (float64): when the block Krylov solver produced the
    // decomposition it also returns u_i^T R^T and v_i^T Rt^T (k x n each) assembled from the half-way products of its own
    // operator applications, and sets *have_mid -- the caller then needs no further corner passes for P = R conj(U), Pt = Rt V
    double* out_uR = nullptr;
    double* out_vRt = nullptr;
    bool* have_mid = nullptr;
};
// complex128 operators: Ut, Vt are planar (re plane k x n, then im plane), rows = u_k^H, v_k^H
int jacobi_svd_top_op(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt);

// ---- Jacobi SVD / eig (jacobi_core.hip, svd_leading.hip, eigh.hip) -----------------------------------------------------------
// Full one-sided block Jacobi on the rows of M (n x n).  Outputs the k leading triplets:
//   S[k] (descending), Ut (k x n, rows = u_i^T), Vt (k x n, rows = v_i^T).
int jacobi_svd_top(ctm_ctx* ctx, const double* M, int n, int k, double* S, double* Ut, double* Vt);
// Symmetric eigendecomposition (lower triangle of A is referenced), k leading eigenpairs by |lambda|:
//   D[k] (signed), Ut (k x n, rows = eigenvectors).
int jacobi_eigh_top(ctm_ctx* ctx, const double* A, int n, int k, double* D, double* Ut, double* warm = nullptr);
// Hermitian eigendecomposition of a planar complex matrix (lower triangle referenced): D[k] real (signed, by |lambda| descending),
// Ut planar k x n (re plane, then im plane), rows = u_i^H
int jacobi_eigh_top_c(ctm_ctx* ctx, const double* Ar, const double* Ai, int n, int k, double* D, double* Ut, double* warm = nullptr);
// singular values only, small matrices (corner spectra)
int jacobi_svdvals(ctm_ctx* ctx, const double* M, const double* Mi /* nullptr: real */, int n, double* S);
