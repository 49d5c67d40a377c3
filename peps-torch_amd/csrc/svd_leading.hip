// Leading-k decompositions of an explicit matrix or of the implicit operator M = R^T Rt (four enlarged corners, never formed):
// alternating block power iteration with Jacobi Rayleigh-Ritz (svd_iter), block Golub-Kahan-Lanczos with full re-orthogonalisation
// (svd_lanczos), the stationary Rayleigh-Ritz half step (svd_stationary), complex128 twins, and the router jacobi_svd_top_op.
// Split out of jacobi.hip in round 5; the dense block Jacobi they fall back to is jacobi_core.hip.
#include "jacobi_internal.h"

// Y (p x n, ldy) = X (p x n, ldx) * op(Z) for a dense n x n matrix Z (row-major): op = 'N' or 'T'
int rows_times(ctm_ctx* ctx, const double* X, long long ldx, int p, int kin, int nout, const double* Z, bool transZ, double* Y, long long ldy) {
    // Y (p x nout) = X (p x kin) op(Z); Z is stored kin x nout (transZ == false) or nout x kin (transZ == true)
    GemmDesc g; g.M = p; g.N = nout; g.K = kin; g.A = X; g.sam = ldx; g.sak = 1; g.B = Z;
    if (transZ) { g.sbk = 1; g.sbn = kin; } else { g.sbk = nout; g.sbn = 1; }
    g.C = Y; g.ldc = ldy;
    return gemm_f64(ctx, g);
}

// A corner pass shared by a rank group (ctm_set_comm / ctm_set_comm_ops, include/ctm_hip.h): this rank computes the column block
// [r nout / g, (r + 1) nout / g) of Y = X op(Z) -- the same kernels on a sub-block of the big operand: column block of Z, or row block for
// the transposed orientation -- into a contiguous staging buffer, the group all-gathers the blocks (ncclAllGather on this stream, or the
// host callback after draining it) and every rank unpacks the same bits into Y.  g == 1 with a communicator: one part, gathered onto itself
// (the only form the RCCL path can take on a one-GPU box).
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
// (a pass whose block does not fit the staging buffers -- the dense fallback materialises M with n rows -- is computed in full by every
//  rank of the group: the same decision on every rank, the same bits)
static bool pass_is_shared(const ctm_ctx* ctx, int nout, int p) {
    if (!(ctx->comm_nccl_allgather || ctx->comm_host_allgather) || ctx->cplx || nout % (2 * ctx->comm_nranks) != 0) return false;
    const long long cnt = (long long)p * (nout / ctx->comm_nranks);
    return ctx->comm_host_allgather ? cnt <= ctx->comm_cap : p <= 1024;
}
int rows_times_shared(ctm_ctx* ctx, const double* X, long long ldx, int p, int kin, int nout, const double* Z, bool transZ, double* Y, long long ldy) {
    const int g = ctx->comm_nranks, r = ctx->comm_rank, nb = nout / g;
    const long long cnt = (long long)p * nb;
    double *send, *recv;
    if (ctx->comm_host_allgather) {
        if (cnt > ctx->comm_cap) { ctx->set_error("shared corner pass: staging buffers of ctm_set_comm_ops too small"); return CTM_ERR_BADARG; }
        send = ctx->comm_send; recv = ctx->comm_recv;
    } else {
        if (ctx->comm_own_cap < cnt * (1 + g)) {
            if (ctx->comm_own_buf) { CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->comm_own_buf); ctx->comm_own_buf = nullptr; ctx->comm_own_cap = 0; }
            CTM_HIP_CHECK(ctx, hipMalloc((void**)&ctx->comm_own_buf, sizeof(double) * (size_t)cnt * (1 + g)));
            ctx->comm_own_cap = cnt * (1 + g);
        }
        send = ctx->comm_own_buf; recv = ctx->comm_own_buf + cnt;
    }
    GemmDesc gd; gd.M = p; gd.N = nb; gd.K = kin; gd.A = X; gd.sam = ldx; gd.sak = 1;
    if (transZ) { gd.B = Z + (size_t)r * nb * kin; gd.sbk = 1; gd.sbn = kin; }       // rows [r nb, (r+1) nb) of Z (nout x kin) = columns of Z^T
    else { gd.B = Z + (size_t)r * nb; gd.sbk = nout; gd.sbn = 1; }                   // columns [r nb, (r+1) nb) of Z (kin x nout)
    gd.C = send; gd.ldc = nb;
    CTM_TRY(gemm_f64(ctx, gd));
    if (ctx->comm_host_allgather) {
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->comm_host_allgather(ctx->comm_user, cnt) != 0) { ctx->set_error("shared corner pass: the all-gather callback failed"); return CTM_ERR_HIP; }
    } else {
        const int st = reinterpret_cast<nccl_allgather_t>(ctx->comm_nccl_allgather)(send, recv, (size_t)cnt, 8 /* ncclFloat64 */, ctx->comm, ctx->stream);
        if (st != 0) { ctx->set_error("shared corner pass: ncclAllGather returned " + std::to_string(st)); return CTM_ERR_HIP; }
    }
    for (int q = 0; q < g; ++q) CTM_TRY(copy2d(ctx, recv + (size_t)q * cnt, nb, Y + (size_t)q * nb, ldy, p, nb));
    ctx->comm_calls += 1; ctx->comm_doubles += (double)cnt * g;
    return CTM_OK;
}

// C = B * M (transpose == false) or B * M^T (transpose == true) for the operator M of `op`
// mid (optional, p x n, leading dimension n; implicit operators only): receives the half-way product B R^T (transpose == false)
// or B Rt^T (transpose == true)
int matop_apply(ctm_ctx* ctx, const MatOp& op, bool transpose, const double* B, long long ldb, int p, double* C, long long ldc,
                double* mid) {
    const int n = op.n;
    if (op.M) return rows_times(ctx, B, ldb, p, n, n, op.M, transpose, C, ldc);
    // implicit M = R^T Rt,  R = opA(cA) opB(cB) (n x m0 x n),  Rt = opC(cC) opD(cD) (n x m1 x n)
    const int m0 = op.mid[0] ? op.mid[0] : n, m1 = op.mid[1] ? op.mid[1] : n, mw = std::max(n, std::max(m0, m1));
    ArenaScope scope(ctx);
    double *t1, *t2 = mid;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * mw, (void**)&t1));
    if (!t2) CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * mw, (void**)&t2));
    // (a rank group that shares this unit splits every pass by output columns: rows_times_shared)
    auto pass = [&](const double* X, long long ldx, int kin, int nout, const double* Z, bool tZ, double* Yo, long long ldy) -> int {
        if (pass_is_shared(ctx, nout, p)) return rows_times_shared(ctx, X, ldx, p, kin, nout, Z, tZ, Yo, ldy);
        return rows_times(ctx, X, ldx, p, kin, nout, Z, tZ, Yo, ldy);
    };
    if (!transpose) {   // B R^T Rt = ((B opB(cB)^T) opA(cA)^T) opC(cC) opD(cD)
        CTM_TRY(pass(B, ldb, n, m0, op.c[1], !op.t[1], t1, m0));
        CTM_TRY(pass(t1, m0, m0, n, op.c[0], !op.t[0], t2, n));
        CTM_TRY(pass(t2, n, n, m1, op.c[2], op.t[2], t1, m1));
        return pass(t1, m1, m1, n, op.c[3], op.t[3], C, ldc);
    }
    // B Rt^T R = ((B opD(cD)^T) opC(cC)^T) opA(cA) opB(cB)
    CTM_TRY(pass(B, ldb, n, m1, op.c[3], !op.t[3], t1, m1));
    CTM_TRY(pass(t1, m1, m1, n, op.c[2], !op.t[2], t2, n));
    CTM_TRY(pass(t2, n, n, m0, op.c[0], op.t[0], t1, m0));
    return pass(t1, m0, m0, n, op.c[1], op.t[1], C, ldc);
}

// Calls of a unit that start cold after its full-block warm probe (two half steps on k + k/2 rows) was handed to the Krylov solver
// with relative residual r.  A probe is only kept below r = 1e-9; the environment of a converging run contracts by a factor
// of a few per sweep and a unit is visited twice per sweep, so the next probe is scheduled for when it could succeed.
// memory of a unit between sweeps (header row of its warm workspace, doubles): [0] calls left that skip the warm probe (svd_iter),
// [1] block steps of the last accepted Krylov solve, [2] its residual estimate / s_0, [3] consecutive warm probes that were handed
// to the Krylov solver (each one quadruples the distance to the next probe: a full-rank environment at its rounding floor, where the
// previous basis stays ~1e-10 away from the new operator for ever, otherwise pays two half steps on k + k/2 rows every few sweeps)
enum { HDR_SKIP = 0, HDR_STEPS = 1, HDR_EST = 2, HDR_FAILS = 3, HDR_BLOCK = 4,      // (HDR_BLOCK: block size the remembered step count belongs to)
       // stationary fast path (svd_stationary, option "warm_accept_tol"): [5] how far the normalised singular values moved between the last two
       // solves (spectrum_movement; a lower bound on the movement of the operator), 0 = unknown; [6] which side the workspace rows hold
       // (0: right vectors, 1: left vectors); [7] accepted Rayleigh-Ritz calls since the last full solve; [8] calls left that do not try
       // the fast path after a rejection; [9] consecutive rejections
       HDR_DIST = 5, HDR_SIDE = 6, HDR_RUN = 7, HDR_SSKIP = 8, HDR_SFAILS = 9,
       // [10] doubles the workspace holds BEHIND the header row (written by whoever allocated it; 0 = none): the Ritz region of the block Krylov
       // solvers -- [0] rows m of the stored rotation matrix, [1] block size of the recurrence it belongs to, [16 ...] the m x m accumulated
       // rotations of the unit's last Ritz extraction (svd_full: rot), the start of the next one (include/ctm_hip.h: warm-start workspace)
       HDR_RITZ_CAP = 10,
       // [11] 1: the last accepted block Krylov solve of the unit needed a third Cholesky-QR pass somewhere (or was repeated on the synchronous path)
       HDR_THIRD = 11, HDR_WORDS = 12,
       HDR_SPREV = 16 /* from here: the k singular values of the previous solve (spectrum_movement) */ };

inline int warm_skip_calls(const ctm_ctx* ctx, double r) {
    const int need = (int)std::ceil(2.0 * std::log(std::max(r, 1e-9) / 1e-9) / std::log(5.0)) - 1;
    return std::max(1, std::min(ctx->si_warm_skip_calls, need));
}

// ---------------------------------------------------------------------------------------------
// leading-k decomposition by alternating block power iteration with Jacobi Rayleigh-Ritz:
//   U M = C  ->  rows of C orthogonalised (same rotations applied to U)  ->  V = rows/|rows|, s = |rows|
//   V M^T = C' ->  ...                                                    ->  U = rows/|rows|
// Each half step is ONE big GEMM (p x n x n, FP64 MFMA) plus a row-Jacobi on p = k + oversampling rows (a few
// rounds, rows already nearly orthogonal after the first steps).  One relation (e.g. U M = S V^T) holds exactly by
// construction; the iteration stops when the other one's residual |V_i M^T - s_i U_i| <= tol * s_0 for all i < k.
// If that does not happen within max_iter half steps the caller falls back to svd_full (same answer, O(n^3)).
// `sym`: M is symmetric (eigenproblem) -- identical iteration, M^T = M.
// ---------------------------------------------------------------------------------------------
int svd_iter(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged, bool* want_krylov) {
    *converged = false;
    if (want_krylov) *want_krylov = false;
    const int n = op.n;
    const int b = 32;
    int p_full = k + std::max(32, k / 2);
    p_full = ((p_full + 2 * b - 1) / (2 * b)) * (2 * b);       // even number of blocks
    if (p_full >= n / 2) return CTM_OK;                         // not worth it: caller uses the full path
    // Rank-adaptive block: start with 64 vectors and double while the spectrum is not exhausted inside the block.
    // Environments of weakly entangled / random states are numerically low rank (tens of singular values above
    // eps * s_0 out of thousands): then the whole decomposition costs a few 64-row power steps.
    int p = std::min(64, p_full);
    ArenaScope scope(ctx);
    const long long ld = 2LL * n;
    double *XA, *XB, *norms, *inv, *res, *sprev;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p_full * ld, (void**)&XA));     // [C | companion]
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p_full * ld, (void**)&XB));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p_full, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p_full, (void**)&inv));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p_full, (void**)&res));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p_full, (void**)&sprev));
    std::vector<double> h(p_full, 0.0), hr(p_full, 0.0);
    // start: the caller's warm basis (orthonormal rows of a previous decomposition of a nearby operator; rows it does not
    // have are zero) completed by pseudo-random rows projected onto its orthogonal complement -- or, cold, a
    // pseudo-random basis (need not be orthonormal)
    int kw = 0;
    double hdr_fails = 0.0;
    // a failed full-block warm probe: remember it, and place the next one further away each time
    auto probe_failed = [&](double r) -> int {
        if (!op.warm_hdr) return CTM_OK;
        const double f = std::min(hdr_fails + 1.0, 8.0);
        const int skip = std::max(warm_skip_calls(ctx, r), std::min(4096, ctx->si_warm_skip_calls << (2 * (int)(f - 1.0))));   // x4 per failure
        CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_SKIP, 1, (double)skip));
        return fill_f64(ctx, op.warm_hdr + HDR_FAILS, 1, f);
    };
    if (op.warm) {
        double hdr = 0.0;
        CTM_TRY(row_norms(ctx, op.warm, k, n, n, norms));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * std::min(k, p_full), hipMemcpyDeviceToHost, ctx->stream));
        if (op.warm_hdr) {
            double hw[HDR_WORDS];
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(hw, op.warm_hdr, sizeof(hw), hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            hdr = hw[HDR_SKIP]; hdr_fails = hw[HDR_FAILS];
        }
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        while (kw < std::min(k, p_full) && std::fabs(h[kw] - 1.0) < 1e-6) ++kw;
        std::fill(h.begin(), h.end(), 0.0);
        // the last warm start of this workspace was hopeless (residual O(s0): the gauge of the environment legs keeps
        // changing between sweeps): do not pay for a full-block probe again for a few calls
        if (hdr >= 1.0 && want_krylov) { kw = 0; CTM_TRY(fill_f64(ctx, op.warm_hdr, 1, hdr - 1.0)); ctx->si_warm_skips += 1; }
    }
    if (kw > 0) {
        // a FULL warm basis (the previous decomposition had at least k significant triplets) starts with the full block, so a
        // nearly converged basis is recognised by the first residual checks instead of triggering the block-growth logic
        p = (kw >= k) ? p_full : std::min(p_full, std::max(64, ((kw + 16 + 63) / 64) * 64));
        // a small numerical rank fits a 32-row block (two 16-row panels, 32 x 32 pair Gram): the strip GEMMs of such a block
        // are HBM-bound instead of MFMA-bound (0.43 vs 0.64 ms per corner pass at n = 16384)
        if (ctx->si_block32 && kw + 8 <= 32 && kw < k && p_full >= 64) p = 32;
        kw = std::min(kw, p - 8);
        CTM_TRY(copy2d(ctx, op.warm, n, XB, ld, kw, n));
        double* Rn = XB + (size_t)kw * ld;
        const int pr = p - kw;
        CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, Rn, pr, n, ld, 0x1234567ULL);
        ArenaScope ws(ctx);
        double* Gw;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)pr * kw, (void**)&Gw));
        GemmDesc g1; g1.M = pr; g1.N = kw; g1.K = n; g1.A = Rn; g1.sam = ld; g1.sak = 1; g1.B = XB; g1.sbk = 1; g1.sbn = ld; g1.C = Gw; g1.ldc = kw;
        CTM_TRY(gemm_f64(ctx, g1));                                  // G = R V^T
        GemmDesc g2; g2.M = pr; g2.N = n; g2.K = kw; g2.A = Gw; g2.sam = kw; g2.sak = 1; g2.B = XB; g2.sbk = ld; g2.sbn = 1; g2.C = Rn; g2.ldc = ld;
        g2.alpha = -1.0; g2.beta = 1.0;
        CTM_TRY(gemm_f64(ctx, g2));                                  // R -= G V
    } else
        CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, XB, p, n, ld, 0x1234567ULL);
    const bool warm = kw > 0;
    double* cur = XB;          // columns [0,n) of `cur` hold the current basis B (p x n)
    double* nxt = XA;
    bool have_prev = false;
    int side = 0;              // 0: C = B M^T (B = right basis V, produces U) ; 1: C = B M (B = U, produces V)
    double s0 = 0.0;
    int rank = 0, kk = k;
    double worst_prev = 0.0;
    std::vector<double> worst_hist;       // residual per checked half step (stagnation: see below)
    const double rank_tol = ctx->rank_tol;      // singular values below rank_tol * s_0 are rounding noise: never required to converge, returned as zeros
    const int max_half = 2 * ctx->si_max_iter;
    int it = 0;
    for (; it < max_half; ++it) {
        // C = B op(M) -> nxt[:, 0:n] ; companion nxt[:, n:2n] = B
        CTM_TRY(matop_apply(ctx, op, side == 0, cur, ld, p, nxt, ld));
        CTM_TRY(copy2d(ctx, cur, ld, nxt + n, ld, p, n));
        if (have_prev) {
            // residual of the relation that is NOT exact by construction: |C_i - s_i A_i| with A = previous normalised rows,
            // which sit in cur[:, n:2n] (companion of the previous half step, rotated along)
            CTM_LAUNCH(ctx, resid_rows_kernel, dim3((p + 3) / 4), dim3(256), 0, nxt, ld, cur + n, ld, sprev, p, n, res);
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(hr.data(), res, sizeof(double) * p, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            std::vector<int> idx(p); std::iota(idx.begin(), idx.end(), 0);
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return h[a] > h[c]; });
            rank = 0;
            for (int i = 0; i < p; ++i) rank += (h[i] > rank_tol * s0);
            const bool exhausted = rank <= p - 8;            // the block holds every singular value above eps * s_0
            if (!exhausted && p < p_full) {
                // the spectrum does not collapse inside the block: a large problem goes to the block Krylov solver
                if (want_krylov && ctx->lz_enable && k >= ctx->lz_min_k) { *want_krylov = true; ctx->si_last_iters = it; ctx->si_total_iters += it; return CTM_OK; }
                // grow the block: fresh pseudo-random rows appended to the current basis
                const int pn = std::min(p_full, 2 * p);
                CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, cur + (size_t)p * ld, pn - p, n, ld,
                                   0x9876543ULL + (unsigned long long)pn);
                if (ctx->jacobi_verbose) fprintf(stderr, "[si] n=%d grow block %d -> %d (rank so far %d)\n", n, p, pn, rank);
                p = pn; have_prev = false;
                continue;
            }
            kk = exhausted ? std::min(k, rank) : k;
            double worst = 0.0;
            for (int i = 0; i < kk; ++i) worst = std::max(worst, hr[idx[i]]);
            if (ctx->jacobi_verbose) fprintf(stderr, "[si] n=%d p=%d half-step %d  rank=%d  max resid/s0 = %.3e\n", n, p, it, rank, worst / std::max(s0, 1e-300));
            // The residual test certifies that the sorted Ritz triplets ARE singular triplets; that they are the LARGEST ones rests on
            // the guard rows of the block.  When the block exhausts the numerical rank, any direction the start did not contain shows up
            // as one more Ritz value above the noise floor after a single application (and must then converge too).  A full block from
            // a warm start carries guard rows that have seen the operator once: a new direction of size sigma_k .. sqrt(n) sigma_k would
            // still hide among them, so such a start is not accepted before the guard rows have had three half steps.
            const bool sound = !warm || exhausted || it >= 3;
            if (sound && worst <= resid_tol(ctx, n) * s0) { *converged = true; break; }
            // a block whose residual contracts slowly (slowly decaying tail): predict the remaining half steps from the
            // observed contraction and hand over to the block Krylov solver when many are left
            if (want_krylov && ctx->lz_enable && k >= ctx->lz_min_k && !exhausted && warm && it == 1 && worst > 1e-9 * s0) {
                // a warm basis that is not close (environment still changing) on a full-rank problem: block Krylov straight away,
                // and the next calls of this unit do not pay for the full-block probe again (see warm_skip_calls())
                CTM_TRY(probe_failed(worst / s0));
                *want_krylov = true; ctx->si_last_iters = it; ctx->si_total_iters += it; return CTM_OK;
            }
            if (want_krylov && ctx->lz_enable && k >= ctx->lz_min_k && !exhausted && worst_prev > 0.0 && worst < worst_prev) {
                const double rate = worst / worst_prev, need = std::log(resid_tol(ctx, n) * s0 / worst) / std::log(rate);
                if (need > ctx->lz_switch_steps) {
                    if (warm && p == p_full) CTM_TRY(probe_failed(worst / s0));
                    *want_krylov = true; ctx->si_last_iters = it; ctx->si_total_iters += it; return CTM_OK;
                }
            } else if (want_krylov && ctx->lz_enable && k >= ctx->lz_min_k && !exhausted && worst_prev > 0.0 && it >= 6) {
                if (warm && p == p_full) CTM_TRY(probe_failed(worst / s0));
                *want_krylov = true; ctx->si_last_iters = it; ctx->si_total_iters += it; return CTM_OK;      // not contracting at all
            }
            worst_prev = worst;
            // Stagnation at the rounding floor of the Rayleigh-Ritz (a flat leading spectrum leaves ~50 x the Jacobi tolerance,
            // above the acceptance threshold): more half steps cannot help, the caller's dense path takes over now rather than
            // after si_max_iter iterations.
            worst_hist.push_back(worst);
            const size_t nh = worst_hist.size();
            if (nh >= 10 && worst < 1e-10 * s0 && worst > 0.5 * worst_hist[nh - 7]) break;
        }
        // Rayleigh-Ritz to convergence (the bases must be orthonormal for the residual test to certify the triplets);
        // the very first one only orthonormalises a power step of the random start, so it is capped
        int st;
        const double fro = host_fro(ctx, nxt, p, n, ld, norms, h, &st);
        CTM_TRY(st);
        // (no verification sweep once a sweep found <= si_quad_exit: the residual test below certifies the triplets of the step that is
        // accepted, and a sweep that found 1e-9 leaves ~1e-18 / gap -- see jacobi_rows)
        ctx->jacobi_quad_exit = ctx->si_quad_exit;
        const int st_rr = jacobi_rows(ctx, nxt, p, ld, n, (int)ld, p == 32 ? 16 : b, std::min(k, p - 1), fro, (have_prev || warm) ? ctx->si_rr_sweeps : std::min(3, ctx->si_rr_sweeps),
                            false, ctx->si_tau_both != 0);
        ctx->jacobi_quad_exit = 0.0;
        CTM_TRY(st_rr);
        CTM_TRY(row_norms(ctx, nxt, p, n, ld, norms));
        h.assign(p_full, 0.0);
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * p, hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(sprev, norms, sizeof(double) * p, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        s0 = *std::max_element(h.begin(), h.begin() + p);
        CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3((p + 255) / 256), dim3(256), 0, norms, inv, p);
        CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, nxt, p, n, ld, inv);
        // now: nxt[:, 0:n] = new orthonormal basis A (left if side==0), nxt[:, n:2n] = rotated B, s = h
        have_prev = true;
        std::swap(cur, nxt);
        side ^= 1;
    }
    ctx->si_last_iters = it; ctx->si_total_iters += it;
    if (!*converged) return CTM_OK;
    // At the break: `cur` holds [B | A-rotated]: B = cur[:, 0:n] (normalised rows from the last RR), A = cur[:, n:2n],
    // s = h (host).  side==1 -> B = U, A = V ; side==0 -> B = V, A = U.  Triplets beyond the numerical rank (or beyond the
    // block) are returned as exact zeros: they are below eps * s_0 and every consumer masks them.
    std::vector<int> idx(p); std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return h[a] > h[c]; });
    const int kv = std::min(k, std::min(p, std::max(kk, 1)));           // verified triplets
    std::vector<double> hs(k, 0.0);
    for (int i = 0; i < kv; ++i) hs[i] = h[idx[i]];
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(int) * k, (void**)&d_idx));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, idx.data(), sizeof(int) * kv, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, hs.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    CTM_TRY(fill_f64(ctx, Ut, (size_t)k * n, 0.0));
    CTM_TRY(fill_f64(ctx, Vt, (size_t)k * n, 0.0));
    const double* Bp = cur; const double* Ap = cur + n;
    CTM_TRY(gather_rows(ctx, side == 1 ? Bp : Ap, ld, d_idx, kv, n, Ut, n, nullptr));
    CTM_TRY(gather_rows(ctx, side == 1 ? Ap : Bp, ld, d_idx, kv, n, Vt, n, nullptr));
    CTM_TRY(reorth_rows(ctx, Ut, kv, n, n, 1));
    CTM_TRY(reorth_rows(ctx, Vt, kv, n, n, 1));
    ctx->si_last_rank = rank;
    ctx->si_warm_starts += warm ? 1 : 0;
    if (warm && op.warm_hdr && hdr_fails > 0.0) CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_FAILS, 1, 0.0));
    return CTM_OK;
}

// Y (R real panel rows x n) = X * op(Z),  Z planar n x n complex, op in {N, T, C = conj, H = conj transpose};
// scratch: R x n doubles for i*X.   (x + iy)(zr + i zi):  Y = X op(Zr) +- (iX) op(Zi)
int rows_times_c(ctm_ctx* ctx, const double* X, long long ldx, int R, int kin, int nout, const double* Zr, const double* Zi, bool trans, bool conj,
                 double* Y, long long ldy, double* scratch) {
    // Z stored kin x nout (trans == false) or nout x kin (trans == true)
    const size_t tot = (size_t)R * kin;
    CTM_LAUNCH(ctx, panel_times_i_kernel, dim3((int)std::min<size_t>((tot + 255) / 256, 2048)), dim3(256), 0, X, ldx, scratch,
                       (long long)kin, R, kin);
    GemmDesc g; g.M = R; g.N = nout; g.K = kin; g.A = X; g.sam = ldx; g.sak = 1; g.B = Zr;
    if (trans) { g.sbk = 1; g.sbn = kin; } else { g.sbk = nout; g.sbn = 1; }
    g.C = Y; g.ldc = ldy;
    CTM_TRY(gemm_f64(ctx, g));
    g.A = scratch; g.sam = kin; g.B = Zi; g.alpha = conj ? -1.0 : 1.0; g.beta = 1.0;
    return gemm_f64(ctx, g);
}

// C = B * M (adjoint == false) or B * M^H (adjoint == true), B and C in panel layout
int matop_apply_c(ctm_ctx* ctx, const MatOp& op, bool adjoint, const double* B, long long ldb, int R, double* C, long long ldc) {
    const int n = op.n;
    const int m0 = op.mid[0] ? op.mid[0] : n, m1 = op.mid[1] ? op.mid[1] : n, mw = std::max(n, std::max(m0, m1));
    ArenaScope scope(ctx);
    double *t1, *t2, *sc;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)R * mw, (void**)&sc));
    if (op.M) return rows_times_c(ctx, B, ldb, R, n, n, op.M, op.Mi, adjoint, adjoint, C, ldc, sc);
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)R * mw, (void**)&t1));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)R * mw, (void**)&t2));
    if (!adjoint) {   // B R^T Rt = ((B opB(cB)^T) opA(cA)^T) opC(cC) opD(cD)       (plain transposes, ctm_projectors.py:263)
        CTM_TRY(rows_times_c(ctx, B, ldb, R, n, m0, op.c[1], op.ci[1], !op.t[1], false, t1, m0, sc));
        CTM_TRY(rows_times_c(ctx, t1, m0, R, m0, n, op.c[0], op.ci[0], !op.t[0], false, t2, n, sc));
        CTM_TRY(rows_times_c(ctx, t2, n, R, n, m1, op.c[2], op.ci[2], op.t[2], false, t1, m1, sc));
        return rows_times_c(ctx, t1, m1, R, m1, n, op.c[3], op.ci[3], op.t[3], false, C, ldc, sc);
    }
    // B M^H = B opD(cD)^H opC(cC)^H conj(opA(cA)) conj(opB(cB))
    CTM_TRY(rows_times_c(ctx, B, ldb, R, n, m1, op.c[3], op.ci[3], !op.t[3], true, t1, m1, sc));
    CTM_TRY(rows_times_c(ctx, t1, m1, R, m1, n, op.c[2], op.ci[2], !op.t[2], true, t2, n, sc));
    CTM_TRY(rows_times_c(ctx, t2, n, R, n, m0, op.c[0], op.ci[0], op.t[0], true, t1, m0, sc));
    return rows_times_c(ctx, t1, m0, R, m0, n, op.c[1], op.ci[1], op.t[1], true, C, ldc, sc);
}

// leading-k triplets of a complex operator: the iteration of svd_iter() on panel rows
int svd_iter_c(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged, bool* want_krylov) {
    *converged = false;
    if (want_krylov) *want_krylov = false;
    const int n = op.n;
    int p_full = k + std::max(32, k / 2);
    p_full = ((p_full + 63) / 64) * 64;
    if (p_full >= n / 2) return CTM_OK;
    int p = std::min(64, p_full);                 // complex rows; 2p real rows
    ArenaScope scope(ctx);
    const long long ld = 2LL * n;
    double *XA, *XB, *norms, *nc, *inv, *res;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)2 * p_full * ld, (void**)&XA));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)2 * p_full * ld, (void**)&XB));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * p_full, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * p_full, (void**)&nc));      // complex row norm, replicated on both real rows
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * p_full, (void**)&inv));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * p_full, (void**)&res));
    std::vector<double> h(p_full, 0.0), hr(p_full, 0.0), tmp(2 * p_full);
    // warm start: see svd_iter(); the caller's basis is planar (k x n re plane, then im plane)
    int kw = 0;
    const size_t wkn = (size_t)k * n;
    if (op.warm) {
        double hdr = 0.0;
        CTM_TRY(row_norms_c128(ctx, op.warm, op.warm + wkn, std::min(k, p_full), n, n, norms));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * std::min(k, p_full), hipMemcpyDeviceToHost, ctx->stream));
        if (op.warm_hdr) CTM_HIP_CHECK(ctx, hipMemcpyAsync(&hdr, op.warm_hdr, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        while (kw < std::min(k, p_full) && std::fabs(h[kw] - 1.0) < 1e-6) ++kw;
        std::fill(h.begin(), h.end(), 0.0);
        if (hdr >= 1.0 && want_krylov) { kw = 0; CTM_TRY(fill_f64(ctx, op.warm_hdr, 1, hdr - 1.0)); ctx->si_warm_skips += 1; }   // see svd_iter()
    }
    if (kw > 0) {
        p = (kw >= k) ? p_full : std::min(p_full, std::max(64, ((kw + 16 + 63) / 64) * 64));
        kw = std::min(kw, p - 8);
        const int pr = p - kw;
        ArenaScope ws(ctx);
        double *Rn, *Gw, *Tw;
        const size_t rn = (size_t)pr * n;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * rn, (void**)&Rn));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)pr * kw, (void**)&Gw));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * rn, (void**)&Tw));
        CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, Rn, 2 * pr, n, (long long)n, 0x1234567ULL);
        XM r{Rn, Rn + rn, n, false, false}, vh{op.warm, op.warm + wkn, n, true, true}, v{op.warm, op.warm + wkn, n, false, false};
        CTM_TRY(xgemm(ctx, pr, kw, n, r, vh, Gw, Gw + (size_t)pr * kw, kw));            // G = R V^H
        XM g{Gw, Gw + (size_t)pr * kw, kw, false, false};
        CTM_TRY(xgemm(ctx, pr, n, kw, g, v, Tw, Tw + rn, n));                             // T = G V
        CTM_LAUNCH(ctx, sub_inplace_kernel, dim3(2048), dim3(256), 0, Rn, Tw, 2 * rn);
        CTM_LAUNCH(ctx, planar_to_panel_kernel, dim3(2048), dim3(256), 0, op.warm, op.warm + wkn, (long long)n, kw, n, XB, ld, 0);
        CTM_LAUNCH(ctx, planar_to_panel_kernel, dim3(2048), dim3(256), 0, Rn, Rn + rn, (long long)n, pr, n, XB, ld, kw);
    } else
        CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, XB, 2 * p, n, ld, 0x1234567ULL);
    const bool warm = kw > 0;
    double* cur = XB; double* nxt = XA;
    bool have_prev = false;
    int side = 0;              // 0: C = B M^H (B = rows v^H, produces s u^H) ; 1: C = B M (B = rows u^H, produces s v^H)
    double s0 = 0.0;
    int rank = 0, kk = k;
    double worst_prev = 0.0;
    std::vector<double> worst_hist;       // residual per checked half step (stagnation: see below)
    const double rank_tol = ctx->rank_tol;      // singular values below rank_tol * s_0 are rounding noise: never required to converge, returned as zeros
    const int max_half = 2 * ctx->si_max_iter;
    int it = 0;
    for (; it < max_half; ++it) {
        const int R = 2 * p;
        CTM_TRY(matop_apply_c(ctx, op, side == 0, cur, ld, R, nxt, ld));
        CTM_TRY(copy2d(ctx, cur, ld, nxt + n, ld, R, n));
        if (have_prev) {
            CTM_LAUNCH(ctx, resid_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, nxt, ld, cur + n, ld, nc, R, n, res);
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(tmp.data(), res, sizeof(double) * R, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            for (int cr = 0; cr < p; ++cr) { const int rr = crow_re(cr); hr[cr] = std::sqrt(tmp[rr] * tmp[rr] + tmp[rr + BC] * tmp[rr + BC]); }
            std::vector<int> idx(p); std::iota(idx.begin(), idx.end(), 0);
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return h[a] > h[c]; });
            rank = 0;
            for (int i = 0; i < p; ++i) rank += (h[i] > rank_tol * s0);
            const bool exhausted = rank <= p - 8;
            if (!exhausted && p < p_full) {
                if (want_krylov && ctx->lz_enable && k >= ctx->lz_min_k) { *want_krylov = true; ctx->si_last_iters = it; ctx->si_total_iters += it; return CTM_OK; }
                const int pn = std::min(p_full, 2 * p);
                CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, cur + (size_t)2 * p * ld, 2 * (pn - p), n, ld,
                                   0x9876543ULL + (unsigned long long)pn);
                if (ctx->jacobi_verbose) fprintf(stderr, "[si-c] n=%d grow block %d -> %d (rank so far %d)\n", n, p, pn, rank);
                p = pn; have_prev = false;
                continue;
            }
            kk = exhausted ? std::min(k, rank) : k;
            double worst = 0.0;
            for (int i = 0; i < kk; ++i) worst = std::max(worst, hr[idx[i]]);
            if (ctx->jacobi_verbose) fprintf(stderr, "[si-c] n=%d p=%d half-step %d  rank=%d  max resid/s0 = %.3e\n", n, p, it, rank, worst / std::max(s0, 1e-300));
            const bool sound = !warm || exhausted || it >= 3;        // see svd_iter()
            if (sound && worst <= resid_tol(ctx, n) * s0) { *converged = true; break; }
            if (want_krylov && ctx->lz_enable && k >= ctx->lz_min_k && !exhausted) {        // same hand-over rules as svd_iter()
                bool sw = warm && it == 1 && worst > 1e-9 * s0;
                if (sw && op.warm_hdr) CTM_TRY(fill_f64(ctx, op.warm_hdr, 1, (double)warm_skip_calls(ctx, worst / s0)));
                if (!sw && worst_prev > 0.0 && worst < worst_prev)
                    sw = std::log(resid_tol(ctx, n) * s0 / worst) / std::log(worst / worst_prev) > ctx->lz_switch_steps;
                else if (!sw && worst_prev > 0.0 && it >= 6) sw = true;
                if (sw) { *want_krylov = true; ctx->si_last_iters = it; ctx->si_total_iters += it; return CTM_OK; }
            }
            worst_prev = worst;
            // Stagnation at the rounding floor of the Rayleigh-Ritz (a flat leading spectrum leaves ~50 x the Jacobi tolerance,
            // above the acceptance threshold): more half steps cannot help, the caller's dense path takes over now rather than
            // after si_max_iter iterations.
            worst_hist.push_back(worst);
            const size_t nh = worst_hist.size();
            if (nh >= 10 && worst < 1e-10 * s0 && worst > 0.5 * worst_hist[nh - 7]) break;
        }
        int st;
        std::vector<double> hh;
        const double fro = host_fro(ctx, nxt, R, n, ld, norms, hh, &st);
        CTM_TRY(st);
        ctx->jacobi_quad_exit = ctx->si_quad_exit;
        const int st_rr = jacobi_rows(ctx, nxt, R, ld, n, (int)ld, 2 * BC, std::min(k, p - 1), fro, (have_prev || warm) ? ctx->si_rr_sweeps : std::min(3, ctx->si_rr_sweeps), true,
                            ctx->si_tau_both != 0);
        ctx->jacobi_quad_exit = 0.0;
        CTM_TRY(st_rr);
        CTM_TRY(row_norms(ctx, nxt, R, n, ld, norms));
        CTM_LAUNCH(ctx, panel_combine_kernel, dim3((R + 255) / 256), dim3(256), 0, norms, nc, R);
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(tmp.data(), nc, sizeof(double) * R, hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        h.assign(p_full, 0.0);
        for (int cr = 0; cr < p; ++cr) h[cr] = tmp[crow_re(cr)];
        s0 = *std::max_element(h.begin(), h.begin() + p);
        CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3((R + 255) / 256), dim3(256), 0, nc, inv, R);
        CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, nxt, R, n, ld, inv);
        have_prev = true;
        std::swap(cur, nxt);
        side ^= 1;
    }
    ctx->si_last_iters = it; ctx->si_total_iters += it;
    if (!*converged) return CTM_OK;
    std::vector<int> idx(p); std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return h[a] > h[c]; });
    const int kv = std::min(k, std::min(p, std::max(kk, 1)));
    std::vector<double> hs(k, 0.0);
    for (int i = 0; i < kv; ++i) hs[i] = h[idx[i]];
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(int) * 2 * k, (void**)&d_idx));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, hs.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t kn = (size_t)k * n;
    CTM_TRY(fill_f64(ctx, Ut, 2 * kn, 0.0));
    CTM_TRY(fill_f64(ctx, Vt, 2 * kn, 0.0));
    // verified triplets first into compact planar (kv x n) buffers, re-orthonormalised, then placed into the k-row outputs
    double *Uc, *Vc;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)kv * n, (void**)&Uc));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)kv * n, (void**)&Vc));
    const double* Bp = cur; const double* Ap = cur + n;
    CTM_TRY(panel_gather(ctx, side == 1 ? Bp : Ap, ld, idx, kv, n, Uc, d_idx));
    CTM_TRY(panel_gather(ctx, side == 1 ? Ap : Bp, ld, idx, kv, n, Vc, d_idx));
    CTM_TRY(reorth_rows_c(ctx, Uc, kv, n, 1));
    CTM_TRY(reorth_rows_c(ctx, Vc, kv, n, 1));
    const size_t kvn = (size_t)kv * n;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(Ut, Uc, sizeof(double) * kvn, hipMemcpyDeviceToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(Ut + kn, Uc + kvn, sizeof(double) * kvn, hipMemcpyDeviceToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vt, Vc, sizeof(double) * kvn, hipMemcpyDeviceToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vt + kn, Vc + kvn, sizeof(double) * kvn, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->si_last_rank = rank;
    ctx->si_warm_starts += warm ? 1 : 0;
    return CTM_OK;
}

// rows of W (64 x n) -> orthonormal rows spanning the same space: unit-norm scaling + two Cholesky-QR passes
// (W <- L^-1 W with W W^T = L L^T); falls back to the row-Jacobi when a pivot signals near dependence.
int orthonormalise_block(ctm_ctx* ctx, double* W, int rows, int n, double* norms, double* inv, double* min_norm, double* max_norm) {
    std::vector<double> h(rows);
    CTM_TRY(row_norms(ctx, W, rows, n, n, norms));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *min_norm = *std::min_element(h.begin(), h.end());
    *max_norm = *std::max_element(h.begin(), h.end());
    CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3((rows + 255) / 256), dim3(256), 0, norms, inv, rows);
    CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, W, rows, n, (long long)n, inv);
    bool ok = (rows == 64 || rows == 32) && (*min_norm > 0.0);
    if (ok) {
        ArenaScope scope(ctx);
        double *G, *Li;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 64 * 64, (void**)&G));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 64 * 64, (void**)&Li));
        double* status = ctx->d_scratch + 16;
        for (int pass = 0; pass < 2 && ok; ++pass) {
            GemmDesc g; g.M = rows; g.N = rows; g.K = n; g.A = W; g.sam = n; g.sak = 1; g.B = W; g.sbk = 1; g.sbn = n; g.C = G; g.ldc = rows;
            CTM_TRY(gemm_f64(ctx, g));
            CTM_LAUNCH(ctx, chol64_inv_kernel, dim3(1), dim3(64), 0, (const double*)G, Li, status, rows);
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scratch + 16, status, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (!(ctx->h_scratch[16] > (pass == 0 ? 1e-10 : 0.5))) { ok = false; break; }      // unit rows: pivots in (0, 1]
            GemmDesc a; a.M = rows; a.N = n; a.K = rows; a.A = Li; a.sam = rows; a.sak = 1; a.B = W; a.sbk = n; a.sbn = 1; a.C = W; a.ldc = n;
            CTM_TRY(gemm_f64(ctx, a));                       // in place: a workgroup reads its whole column strip before it writes
        }
    }
    if (ok) return CTM_OK;
    // near-dependent rows: one-sided Jacobi (rank revealing), then unit norms
    int st;
    const double fro = host_fro(ctx, W, rows, n, n, norms, h, &st);
    CTM_TRY(st);
    CTM_TRY(jacobi_rows(ctx, W, rows, n, n, n, rows >= 64 ? 32 : 16, 0, fro, ctx->si_rr_sweeps));
    CTM_TRY(row_norms(ctx, W, rows, n, n, norms));
    CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3((rows + 255) / 256), dim3(256), 0, norms, inv, rows);
    CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, W, rows, n, (long long)n, inv);
    return CTM_OK;
}

// rows of W (64 x n) -> orthonormal rows spanning the same space: two Cholesky-QR passes, and a third one -- decided on the device --
// when the first had to shift (nearly dependent rows); no host synchronisation.  Status words of the three passes go to
// `status` (9 doubles: pivot, min norm, max norm per pass; a skipped third pass reports 1), `flag3` is a device word.
// npass = 2: the third, flag-skipped pass is not even launched (three idle launches per block); a first pass that shifted is then visible
// in its status word (negative pivot) and the caller repeats the solve with three passes (svd_lanczos: units whose previous solve needed none).
int orthonormalise_block_async(ctm_ctx* ctx, double* W, int b, int n, double* G, double* Li, double* status, int* flag3, int npass) {
    // (Two passes for 32-row blocks were tried: no shifted first pass in any run on random tensors, three idle launches per block saved, < 0.5 % of
    // a sweep -- but on an SU(2)-symmetric state (RVB D = 3 tiled on the 2 x 2 cell, chi = 80: exactly dependent rows inside multiplets) a shifted
    // pass then sends the whole solve to the synchronous path.  The third, flag-skipped pass stays for every block size.)
    for (int pass = 0; pass < npass; ++pass) {
        GemmDesc g; g.M = b; g.N = b; g.K = n; g.A = W; g.sam = n; g.sak = 1; g.B = W; g.sbk = 1; g.sbn = n; g.C = G; g.ldc = b;
        if (pass == 2) g.skip_all = flag3;
        CTM_TRY(gemm_f64(ctx, g));
        if (b == 64) CTM_LAUNCH(ctx, chol64_scaled_inv_kernel<64>, dim3(1), dim3(64), 0, (const double*)G, Li, status + 3 * pass, flag3, pass);
        else if (b == 32) CTM_LAUNCH(ctx, chol64_scaled_inv_kernel<32>, dim3(1), dim3(64), 0, (const double*)G, Li, status + 3 * pass, flag3, pass);
        else { ctx->set_error("orthonormalise_block_async: 32 or 64 rows"); return CTM_ERR_BADARG; }
        GemmDesc a; a.M = b; a.N = n; a.K = b; a.A = Li; a.sam = b; a.sak = 1; a.B = W; a.sbk = n; a.sbn = 1; a.C = W; a.ldc = n;
        if (pass == 2) a.skip_all = flag3;
        CTM_TRY(gemm_f64(ctx, a));                       // in place: a workgroup reads its whole column strip before it writes
    }
    return CTM_OK;
}

// W (b x n) -= (W B^T) B for the orthonormal row basis B (m x n); twice ("twice is enough")
// local > 0 (block recurrences): in exact arithmetic the new block only overlaps the LAST `local` rows of B (three-term recurrence),
// and that overlap is O(|W|) while the overlaps with older rows are at rounding level.  The first pass then projects on those
// rows only -- it removes the large component, and unlike a full first pass it does not inject its own rounding errors
// (eps sqrt(n) |W| per coefficient) along every old direction -- and the second pass, on the now small remainder, runs over all of B.
int project_out(ctm_ctx* ctx, double* W, int b, int n, const double* B, int m, double* G, int reps, int local) {
    if (m <= 0) return CTM_OK;
    for (int rep = 0; rep < reps; ++rep) {
        const int off = (rep == 0 && reps > 1 && local > 0 && m > local) ? m - local : 0, mm = m - off;
        const double* Bo = B + (size_t)off * n;
        GemmDesc g1; g1.M = b; g1.N = mm; g1.K = n; g1.A = W; g1.sam = n; g1.sak = 1; g1.B = Bo; g1.sbk = 1; g1.sbn = n; g1.C = G; g1.ldc = mm;
        CTM_TRY(gemm_f64(ctx, g1));
        GemmDesc g2; g2.M = b; g2.N = n; g2.K = mm; g2.A = G; g2.sam = mm; g2.sak = 1; g2.B = Bo; g2.sbk = n; g2.sbn = 1; g2.C = W; g2.ldc = n;
        g2.alpha = -1.0; g2.beta = 1.0;
        CTM_TRY(gemm_f64(ctx, g2));
    }
    return CTM_OK;
}

// memory of a unit between sweeps (header row of its warm workspace, doubles): [0] calls left that skip the warm probe (svd_iter),
// [1] block steps of the last accepted Krylov solve, [2] its residual estimate / s_0
// (HDR_* are declared ahead of svd_iter)

int svd_lanczos(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged) {
    *converged = false;
    // Block size of the recurrence.  The basis a solve needs shrinks with the block: for the leading k of a slowly decaying spectrum a
    // block of b rows reaches polynomial degree m / b with m basis rows (measured on the D = 6 chi = 128 spectrum, tools emulation in
    // 2.5 k rows with b = k/8, 3.2 k with k/4, 4-4.5 k with k/2, 6 k with b = k) -- fewer corner passes in total AND a
    // smaller Ritz matrix for the dense Jacobi SVD, against twice the orthonormalisation steps and passes that are HBM-bound
    // (<= 32 rows: 0.36 ms for 32 rows against 0.62 ms for 64 at n = 16384).
    const int n = op.n, b = (ctx->lz_block == 32 || (ctx->lz_block == 0 && k > ctx->lz_block32_min_k)) ? 32 : 64;
    const int jmin = (k + b - 1) / b + 1;                        // first step with at least k + b basis rows... (k rows needed)
    int jmax = std::min((n / 2) / b, (6 * k) / b + 8);
    if (jmax < jmin + 1) return CTM_OK;
    ArenaScope scope(ctx);
    double *Uall, *Vall, *Zraw, *Wraw, *G, *norms, *inv;
    const size_t rows_max = (size_t)jmax * b;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * rows_max * n, (void**)&Uall));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (rows_max + b) * n, (void**)&Vall));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * rows_max * n, (void**)&Zraw));         // raw products U_j M
    CTM_TRY(arena_alloc(ctx, sizeof(double) * rows_max * n, (void**)&Wraw));         // raw products V_j M^T
    const bool want_mid = !op.M && op.out_uR && op.out_vRt && op.have_mid;
    double *URall = nullptr, *VRall = nullptr;                                       // half-way products U_j R^T, V_j Rt^T
    if (want_mid) {
        *op.have_mid = false;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * rows_max * n, (void**)&URall));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * rows_max * n, (void**)&VRall));
    }
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)b * (rows_max + b), (void**)&G));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * std::max<size_t>(rows_max, 1024), (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * std::max<size_t>(rows_max, 1024), (void**)&inv));
    // Sync-free recurrence (`lz_async`): the block steps are issued without waiting for the device -- orthonormalisation by
    // orthonormalise_block_async(), whose status words (pivots, row norms: 6 doubles per call) are collected in `ostat` and examined
    // at the next scheduled host synchronisation (the Ritz extraction).  Anything unusual there (a pivot that signals near
    // dependence, a breakdown of the recurrence) sends the whole solve through the synchronous path, which handles those cases.
    const bool async = ctx->lz_async && !ctx->lz_force_sync;
    double *ostat = nullptr, *Li = nullptr;
    int* flag3 = nullptr;
    const int nstat = 2 * jmax + 1, SW = 9;                    // status words per orthonormalisation
    if (async) {
        CTM_TRY(arena_alloc(ctx, sizeof(double) * SW * nstat, (void**)&ostat));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 64 * 64, (void**)&Li));
        CTM_TRY(arena_alloc(ctx, sizeof(int) * 64, (void**)&flag3));
        CTM_TRY(fill_f64(ctx, ostat, (size_t)SW * nstat, -1.0));       // (a pass that is not launched reads as "skipped")
    }
    auto resync = [&]() -> int {          // redo this solve on the synchronous path
        if (ctx->jacobi_verbose) fprintf(stderr, "[lz] n=%d asynchronous recurrence flagged: repeating on the synchronous path\n", n);
        ctx->lz_async_fallbacks += 1;
        ctx->lz_force_sync = true;
        const int st = svd_lanczos(ctx, op, k, S, Ut, Vt, converged);
        ctx->lz_force_sync = false;
        if (st == CTM_OK && op.warm_hdr) return fill_f64(ctx, op.warm_hdr + HDR_THIRD, 1, 1.0);      // the unit's next solve launches all three passes
        return st;
    };
    const double tol = resid_tol(ctx, n);
    // When to look: a Ritz extraction (dense SVD of the m x m projected matrix) costs as much as 6-8 block steps, so it is
    // scheduled, not repeated.  The operator of a unit changes slowly from sweep to sweep: the step count that was
    // accepted last time is tried first (one less when it passed with orders of magnitude to spare); cold, the first look
    // comes when the basis holds lz_first_factor * k rows; after a failed look the next one is placed where the observed (or a
    // typical) contraction of the residual estimate predicts convergence.
    double hdr[HDR_WORDS] = {0.0};
    if (op.warm_hdr) {
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(hdr, op.warm_hdr, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    int jnext;
    if (ctx->lz_first > 0) jnext = ctx->lz_first;
    else if (hdr[HDR_STEPS] >= jmin && hdr[HDR_STEPS] <= jmax && hdr[HDR_EST] > 0.0 && (int)hdr[HDR_BLOCK] == b)
        // the estimate falls by a factor 7-17 per block step (measured, D = 6 and 8): one step less when it passed with more than
        // that to spare, one more when it passed narrowly (a failed look costs five steps)
        // (with the Ritz extraction warm started from the previous one -- same basis size: 3 Jacobi sweeps; a basis that grew by a block: 8; one that
        //  shrank: a cold start, 13 -- a step less is only worth it when the estimate passed with orders of magnitude to spare: 30 until round 5
        //  made units oscillate between two step counts)
        jnext = (int)hdr[HDR_STEPS] + (hdr[HDR_EST] <= tol * (ctx->ritz_warm ? 1e-3 : 1.0 / 30.0) ? -1 : (hdr[HDR_EST] > tol / 3.0 ? 1 : 0));
    else jnext = (int)std::ceil((b == 32 ? ctx->lz_first_factor32 : ctx->lz_first_factor) * k / b);
    jnext = std::max(jmin, std::min(jnext, jmax));
    double est_prev = 0.0; int steps_prev = 0;
    double mn, mx, s0 = 0.0;
    // two Cholesky-QR passes per block where the unit's previous accepted solve never needed the third (its launches are idle then: ~130 of a
    // unit's ~1100); a first pass that shifts after all is seen at the next look and the solve is repeated (resync)
    const int npass = (ctx->lz_two_pass && op.warm_hdr && hdr[HDR_STEPS] >= 1.0 && hdr[HDR_THIRD] < 1.0) ? 2 : 3;
    bool any_third = false;
    CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, Vall, b, n, (long long)n, 0x51f15eedULL);
    if (async) CTM_TRY(orthonormalise_block_async(ctx, Vall, b, n, G, Li, ostat + SW * (nstat - 1), flag3, npass));
    else CTM_TRY(orthonormalise_block(ctx, Vall, b, n, norms, inv, &mn, &mx));
    int applications = 0;
    for (int j = 0; j < jmax; ++j) {
        double* Uj = Uall + (size_t)j * b * n;
        double* Vj = Vall + (size_t)j * b * n;
        double* Vn = Vall + (size_t)(j + 1) * b * n;
        double* Zj = Zraw + (size_t)j * b * n;
        double* Wj = Wraw + (size_t)j * b * n;
        // U_j
        CTM_TRY(matop_apply(ctx, op, true, Vj, n, b, Wj, n, want_mid ? VRall + (size_t)j * b * n : nullptr)); applications += b;
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Uj, Wj, sizeof(double) * (size_t)b * n, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_TRY(project_out(ctx, Uj, b, n, Uall, j * b, G, 2, ctx->lz_local_project ? b : 0));
        if (async) CTM_TRY(orthonormalise_block_async(ctx, Uj, b, n, G, Li, ostat + SW * (2 * j), flag3, npass));
        else {
            CTM_TRY(orthonormalise_block(ctx, Uj, b, n, norms, inv, &mn, &mx));
            s0 = std::max(s0, mx);
            if (mn <= 1e-13 * s0) { if (ctx->jacobi_verbose) fprintf(stderr, "[lz] n=%d breakdown at step %d (U)\n", n, j); return CTM_OK; }
        }
        // raw product and V_{j+1}
        CTM_TRY(matop_apply(ctx, op, false, Uj, n, b, Zj, n, want_mid ? URall + (size_t)j * b * n : nullptr)); applications += b;
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vn, Zj, sizeof(double) * (size_t)b * n, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_TRY(project_out(ctx, Vn, b, n, Vall, (j + 1) * b, G, 2, ctx->lz_local_project ? b : 0));
        if (async) CTM_TRY(orthonormalise_block_async(ctx, Vn, b, n, G, Li, ostat + SW * (2 * j + 1), flag3, npass));
        else {
            CTM_TRY(orthonormalise_block(ctx, Vn, b, n, norms, inv, &mn, &mx));
            if (mn <= 1e-13 * s0) { if (ctx->jacobi_verbose) fprintf(stderr, "[lz] n=%d breakdown at step %d (V)\n", n, j); return CTM_OK; }
        }
        const int steps = j + 1, m = steps * b;
        if (steps < jnext && steps < jmax) continue;
        if (async) {
            // the status words of every orthonormalisation so far (the copy waits for the steps issued above)
            std::vector<double> hst((size_t)SW * nstat, 0.0);
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(hst.data(), ostat, sizeof(double) * SW * nstat, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            bool bad = false, broke = false;
            double s0a = 0.0;
            auto check = [&](int slot, bool left) {
                const double* q = hst.data() + SW * slot;
                const bool third = q[7] >= 0.0;                                // the device ran the third pass (first one was shifted)
                const bool shifted = q[0] < 0.0;                               // (chol64_scaled_inv_kernel negates the pivot of a shifted first pass)
                const double last = third ? q[6] : q[3];                       // pivot of the last pass: rows orthonormal to rounding iff ~1
                if (shifted || third) any_third = true;
                if (shifted && !third) { bad = true; return; }                 // two-pass mode met a block that needs the third pass: repeat
                if (left) s0a = std::max(s0a, q[2]);
                // breakdown of the recurrence: a new block with (numerically) nothing in it -- the Krylov space has exhausted the range of a
                // rank-deficient operator (symmetric states at small chi: every solve).  Not a case for this solver on either path: leave
                // at once, as the synchronous recurrence does, instead of repeating the whole solve there to find the same thing
                if (q[1] == q[1] && !(q[1] > 1e-13 * std::max(s0a, 1e-300)) && slot != nstat - 1) { broke = true; return; }
                if (!(std::fabs(q[0]) > 0.0) || !(last > 0.5) || !(q[1] > 0.0)) bad = true;   // (NaN fails too), zero row
                if (third) ctx->lz_third_passes += 1;
            };
            check(nstat - 1, false);
            for (int jj = 0; jj <= j && !bad && !broke; ++jj) { check(2 * jj, true); check(2 * jj + 1, false); }
            if (broke && !bad) {
                if (ctx->jacobi_verbose) fprintf(stderr, "[lz] n=%d breakdown of the asynchronous recurrence by step %d (rank-deficient operator)\n", n, j + 1);
                return CTM_OK;
            }
            if (bad) return resync();
        }
        // ---- small problem T = (U_all M) V_all^T  (m x m),  coupling E = (U_all M) V_{j+1}^T  (m x b)
        ArenaScope rs(ctx);
        double *T, *E, *Ss, *Xt, *Yt, *XE, *rn;
        const int kq = std::min(k, m);
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)m * m, (void**)&T));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)m * b, (void**)&E));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * m, (void**)&Ss));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kq * m, (void**)&Xt));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kq * m, (void**)&Yt));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kq * b, (void**)&XE));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * std::max(kq, 1), (void**)&rn));
        GemmDesc gt; gt.M = m; gt.N = m; gt.K = n; gt.A = Zraw; gt.sam = n; gt.sak = 1; gt.B = Vall; gt.sbk = 1; gt.sbn = n; gt.C = T; gt.ldc = m;
        CTM_TRY(gemm_f64(ctx, gt));
        GemmDesc ge; ge.M = m; ge.N = b; ge.K = n; ge.A = Zraw; ge.sam = n; ge.sak = 1; ge.B = Vn; ge.sbk = 1; ge.sbn = n; ge.C = E; ge.ldc = b;
        CTM_TRY(gemm_f64(ctx, ge));
        // The sweeps start from the rotations of the unit's previous extraction when the workspace carries them (a look of this solve a few
        // block steps earlier, or the previous CTM sweep's solve: the recurrence starts from the same block and is continuous in the
        // operator -- measured on D = 4 chi = 64, numpy oracle: |dT| / |T| = 2e-3 ... 9e-3 between consecutive sweeps of a moving
        // environment, off-diagonal part of the rotated matrix 2e-4 ... 3e-3 s_0).  A basis that grew by whole blocks keeps its leading
        // principal part of T, so the old rotations are padded with the identity.
        // (The dense solver pads the problem to an even number of 32-row panels: the rotations are mr x mr, mr >= m.  The old rotations
        // act on the old rows and -- when the old problem was padded -- on one block of coordinates that are new basis rows now: an orthogonal
        // mixing of rows that are new anyway.)
        double* rot = nullptr; bool rot_valid = false;
        const int mr = svd_full_rot_rows(ctx, m);
        double* ritz = (ctx->ritz_warm && op.warm_hdr && hdr[HDR_RITZ_CAP] >= 16.0 + (double)mr * mr) ? op.warm_hdr + n : nullptr;
        if (ritz) {
            double rh[3];
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)mr * mr, (void**)&rot));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(rh, ritz, sizeof(rh), hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            const int mp = (int)rh[0], mrp = (int)rh[2];
            if (mp >= b && mp % b == 0 && (int)rh[1] == b && mrp >= mp && ((mp == m && mrp == mr) || (mp < m && mrp <= m))) {
                if (mrp < mr) CTM_TRY(set_identity(ctx, rot, mr, mr));
                CTM_TRY(copy2d(ctx, ritz + 16, mrp, rot, mr, mrp, mrp));
                rot_valid = true;
            }
            if (ctx->jacobi_verbose) fprintf(stderr, "[lz] n=%d Ritz region: %d rows of block %d stored, %d wanted: %s\n", n, mp, (int)rh[1], m, rot_valid ? "warm" : "cold");
        } else if (ctx->jacobi_verbose) fprintf(stderr, "[lz] n=%d no Ritz region (capacity word %.0f, need %.0f)\n", n, hdr[HDR_RITZ_CAP], 16.0 + (double)mr * mr);
        const bool save = ctx->si_enable; ctx->si_enable = false;
        ctx->force_abs = ctx->lz_abs_accuracy != 0;
        ctx->jacobi_quad_exit = ctx->lz_quad_exit;
        const long sw0 = ctx->total_sweeps;
        const int st = svd_full(ctx, T, m, kq, Ss, Xt, Yt, nullptr, rot, rot_valid);      // rows of Xt / Yt: x_i^T, y_i^T
        ctx->jacobi_quad_exit = 0.0;
        ctx->force_abs = false;
        ctx->si_enable = save;
        CTM_TRY(st);
        ctx->lz_extractions += 1; ctx->ritz_sweeps += ctx->total_sweeps - sw0;
        if (ritz) {
            const double rh[3] = {(double)m, (double)b, (double)mr};
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(ritz, rh, sizeof(rh), hipMemcpyHostToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(ritz + 16, rot, sizeof(double) * (size_t)mr * mr, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));       // (rh is a stack array)
        }
        GemmDesc gx; gx.M = kq; gx.N = b; gx.K = m; gx.A = Xt; gx.sam = m; gx.sak = 1; gx.B = E; gx.sbk = b; gx.sbn = 1; gx.C = XE; gx.ldc = b;
        CTM_TRY(gemm_f64(ctx, gx));
        CTM_TRY(row_norms(ctx, XE, kq, b, b, rn));
        std::vector<double> hr(kq), hs(kq);
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(hr.data(), rn, sizeof(double) * kq, hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(hs.data(), Ss, sizeof(double) * kq, hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        int kv = 0;                                   // Ritz values above the noise floor: the only ones that can (and must) converge
        while (kv < kq && hs[kv] > ctx->rank_tol * hs[0]) ++kv;
        const double est = *std::max_element(hr.begin(), hr.begin() + std::max(kv, 1));
        if (ctx->jacobi_verbose) fprintf(stderr, "[lz] n=%d step %d basis %d  s0=%.3e  residual estimate/s0 = %.3e\n", n, steps, m, hs[0], est / hs[0]);
        if (kq < k || (est > tol * hs[0] && steps < jmax)) {
            // place the next look where the residual estimate is predicted to pass (geometric contraction per block step)
            double rate = 0.15;
            if (est_prev > 0.0 && est < est_prev) rate = std::min(0.6, std::max(1e-3, std::pow(est / est_prev, 1.0 / (steps - steps_prev))));
            int need = (kq < k) ? jmin - steps : (int)std::ceil(std::log(0.5 * tol * hs[0] / est) / std::log(rate));
            need = std::max(1, std::min(need, 4));
            if (ctx->lz_stride > 0) need = ctx->lz_stride;
            jnext = std::min(jmax, steps + need);
            est_prev = est; steps_prev = steps;
            continue;
        }
        // ---- Ritz triplets.  Both relations are then checked on the triplets as they will be returned (after the
        // re-orthonormalisation), without further operator applications: u_i = sum_r X'[i,r] U_r, so u_i M = sum_r X'[i,r] (U_r M),
        // and the products U_r M, V_r M^T of every basis row are the ones the recurrence was built from (stored raw).
        GemmDesc gu; gu.M = k; gu.N = n; gu.K = m; gu.A = Xt; gu.sam = m; gu.sak = 1; gu.B = Uall; gu.sbk = n; gu.sbn = 1; gu.C = Ut; gu.ldc = n;
        CTM_TRY(gemm_f64(ctx, gu));
        GemmDesc gv = gu; gv.A = Yt; gv.B = Vall; gv.C = Vt;
        CTM_TRY(gemm_f64(ctx, gv));
        CTM_TRY(reorth_rows(ctx, Ut, k, n, n, 1));
        CTM_TRY(reorth_rows(ctx, Vt, k, n, n, 1));
        double *C1, *res, *Xc;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&C1));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&res));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * m, (void**)&Xc));
        double worst = 0.0, worst_op = 0.0;
        std::vector<double> r1(k);
        for (int rel = 0; rel < 2; ++rel) {        // rel 0: Ut M = S Vt ; rel 1: Vt M^T = S Ut
            const double* F = rel == 0 ? Ut : Vt; const double* Bs = rel == 0 ? Uall : Vall; const double* Pr = rel == 0 ? Zraw : Wraw;
            GemmDesc gc; gc.M = k; gc.N = m; gc.K = n; gc.A = F; gc.sam = n; gc.sak = 1; gc.B = Bs; gc.sbk = 1; gc.sbn = n; gc.C = Xc; gc.ldc = m;
            CTM_TRY(gemm_f64(ctx, gc));                                              // coordinates of the returned rows in the basis
            GemmDesc gp; gp.M = k; gp.N = n; gp.K = m; gp.A = Xc; gp.sam = m; gp.sak = 1; gp.B = Pr; gp.sbk = n; gp.sbn = 1; gp.C = C1; gp.ldc = n;
            CTM_TRY(gemm_f64(ctx, gp));
            if (want_mid) {      // the same coordinates give u_i^T R^T (rel 0) and v_i^T Rt^T (rel 1) from the stored half-way products
                GemmDesc gm = gp; gm.B = rel == 0 ? URall : VRall; gm.C = rel == 0 ? op.out_uR : op.out_vRt;
                CTM_TRY(gemm_f64(ctx, gm));
            }
            CTM_LAUNCH(ctx, resid_rows_kernel, dim3((k + 3) / 4), dim3(256), 0, C1, (long long)n, rel == 0 ? Vt : Ut, (long long)n, Ss, k, n, res);
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(r1.data(), res, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            worst = std::max(worst, *std::max_element(r1.begin(), r1.begin() + std::max(kv, 1)));
            if (ctx->lz_verify_op) {
                CTM_TRY(matop_apply(ctx, op, rel == 1, F, n, k, C1, n)); applications += k;
                CTM_LAUNCH(ctx, resid_rows_kernel, dim3((k + 3) / 4), dim3(256), 0, C1, (long long)n, rel == 0 ? Vt : Ut, (long long)n, Ss, k, n, res);
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(r1.data(), res, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
                CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                worst_op = std::max(worst_op, *std::max_element(r1.begin(), r1.begin() + std::max(kv, 1)));
            }
        }
        if (ctx->jacobi_verbose) {
            fprintf(stderr, "[lz] n=%d verified residual/s0 = %.3e after %d row applications", n, worst / hs[0], applications);
            if (ctx->lz_verify_op) fprintf(stderr, "  (with operator applications: %.3e)", worst_op / hs[0]);
            fprintf(stderr, "\n");
        }
        if (ctx->lz_verify_op) worst = std::max(worst, worst_op);
        ctx->lz_last_resid = worst / hs[0];
        ctx->lz_last_est = est / hs[0]; ctx->lz_last_steps = steps;
        if (worst > tol * hs[0] && worst <= 1e-11 * hs[0] && est <= tol * hs[0]) {
            // the Krylov recurrence has converged but the long linear combinations left the Ritz vectors a few ulps short of the
            // acceptance threshold: the caller polishes them with a warm-started subspace step
            ctx->lz_hits += 1; ctx->lz_total_steps += steps;
            return CTM_OK;
        }
        if (worst <= tol * hs[0]) {
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, Ss, sizeof(double) * k, hipMemcpyDeviceToDevice, ctx->stream));
            if (kv < k) {                             // triplets in the noise floor are reported as exact zeros (as svd_iter does)
                CTM_TRY(fill_f64(ctx, S + kv, (size_t)(k - kv), 0.0));
                CTM_TRY(fill_f64(ctx, Ut + (size_t)kv * n, (size_t)(k - kv) * n, 0.0));
                CTM_TRY(fill_f64(ctx, Vt + (size_t)kv * n, (size_t)(k - kv) * n, 0.0));
            }
            if (want_mid) {
                if (kv < k) {
                    CTM_TRY(fill_f64(ctx, op.out_uR + (size_t)kv * n, (size_t)(k - kv) * n, 0.0));
                    CTM_TRY(fill_f64(ctx, op.out_vRt + (size_t)kv * n, (size_t)(k - kv) * n, 0.0));
                }
                *op.have_mid = true;
            }
            if (op.warm_hdr) {
                CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_STEPS, 1, (double)steps));
                CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_EST, 1, std::max(est / hs[0], 1e-300)));
                CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_BLOCK, 1, (double)b));
                if (async) CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_THIRD, 1, any_third ? 1.0 : 0.0));
            }
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            *converged = true;
            ctx->lz_hits += 1; ctx->lz_total_steps += steps; ctx->lz_total_rows += (long)steps * b;
            return CTM_OK;
        }
        // estimate passed, verification did not (rounding beyond the polishing range): more basis does not help
        if (steps >= jmax) break;
        jnext = std::min(jmax, steps + 2);
        est_prev = 0.0;
    }
    return CTM_OK;
}

// rows of W (64 x n complex) -> orthonormal rows: unit-norm scaling + two Cholesky-QR passes; *ok = false on near dependence
int orthonormalise_block_c(ctm_ctx* ctx, CRows W, int rows, int n, double* norms, double* inv, double* min_norm, double* max_norm, bool* ok) {
    std::vector<double> h(rows);
    CTM_TRY(row_norms_c128(ctx, W.re, W.im, rows, n, n, norms));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *min_norm = *std::min_element(h.begin(), h.end());
    *max_norm = *std::max_element(h.begin(), h.end());
    *ok = *min_norm > 0.0;
    if (!*ok) return CTM_OK;
    CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3(1), dim3(256), 0, norms, inv, rows);
    CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, W.re, rows, n, (long long)n, inv);
    CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, W.im, rows, n, (long long)n, inv);
    ArenaScope scope(ctx);
    // near-dependent rows (e.g. the first power step of a random block): one-sided complex Jacobi in the panel layout
    auto jacobi_fallback = [&]() -> int {
        double *P, *Tp; int* d_idx;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)rows * n, (void**)&P));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)rows * n, (void**)&Tp));
        CTM_TRY(arena_alloc(ctx, sizeof(int) * 2 * rows, (void**)&d_idx));
        CTM_LAUNCH(ctx, planar_to_panel_kernel, dim3(2048), dim3(256), 0, (const double*)W.re, (const double*)W.im, (long long)n, rows, n, P, (long long)n, 0);
        std::vector<double> hh; int st;
        const double fro = host_fro(ctx, P, 2 * rows, n, n, norms, hh, &st);
        CTM_TRY(st);
        CTM_TRY(jacobi_rows(ctx, P, 2 * rows, n, n, n, 2 * BC, 0, fro, ctx->si_rr_sweeps, true));     // (rows = 32: two panels of 16 complex rows)
        std::vector<int> idx(rows); std::iota(idx.begin(), idx.end(), 0);
        CTM_TRY(panel_gather(ctx, P, n, idx, rows, n, Tp, d_idx));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(W.re, Tp, sizeof(double) * (size_t)rows * n, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(W.im, Tp + (size_t)rows * n, sizeof(double) * (size_t)rows * n, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_TRY(row_norms_c128(ctx, W.re, W.im, rows, n, n, norms));
        CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3(1), dim3(256), 0, norms, inv, rows);
        CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, W.re, rows, n, (long long)n, inv);
        CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, W.im, rows, n, (long long)n, inv);
        return CTM_OK;
    };
    double *G, *Li, *T;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * 64 * 64, (void**)&G));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * 64 * 64, (void**)&Li));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)rows * n, (void**)&T));
    double* status = ctx->d_scratch + 16;
    for (int pass = 0; pass < 2; ++pass) {
        XM w{W.re, W.im, n, false, false}, wh{W.re, W.im, n, true, true};
        CTM_TRY(xgemm(ctx, rows, rows, n, w, wh, G, G + 4096, rows));                 // G = W W^H
        CTM_LAUNCH(ctx, chol64_inv_c_kernel, dim3(1), dim3(256), 0, (const double*)G, (const double*)(G + 4096), Li, Li + 4096, status, rows);
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scratch + 16, status, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (!(ctx->h_scratch[16] > (pass == 0 ? 1e-10 : 0.5))) return jacobi_fallback();
        XM l{Li, Li + 4096, rows, false, false};
        CTM_TRY(xgemm(ctx, rows, n, rows, l, w, T, T + (size_t)rows * n, n));         // W <- L^-1 W
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(W.re, T, sizeof(double) * (size_t)rows * n, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(W.im, T + (size_t)rows * n, sizeof(double) * (size_t)rows * n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return CTM_OK;
}

// W (b x n) -= (W B^H) B for the orthonormal planar row basis B (m rows, planes `bplane` apart); twice
int project_out_c(ctm_ctx* ctx, CRows W, int b, int n, const double* Bre, const double* Bim, int m, double* G, double* T) {
    if (m <= 0) return CTM_OK;
    for (int rep = 0; rep < 2; ++rep) {
        XM w{W.re, W.im, n, false, false}, bh{Bre, Bim, n, true, true}, bb{Bre, Bim, n, false, false};
        CTM_TRY(xgemm(ctx, b, m, n, w, bh, G, G + (size_t)b * m, m));
        XM g{G, G + (size_t)b * m, m, false, false};
        CTM_TRY(xgemm(ctx, b, n, m, g, bb, T, T + (size_t)b * n, n));
        CTM_LAUNCH(ctx, sub_inplace_kernel, dim3(1024), dim3(256), 0, W.re, (const double*)T, (size_t)b * n);
        CTM_LAUNCH(ctx, sub_inplace_kernel, dim3(1024), dim3(256), 0, W.im, (const double*)(T + (size_t)b * n), (size_t)b * n);
    }
    return CTM_OK;
}

// C = B M (adjoint == false) or B M^H on planar complex rows
int matop_apply_planar(ctm_ctx* ctx, const MatOp& op, bool adjoint, const double* Bre, const double* Bim, int rows, double* Cre, double* Cim) {
    const int n = op.n;
    XM b{Bre, Bim, n, false, false};
    if (op.M) { XM m{op.M, op.Mi, n, adjoint, adjoint}; return xgemm(ctx, rows, n, n, b, m, Cre, Cim, n); }
    const int m0 = op.mid[0] ? op.mid[0] : n, m1 = op.mid[1] ? op.mid[1] : n, mw = std::max(n, std::max(m0, m1));
    ArenaScope scope(ctx);
    double *t1, *t2;
    const size_t rn = (size_t)rows * mw;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * rn, (void**)&t1));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * rn, (void**)&t2));
    // factor i as the (kin x nout) right operand: stored kin x nout when !t (ld = nout), nout x kin when t (ld = kin)
    auto f = [&](int i, bool t, bool c, int kin, int nout) { XM x{op.c[i], op.ci[i], t ? kin : nout, t, c}; return x; };
    // a block whose planes lie one behind the other is what xgemm multiplies as ONE real block of twice the rows (two real products per
    // corner pass instead of four): the intermediates are laid out that way (plane stride = rows x ld), and when all four inner
    // dimensions are equal (uniform bond dimension) the caller's block is staged into that layout and the result copied out of it
    const bool stack = ctx->xgemm_stack_rows && m0 == n && m1 == n && rows <= 64 && rows % 16 == 0;
    const size_t rw = (size_t)rows * n;
    auto x1 = [&](int ld) { XM x{t1, t1 + (stack ? (size_t)rows * ld : rn), ld, false, false}; return x; };
    auto x2 = [&](int ld) { XM x{t2, t2 + (stack ? (size_t)rows * ld : rn), ld, false, false}; return x; };
    if (stack) {
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(t2, Bre, sizeof(double) * rw, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(t2 + rw, Bim, sizeof(double) * rw, hipMemcpyDeviceToDevice, ctx->stream));
        b = x2(n);
    }
    double* Or = stack ? t2 : Cre;                       // (the last product writes a stacked block, copied out below)
    double* Oi = stack ? t2 + rw : Cim;
    if (!adjoint) {   // B opB(cB)^T opA(cA)^T opC(cC) opD(cD)
        CTM_TRY(xgemm(ctx, rows, m0, n, b, f(1, !op.t[1], false, n, m0), t1, x1(m0).im ? const_cast<double*>(x1(m0).im) : nullptr, m0));
        CTM_TRY(xgemm(ctx, rows, n, m0, x1(m0), f(0, !op.t[0], false, m0, n), t2, const_cast<double*>(x2(n).im), n));
        CTM_TRY(xgemm(ctx, rows, m1, n, x2(n), f(2, op.t[2], false, n, m1), t1, const_cast<double*>(x1(m1).im), m1));
        CTM_TRY(xgemm(ctx, rows, n, m1, x1(m1), f(3, op.t[3], false, m1, n), Or, Oi, n));
    } else {          // B opD(cD)^H opC(cC)^H conj(opA(cA)) conj(opB(cB))
        CTM_TRY(xgemm(ctx, rows, m1, n, b, f(3, !op.t[3], true, n, m1), t1, const_cast<double*>(x1(m1).im), m1));
        CTM_TRY(xgemm(ctx, rows, n, m1, x1(m1), f(2, !op.t[2], true, m1, n), t2, const_cast<double*>(x2(n).im), n));
        CTM_TRY(xgemm(ctx, rows, m0, n, x2(n), f(0, op.t[0], true, n, m0), t1, const_cast<double*>(x1(m0).im), m0));
        CTM_TRY(xgemm(ctx, rows, n, m0, x1(m0), f(1, op.t[1], true, m0, n), Or, Oi, n));
    }
    if (stack) {
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Cre, t2, sizeof(double) * rw, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Cim, t2 + rw, sizeof(double) * rw, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return CTM_OK;
}

int svd_lanczos_c(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged) {
    *converged = false;
    const int n = op.n, b = ctx->lz_block_c == 32 ? 32 : 64;          // complex rows per block (see svd_lanczos on the block size)
    const int jmin = (k + b - 1) / b + 1;
    const int jmax = std::min((n / 2) / b, (6 * k) / b + 8);
    if (jmax < jmin + 1) return CTM_OK;
    ArenaScope scope(ctx);
    const size_t rows_max = (size_t)jmax * b, bn = (size_t)b * n;
    // planar bases: re plane [rows][n], im plane at +plane
    const size_t planeU = rows_max * n, planeV = (rows_max + b) * n;
    double *Uall, *Vall, *Zraw, *Wraw, *G, *T2, *norms, *inv;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * planeU, (void**)&Uall));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * planeV, (void**)&Vall));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * planeU, (void**)&Zraw));         // raw products U_j M
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * planeU, (void**)&Wraw));         // raw products V_j M^H
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)b * (rows_max + b), (void**)&G));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * bn, (void**)&T2));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * std::max<size_t>(rows_max, 1024), (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * std::max<size_t>(rows_max, 1024), (void**)&inv));
    auto Ur = [&](int j) { CRows r{Uall + (size_t)j * bn, Uall + planeU + (size_t)j * bn}; return r; };
    auto Vr = [&](int j) { CRows r{Vall + (size_t)j * bn, Vall + planeV + (size_t)j * bn}; return r; };
    auto Zr = [&](int j) { CRows r{Zraw + (size_t)j * bn, Zraw + planeU + (size_t)j * bn}; return r; };
    auto Wr = [&](int j) { CRows r{Wraw + (size_t)j * bn, Wraw + planeU + (size_t)j * bn}; return r; };
    const double tol = resid_tol(ctx, n);
    // scheduling of the Ritz extractions: see svd_lanczos()
    double hdr[HDR_WORDS] = {0.0};
    if (op.warm_hdr) {
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(hdr, op.warm_hdr, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    int jnext;
    if (ctx->lz_first > 0) jnext = ctx->lz_first;
    else if (hdr[HDR_STEPS] >= jmin && hdr[HDR_STEPS] <= jmax && hdr[HDR_EST] > 0.0 && (int)hdr[HDR_BLOCK] == b)
        // (with the Ritz extraction warm started from the previous one -- same basis size: 3 Jacobi sweeps; a basis that grew by a block: 8; one that
        //  shrank: a cold start, 13 -- a step less is only worth it when the estimate passed with orders of magnitude to spare: 30 until round 5
        //  made units oscillate between two step counts)
        jnext = (int)hdr[HDR_STEPS] + (hdr[HDR_EST] <= tol * (ctx->ritz_warm ? 1e-3 : 1.0 / 30.0) ? -1 : (hdr[HDR_EST] > tol / 3.0 ? 1 : 0));
    else jnext = (int)std::ceil((b == 32 ? ctx->lz_first_factor32 : ctx->lz_first_factor) * k / b);
    jnext = std::max(jmin, std::min(jnext, jmax));
    double est_prev = 0.0; int steps_prev = 0;
    double mn, mx, s0 = 0.0;
    bool ok;
    CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, Vall, b, n, (long long)n, 0x51f15eedULL);
    CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, Vall + planeV, b, n, (long long)n, 0x0dd5eedULL);
    CTM_TRY(orthonormalise_block_c(ctx, Vr(0), b, n, norms, inv, &mn, &mx, &ok));
    if (!ok) return CTM_OK;
    int applications = 0;
    for (int j = 0; j < jmax; ++j) {
        const CRows Uj = Ur(j), Vj = Vr(j), Vn = Vr(j + 1), Zj = Zr(j), Wj = Wr(j);
        CTM_TRY(matop_apply_planar(ctx, op, true, Vj.re, Vj.im, b, Wj.re, Wj.im)); applications += b;
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Uj.re, Wj.re, sizeof(double) * bn, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Uj.im, Wj.im, sizeof(double) * bn, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_TRY(project_out_c(ctx, Uj, b, n, Uall, Uall + planeU, j * b, G, T2));
        CTM_TRY(orthonormalise_block_c(ctx, Uj, b, n, norms, inv, &mn, &mx, &ok));
        s0 = std::max(s0, mx);
        if (!ok || mn <= 1e-13 * s0) { if (ctx->jacobi_verbose) fprintf(stderr, "[lz-c] n=%d breakdown at step %d (U)\n", n, j); return CTM_OK; }
        CTM_TRY(matop_apply_planar(ctx, op, false, Uj.re, Uj.im, b, Zj.re, Zj.im)); applications += b;
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vn.re, Zj.re, sizeof(double) * bn, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vn.im, Zj.im, sizeof(double) * bn, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_TRY(project_out_c(ctx, Vn, b, n, Vall, Vall + planeV, (j + 1) * b, G, T2));
        CTM_TRY(orthonormalise_block_c(ctx, Vn, b, n, norms, inv, &mn, &mx, &ok));
        if (!ok || mn <= 1e-13 * s0) { if (ctx->jacobi_verbose) fprintf(stderr, "[lz-c] n=%d breakdown at step %d (V)\n", n, j); return CTM_OK; }
        const int steps = j + 1, m = steps * b;
        if (steps < jnext && steps < jmax) continue;
        ArenaScope rs(ctx);
        const int kq = std::min(k, m);
        const size_t mm = (size_t)m * m, mb = (size_t)m * b, km = (size_t)kq * m, kb = (size_t)kq * b;
        double *T, *E, *Ss, *Xt, *Yt, *XE, *rn;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * mm, (void**)&T));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * mb, (void**)&E));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * m, (void**)&Ss));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * km, (void**)&Xt));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * km, (void**)&Yt));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * kb, (void**)&XE));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * std::max(kq, 1), (void**)&rn));
        XM z{Zraw, Zraw + planeU, n, false, false}, vh{Vall, Vall + planeV, n, true, true}, vnh{Vn.re, Vn.im, n, true, true};
        CTM_TRY(xgemm(ctx, m, m, n, z, vh, T, T + mm, m));                           // T = (U_all M) V_all^H
        CTM_TRY(xgemm(ctx, m, b, n, z, vnh, E, E + mb, b));                          // E = (U_all M) V_{j+1}^H
        // warm start from the rotations of the unit's previous extraction (planar m x m, see svd_lanczos): padded with the identity when
        // the basis grew by whole blocks
        double* rot = nullptr; bool rot_valid = false;
        double* ritz = (ctx->ritz_warm && op.warm_hdr && hdr[HDR_RITZ_CAP] >= 16.0 + 2.0 * (double)m * m) ? op.warm_hdr + n : nullptr;
        if (ritz) {
            double rh[2];
            CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * mm, (void**)&rot));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(rh, ritz, sizeof(rh), hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            const int mp = (int)rh[0];
            if (mp >= b && mp <= m && mp % b == 0 && (int)rh[1] == b) {
                if (mp < m) { CTM_TRY(set_identity(ctx, rot, m, m)); CTM_TRY(fill_f64(ctx, rot + mm, mm, 0.0)); }
                CTM_TRY(copy2d(ctx, ritz + 16, mp, rot, m, mp, mp));
                CTM_TRY(copy2d(ctx, ritz + 16 + (size_t)mp * mp, mp, rot + mm, m, mp, mp));
                rot_valid = true;
            }
        }
        const long sw0 = ctx->total_sweeps;
        CTM_TRY(svd_full_c(ctx, T, T + mm, m, kq, Ss, Xt, Yt, nullptr, rot, rot_valid));      // T = Xt^H diag(Ss) Yt
        ctx->lz_extractions += 1; ctx->ritz_sweeps += ctx->total_sweeps - sw0;
        if (ritz) {
            const double rh[2] = {(double)m, (double)b};
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(ritz, rh, sizeof(rh), hipMemcpyHostToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(ritz + 16, rot, sizeof(double) * 2 * mm, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        }
        XM x{Xt, Xt + km, m, false, false}, e{E, E + mb, b, false, false};
        CTM_TRY(xgemm(ctx, kq, b, m, x, e, XE, XE + kb, b));
        CTM_TRY(row_norms_c128(ctx, XE, XE + kb, kq, b, b, rn));
        std::vector<double> hr(kq), hs(kq);
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(hr.data(), rn, sizeof(double) * kq, hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(hs.data(), Ss, sizeof(double) * kq, hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        int kv = 0;
        while (kv < kq && hs[kv] > ctx->rank_tol * hs[0]) ++kv;
        const double est = *std::max_element(hr.begin(), hr.begin() + std::max(kv, 1));
        if (ctx->jacobi_verbose) fprintf(stderr, "[lz-c] n=%d step %d basis %d  s0=%.3e  residual estimate/s0 = %.3e\n", n, steps, m, hs[0], est / hs[0]);
        if (kq < k || (est > tol * hs[0] && steps < jmax)) {
            double rate = 0.15;
            if (est_prev > 0.0 && est < est_prev) rate = std::min(0.6, std::max(1e-3, std::pow(est / est_prev, 1.0 / (steps - steps_prev))));
            int need = (kq < k) ? jmin - steps : (int)std::ceil(std::log(0.5 * tol * hs[0] / est) / std::log(rate));
            need = std::max(1, std::min(need, 4));
            if (ctx->lz_stride > 0) need = ctx->lz_stride;
            jnext = std::min(jmax, steps + need);
            est_prev = est; steps_prev = steps;
            continue;
        }
        // Ritz triplets (rows u^H, v^H); both relations checked on the returned rows from the stored raw products (see svd_lanczos())
        const size_t kn = (size_t)k * n;
        XM y{Yt, Yt + km, m, false, false}, ua{Uall, Uall + planeU, n, false, false}, va{Vall, Vall + planeV, n, false, false};
        CTM_TRY(xgemm(ctx, k, n, m, x, ua, Ut, Ut + kn, n));
        CTM_TRY(xgemm(ctx, k, n, m, y, va, Vt, Vt + kn, n));
        CTM_TRY(reorth_rows_c(ctx, Ut, k, n, 1));
        CTM_TRY(reorth_rows_c(ctx, Vt, k, n, 1));
        double *C1, *res, *Xc;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * kn, (void**)&C1));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * k, (void**)&res));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)k * m, (void**)&Xc));
        double worst = 0.0, worst_op = 0.0;
        std::vector<double> r1(2 * k);
        auto resid = [&](const double* dst, double* acc) -> int {
            CTM_LAUNCH(ctx, resid_rows_kernel, dim3((k + 3) / 4), dim3(256), 0, C1, (long long)n, dst, (long long)n, Ss, k, n, res);
            CTM_LAUNCH(ctx, resid_rows_kernel, dim3((k + 3) / 4), dim3(256), 0, C1 + kn, (long long)n, dst + kn, (long long)n, Ss, k, n, res + k);
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(r1.data(), res, sizeof(double) * 2 * k, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            for (int i = 0; i < std::max(kv, 1); ++i) *acc = std::max(*acc, std::sqrt(r1[i] * r1[i] + r1[k + i] * r1[k + i]));
            return CTM_OK;
        };
        for (int rel = 0; rel < 2; ++rel) {        // rel 0: Ut M = S Vt ; rel 1: Vt M^H = S Ut
            const double* src = rel == 0 ? Ut : Vt; const double* dst = rel == 0 ? Vt : Ut;
            XM f{src, src + kn, n, false, false};
            XM bh = rel == 0 ? XM{Uall, Uall + planeU, n, true, true} : XM{Vall, Vall + planeV, n, true, true};
            XM pr = rel == 0 ? XM{Zraw, Zraw + planeU, n, false, false} : XM{Wraw, Wraw + planeU, n, false, false};
            CTM_TRY(xgemm(ctx, k, m, n, f, bh, Xc, Xc + (size_t)k * m, m));               // coordinates of the returned rows in the basis
            XM xc{Xc, Xc + (size_t)k * m, m, false, false};
            CTM_TRY(xgemm(ctx, k, n, m, xc, pr, C1, C1 + kn, n));
            CTM_TRY(resid(dst, &worst));
            if (ctx->lz_verify_op) {
                CTM_TRY(matop_apply_planar(ctx, op, rel == 1, src, src + kn, k, C1, C1 + kn)); applications += k;
                CTM_TRY(resid(dst, &worst_op));
            }
        }
        if (ctx->jacobi_verbose) {
            fprintf(stderr, "[lz-c] n=%d verified residual/s0 = %.3e after %d row applications", n, worst / hs[0], applications);
            if (ctx->lz_verify_op) fprintf(stderr, "  (with operator applications: %.3e)", worst_op / hs[0]);
            fprintf(stderr, "\n");
        }
        if (ctx->lz_verify_op) worst = std::max(worst, worst_op);
        ctx->lz_last_resid = worst / hs[0];
        ctx->lz_last_est = est / hs[0]; ctx->lz_last_steps = steps;
        if (worst > tol * hs[0] && worst <= 1e-11 * hs[0] && est <= tol * hs[0]) {
            ctx->lz_hits += 1; ctx->lz_total_steps += steps;
            return CTM_OK;          // the caller polishes with a warm-started subspace pass
        }
        if (worst <= tol * hs[0]) {
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, Ss, sizeof(double) * k, hipMemcpyDeviceToDevice, ctx->stream));
            if (kv < k) {
                CTM_TRY(fill_f64(ctx, S + kv, (size_t)(k - kv), 0.0));
                for (int pl = 0; pl < 2; ++pl) {
                    CTM_TRY(fill_f64(ctx, Ut + pl * kn + (size_t)kv * n, (size_t)(k - kv) * n, 0.0));
                    CTM_TRY(fill_f64(ctx, Vt + pl * kn + (size_t)kv * n, (size_t)(k - kv) * n, 0.0));
                }
            }
            if (op.warm_hdr) {
                CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_STEPS, 1, (double)steps));
                CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_EST, 1, std::max(est / hs[0], 1e-300)));
                CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_BLOCK, 1, (double)b));
            }
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            *converged = true;
            ctx->lz_hits += 1; ctx->lz_total_steps += steps;
            return CTM_OK;
        }
        if (steps >= jmax) break;
        jnext = std::min(jmax, steps + 2);
        est_prev = 0.0;
    }
    return CTM_OK;
}


// ---------------------------------------------------------------------------------------------
// Stationary environments (option "warm_accept_tol" > 0; ctm_args.projector_warm_tol on the host side).  Once a run has converged the
// operator of a unit changes by ~1e-10 s_0 from sweep to sweep -- its own rounding floor: projectors carry S^-1/2 of values down to
// 1e-8 s_0 -- and a cold block Krylov solve to 6e-14 s_0 resolves the operator far below the noise it carries.  This path takes the
// previous row basis W (k rows of ONE side), completes it with pseudo-random guard rows, and does ONE Rayleigh-Ritz half step:
//   C = W op(M)  ->  one-sided Jacobi on the rows of [C | W | W half-way]  ->  fresh side F = rows / |rows|, s = |rows|, W' = rotated W
// (the relation W' op(M) = s F holds by construction), then verifies the other relation with one application on the k leading rows,
// |F op(M)^T - s W'| <= warm_accept_tol s_0, and returns (F, s, W') or nothing.  The half-way products of the two applications are
// u_i^T R^T and v_i^T Rt^T, so the projectors need no further corner passes.  The workspace keeps the FRESH side (HDR_SIDE says which):
// successive calls alternate sides, i.e. they are the half steps of a subspace iteration that follows the slowly moving operator.
// 8 corner passes on ~k + 64 / k rows and one Rayleigh-Ritz instead of ~170 passes of 32 rows and the dense SVD of the Ritz matrix.
// The residual certifies singular triplets, not that they are the largest: the caller re-solves from scratch every
// "warm_accept_max_run" accepted calls, and whenever the residual test fails.
// ---------------------------------------------------------------------------------------------
int svd_stationary(ctm_ctx* ctx, const MatOp& op, int k, int side0, double* S, double* Ut, double* Vt, bool* accepted, double* resid_rel) {
    *accepted = false; *resid_rel = 0.0;
    const int n = op.n, b = 32;
    const int p = ((k + 32 + 63) / 64) * 64, ng = 32;          // 32 orthonormal guard rows, zero rows up to the panel pairs of the Jacobi
    if (p >= n / 2 || op.M || !op.warm) return CTM_OK;
    const bool want_mid = op.out_uR && op.out_vRt && op.have_mid;
    ArenaScope scope(ctx);
    const long long ld = (want_mid ? 3LL : 2LL) * n;
    double *X, *B0, *M1 = nullptr, *norms, *inv, *res, *F, *G0, *M0 = nullptr, *C2, *M2 = nullptr, *dS;
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * ld, (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * n, (void**)&B0));
    if (want_mid) CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * n, (void**)&M1));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p, (void**)&inv));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p, (void**)&res));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p, (void**)&dS));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * p, (void**)&d_idx));
    // start block: the previous rows, ng pseudo-random guard rows made ORTHONORMAL in their orthogonal complement (the trial basis of a
    // single Rayleigh-Ritz step must be orthonormal: rows of norm 9 inflate the Ritz values of everything they are rotated with --
    // measured: residual 1e-5 s_0 on the very operator the basis came from, 8e-16 with this), zero rows behind them
    CTM_TRY(copy2d(ctx, op.warm, n, B0, n, k, n));
    {
        double* Rn = B0 + (size_t)k * n;
        CTM_TRY(fill_f64(ctx, Rn, (size_t)(p - k) * n, 0.0));
        CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, Rn, ng, n, (long long)n, 0x7e57ab1eULL + (unsigned long long)ctx->warm_accepts);
        ArenaScope ws(ctx);
        double* Gw;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)ng * k, (void**)&Gw));
        for (int rep = 0; rep < 2; ++rep) {
            GemmDesc g1; g1.M = ng; g1.N = k; g1.K = n; g1.A = Rn; g1.sam = n; g1.sak = 1; g1.B = B0; g1.sbk = 1; g1.sbn = n; g1.C = Gw; g1.ldc = k;
            CTM_TRY(gemm_f64(ctx, g1));
            GemmDesc g2; g2.M = ng; g2.N = n; g2.K = k; g2.A = Gw; g2.sam = k; g2.sak = 1; g2.B = B0; g2.sbk = n; g2.sbn = 1; g2.C = Rn; g2.ldc = n;
            g2.alpha = -1.0; g2.beta = 1.0;
            CTM_TRY(gemm_f64(ctx, g2));
            if (rep == 0) { double mn, mx; CTM_TRY(orthonormalise_block(ctx, Rn, ng, n, norms, inv, &mn, &mx)); }
        }
    }
    // first application: side0 == 0: W = right vectors, C = W M^T (fresh side: left); side0 == 1: W = left vectors, C = W M
    // (the rows behind the guard rows are zero and stay zero: only the k + ng real rows go through the corner passes)
    const int pa = k + ng;
    CTM_TRY(fill_f64(ctx, X + (size_t)pa * ld, (size_t)(p - pa) * ld, 0.0));
    CTM_TRY(matop_apply(ctx, op, side0 == 0, B0, n, pa, X, ld, M1));
    CTM_TRY(copy2d(ctx, B0, n, X + n, ld, pa, n));
    if (want_mid) CTM_TRY(copy2d(ctx, M1, n, X + 2 * (size_t)n, ld, pa, n));
    std::vector<double> h(p, 0.0);
    int st;
    const double fro = host_fro(ctx, X, p, n, ld, norms, h, &st);
    CTM_TRY(st);
    if (!(fro > 0.0)) return CTM_OK;
    ctx->jacobi_quad_exit = ctx->si_quad_exit;
    const int st_rr = jacobi_rows(ctx, X, p, ld, n, (int)ld, b, std::min(k, p - 1), fro, ctx->si_rr_sweeps, false, ctx->si_tau_both != 0);
    ctx->jacobi_quad_exit = 0.0;
    CTM_TRY(st_rr);
    CTM_TRY(row_norms(ctx, X, p, n, ld, norms));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * p, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> idx(p); std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return h[a] > h[c]; });
    const double s0 = h[idx[0]];
    int kv = 0;
    while (kv < k && h[idx[kv]] > ctx->rank_tol * s0) ++kv;
    if (kv < k || !(s0 > 0.0)) return CTM_OK;                 // numerically low rank inside the block: the regular route is the cheap one there
    std::vector<double> hs(k), hinv(k);
    for (int i = 0; i < k; ++i) { hs[i] = h[idx[i]]; hinv[i] = 1.0 / hs[i]; }
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, idx.data(), sizeof(int) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(dS, hs.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(inv, hinv.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // sorted leading k: fresh rows (normalised), rotated start rows, rotated half-way products
    double* fresh = side0 == 0 ? Ut : Vt;
    double* kept = side0 == 0 ? Vt : Ut;
    F = fresh; G0 = kept;
    CTM_TRY(gather_rows(ctx, X, ld, d_idx, k, n, F, n, inv));
    CTM_TRY(gather_rows(ctx, X + n, ld, d_idx, k, n, G0, n, nullptr));
    if (want_mid) {
        M0 = side0 == 0 ? op.out_vRt : op.out_uR;            // W Rt^T (W = V) resp. W R^T (W = U), rotated with W
        M2 = side0 == 0 ? op.out_uR : op.out_vRt;
        CTM_TRY(gather_rows(ctx, X + 2 * (size_t)n, ld, d_idx, k, n, M0, n, nullptr));
    }
    // second application, on the k fresh rows: the relation that does not hold by construction
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&C2));
    CTM_TRY(matop_apply(ctx, op, side0 != 0, F, n, k, C2, n, M2));
    CTM_LAUNCH(ctx, resid_rows_kernel, dim3((k + 3) / 4), dim3(256), 0, C2, (long long)n, G0, (long long)n, dS, k, n, res);
    std::vector<double> hr(k);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(hr.data(), res, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const double worst = *std::max_element(hr.begin(), hr.end());
    *resid_rel = worst / s0;
    if (ctx->jacobi_verbose) {
        fprintf(stderr, "[stat] n=%d k=%d p=%d side %d  residual/s0 = %.3e (accept <= %.1e), %d Jacobi sweeps\n", n, k, p, side0, worst / s0, ctx->warm_accept_tol, ctx->last_sweeps);
        if (ctx->jacobi_verbose > 1) {
            const int wi = (int)(std::max_element(hr.begin(), hr.end()) - hr.begin());
            fprintf(stderr, "[stat]   worst row %d (source row %d, s/s0 = %.3e); res/s0 at 0,1,k/2,k-2,k-1: %.2e %.2e %.2e %.2e %.2e; s_k/s0 = %.3e; source rows of the last 4: %d %d %d %d; next Ritz value/s0 %.3e (row %d)\n",
                    wi, idx[wi], hs[wi] / s0, hr[0] / s0, hr[1] / s0, hr[k / 2] / s0, hr[k - 2] / s0, hr[k - 1] / s0, hs[k - 1] / s0, idx[k - 4], idx[k - 3], idx[k - 2], idx[k - 1], h[idx[k]] / s0, idx[k]);
        }
    }
    if (!(worst <= ctx->warm_accept_tol * s0)) return CTM_OK;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, dS, sizeof(double) * k, hipMemcpyDeviceToDevice, ctx->stream));
    if (want_mid) *op.have_mid = true;
    // the workspace keeps the fresh side
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm, F, sizeof(double) * (size_t)k * n, hipMemcpyDeviceToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *accepted = true;
    return CTM_OK;
}

// complex128 twin of svd_stationary(): planar warm rows (u^H or v^H), panel layout for the Rayleigh-Ritz, no half-way products (the
// complex projectors are built from the returned vectors by the caller's corner passes, as after a full complex solve).
int svd_stationary_c(ctm_ctx* ctx, const MatOp& op, int k, int side0, double* S, double* Ut, double* Vt, bool* accepted, double* resid_rel) {
    *accepted = false; *resid_rel = 0.0;
    const int n = op.n, ng = 32;
    const int p = ((k + ng + 63) / 64) * 64;               // complex rows; 2p real panel rows
    if (p >= n / 2 || op.M || !op.warm) return CTM_OK;
    ArenaScope scope(ctx);
    const long long ld = 2LL * n;
    const int R = 2 * p;
    const size_t wkn = (size_t)k * n;
    double *XA, *XB, *norms, *nc, *inv, *res;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)R * ld, (void**)&XA));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)R * ld, (void**)&XB));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * R, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * R, (void**)&nc));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * R, (void**)&inv));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * R, (void**)&res));
    CTM_TRY(fill_f64(ctx, XB, (size_t)R * ld, 0.0));
    {   // guard rows: pseudo-random, projected out of the previous rows, orthonormalised, projected again
        ArenaScope ws(ctx);
        double *Rn, *Gw, *Tw;
        const size_t rn = (size_t)ng * n;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * rn, (void**)&Rn));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)ng * k, (void**)&Gw));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * rn, (void**)&Tw));
        CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, Rn, 2 * ng, n, (long long)n, 0x7e57ab1eULL + (unsigned long long)ctx->warm_accepts);
        XM r{Rn, Rn + rn, n, false, false}, vh{op.warm, op.warm + wkn, n, true, true}, v{op.warm, op.warm + wkn, n, false, false};
        for (int rep = 0; rep < 2; ++rep) {
            CTM_TRY(xgemm(ctx, ng, k, n, r, vh, Gw, Gw + (size_t)ng * k, k));              // G = R W^H
            XM g{Gw, Gw + (size_t)ng * k, k, false, false};
            CTM_TRY(xgemm(ctx, ng, n, k, g, v, Tw, Tw + rn, n));                           // T = G W
            CTM_LAUNCH(ctx, sub_inplace_kernel, dim3(2048), dim3(256), 0, Rn, Tw, 2 * rn);
            if (rep == 0) {
                double mn, mx; bool ok = false;
                CTM_TRY(orthonormalise_block_c(ctx, CRows{Rn, Rn + rn}, ng, n, norms, inv, &mn, &mx, &ok));
                if (!ok) return CTM_OK;
            }
        }
        CTM_LAUNCH(ctx, planar_to_panel_kernel, dim3(2048), dim3(256), 0, op.warm, op.warm + wkn, (long long)n, k, n, XB, ld, 0);
        CTM_LAUNCH(ctx, planar_to_panel_kernel, dim3(2048), dim3(256), 0, Rn, Rn + rn, (long long)n, ng, n, XB, ld, k);
    }
    // first application (all panel rows: the zero rows of the padding cost little next to 4 complex corner passes' fixed part)
    CTM_TRY(matop_apply_c(ctx, op, side0 == 0, XB, ld, R, XA, ld));
    CTM_TRY(copy2d(ctx, XB, ld, XA + n, ld, R, n));
    int st;
    std::vector<double> hh;
    const double fro = host_fro(ctx, XA, R, n, ld, norms, hh, &st);
    CTM_TRY(st);
    if (!(fro > 0.0)) return CTM_OK;
    ctx->jacobi_quad_exit = ctx->si_quad_exit;
    const int st_rr = jacobi_rows(ctx, XA, R, ld, n, (int)ld, 2 * BC, std::min(k, p - 1), fro, ctx->si_rr_sweeps, true, ctx->si_tau_both != 0);
    ctx->jacobi_quad_exit = 0.0;
    CTM_TRY(st_rr);
    std::vector<double> tmp(R), h(p, 0.0);
    CTM_TRY(row_norms(ctx, XA, R, n, ld, norms));
    CTM_LAUNCH(ctx, panel_combine_kernel, dim3((R + 255) / 256), dim3(256), 0, norms, nc, R);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(tmp.data(), nc, sizeof(double) * R, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int cr = 0; cr < p; ++cr) h[cr] = tmp[crow_re(cr)];
    std::vector<int> idx(p); std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return h[a] > h[c]; });
    const double s0 = h[idx[0]];
    int kv = 0;
    while (kv < k && h[idx[kv]] > ctx->rank_tol * s0) ++kv;
    if (kv < k || !(s0 > 0.0)) return CTM_OK;
    CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3((R + 255) / 256), dim3(256), 0, nc, inv, R);
    CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, XA, R, n, ld, inv);
    // sorted leading k complex rows, planar: the fresh side (normalised) and the rotated start rows
    double* fresh = side0 == 0 ? Ut : Vt;
    double* kept = side0 == 0 ? Vt : Ut;
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(int) * 2 * k, (void**)&d_idx));
    CTM_TRY(panel_gather(ctx, XA, ld, idx, k, n, fresh, d_idx));
    CTM_TRY(panel_gather(ctx, XA + n, ld, idx, k, n, kept, d_idx));
    // second application on the k fresh rows (padded to whole 16-row panels): the relation that does not hold by construction
    const int kp = ((k + BC - 1) / BC) * BC, R2 = 2 * kp;
    double *P2, *C2, *G2, *srep;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)R2 * n, (void**)&P2));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)R2 * n, (void**)&C2));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)R2 * n, (void**)&G2));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * R2, (void**)&srep));
    CTM_TRY(fill_f64(ctx, P2, (size_t)R2 * n, 0.0));
    CTM_TRY(fill_f64(ctx, G2, (size_t)R2 * n, 0.0));
    CTM_LAUNCH(ctx, planar_to_panel_kernel, dim3(2048), dim3(256), 0, (const double*)fresh, (const double*)(fresh + wkn), (long long)n, k, n, P2, (long long)n, 0);
    CTM_LAUNCH(ctx, planar_to_panel_kernel, dim3(2048), dim3(256), 0, (const double*)kept, (const double*)(kept + wkn), (long long)n, k, n, G2, (long long)n, 0);
    std::vector<double> hs(k), hrep(R2, 0.0);
    for (int i = 0; i < k; ++i) { hs[i] = h[idx[i]]; hrep[crow_re(i)] = hs[i]; hrep[crow_re(i) + BC] = hs[i]; }
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(srep, hrep.data(), sizeof(double) * R2, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    CTM_TRY(matop_apply_c(ctx, op, side0 != 0, P2, n, R2, C2, n));
    CTM_LAUNCH(ctx, resid_rows_kernel, dim3((R2 + 3) / 4), dim3(256), 0, (const double*)C2, (long long)n, (const double*)G2, (long long)n, (const double*)srep, R2, n, res);
    std::vector<double> tr(R2);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(tr.data(), res, sizeof(double) * R2, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double worst = 0.0;
    for (int i = 0; i < k; ++i) { const int rr = crow_re(i); worst = std::max(worst, std::sqrt(tr[rr] * tr[rr] + tr[rr + BC] * tr[rr + BC])); }
    *resid_rel = worst / s0;
    if (ctx->jacobi_verbose) fprintf(stderr, "[stat-c] n=%d k=%d p=%d side %d  residual/s0 = %.3e (accept <= %.1e), %d Jacobi sweeps\n", n, k, p, side0, worst / s0, ctx->warm_accept_tol, ctx->last_sweeps);
    if (!(worst <= ctx->warm_accept_tol * s0)) return CTM_OK;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, hs.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm, fresh, sizeof(double) * 2 * wkn, hipMemcpyDeviceToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *accepted = true;
    return CTM_OK;
}

// How far a unit's operator moved since its previous solve, measured on what both solves share without further operator
// applications: the singular values.  |s_i - s_i^prev| <= |M - M^prev|_2 (Weyl), so max_i |ds_i| / s_0 is a LOWER bound on the relative
// movement of the operator -- and in a converging CTM run, where the operator changes by a smooth perturbation, also its order of
// magnitude.  (A distance between the singular SUBSPACES is useless here: the vectors at the truncation boundary rotate by
// movement / (s_k - s_{k+1}), 1e-4 and more for an operator that moved by 1e-10, while the residual of the previous triplets -- what the
// fast path is accepted on -- is of the order of the movement itself.)  The previous values live in the header row, HDR_SPREV onwards.
// S: device pointer to the k new values.  Writes HDR_DIST (0 = no previous values) and the new values; returns the measure.
int spectrum_movement(ctm_ctx* ctx, double* hdr_row, int n, const double* S, int k, double* moved) {
    *moved = 0.0;
    if (n < HDR_SPREV + k) return CTM_OK;
    std::vector<double> h(2 * (size_t)k);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), S, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data() + k, hdr_row + HDR_SPREV, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h[k] > 0.0 && h[0] > 0.0) {
        double d = 0.0;
        for (int i = 0; i < k; ++i) d = std::max(d, std::fabs(h[i] / h[0] - h[k + i] / h[k]));      // (normalised: the move's own normalisation rescales the operator)
        *moved = std::max(d, 1e-300);
    }
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(hdr_row + HDR_SPREV, S, sizeof(double) * k, hipMemcpyDeviceToDevice, ctx->stream));
    return CTM_OK;
}

int jacobi_svd_top_op(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt) {
    const int n = op.n;
    if (n <= 0 || k <= 0 || k > n) { ctx->set_error("jacobi_svd_top: bad n/k"); return CTM_ERR_BADARG; }
    const size_t wz = (op.Mi || op.ci[0]) ? 2 : 1;
    auto keep_warm = [&]() -> int {     // the right row factor is the next call's starting basis
        if (op.warm && Vt) CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm, Vt, sizeof(double) * wz * (size_t)k * n, hipMemcpyDeviceToDevice, ctx->stream));
        return CTM_OK;
    };
    // after a full solve: with the stationary fast path enabled measure how far the previous basis was from this solve's (decides whether
    // the next call tries the fast path); the workspace then holds RIGHT vectors again
    double hdr[HDR_WORDS] = {0.0};
    auto keep_warm_dist = [&](const double* hdr_old) -> int {
        if (op.warm && op.warm_hdr && !op.M && (ctx->warm_accept_tol > 0.0 || hdr_old[HDR_SIDE] >= 1.0)) {
            double dist = 0.0;
            if (ctx->warm_accept_tol > 0.0) CTM_TRY(spectrum_movement(ctx, op.warm_hdr, n, S, k, &dist));
            const double w[5] = {dist, 0.0, 0.0, hdr_old[HDR_SSKIP], hdr_old[HDR_SFAILS]};     // HDR_DIST, HDR_SIDE, HDR_RUN, HDR_SSKIP, HDR_SFAILS
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm_hdr + HDR_DIST, w, sizeof(w), hipMemcpyHostToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            ctx->warm_last_dist = dist;
        }
        return keep_warm();
    };
    // stationary fast path (option "warm_accept_tol"): one Rayleigh-Ritz half step from the previous basis when the unit's singular values
    // hardly moved between its last two solves; *done: the triplets are in S, Ut, Vt and the workspace / header are up to date
    const bool cplx_op = op.Mi || op.ci[0];
    auto stationary_attempt = [&](bool* done) -> int {
        *done = false;
        if (!(ctx->warm_accept_tol > 0.0 && op.warm && !op.M && hdr[HDR_STEPS] >= 1.0)) return CTM_OK;
        if (hdr[HDR_SSKIP] >= 1.0) { hdr[HDR_SSKIP] -= 1.0; return fill_f64(ctx, op.warm_hdr + HDR_SSKIP, 1, hdr[HDR_SSKIP]); }
        if (!(hdr[HDR_DIST] > 0.0 && hdr[HDR_DIST] <= ctx->warm_try_factor * ctx->warm_accept_tol &&
              (ctx->warm_accept_max_run <= 0 || hdr[HDR_RUN] < ctx->warm_accept_max_run))) return CTM_OK;
        bool acc = false; double rr = 0.0;
        const int side0 = hdr[HDR_SIDE] >= 1.0 ? 1 : 0;
        if (cplx_op) CTM_TRY(svd_stationary_c(ctx, op, k, side0, S, Ut, Vt, &acc, &rr));
        else CTM_TRY(svd_stationary(ctx, op, k, side0, S, Ut, Vt, &acc, &rr));
        if (acc) {
            double mv = 0.0;
            CTM_TRY(spectrum_movement(ctx, op.warm_hdr, n, S, k, &mv));
            const double w[5] = {std::max(mv, 1e-300), side0 ? 0.0 : 1.0, hdr[HDR_RUN] + 1.0, 0.0, 0.0};
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm_hdr + HDR_DIST, w, sizeof(w), hipMemcpyHostToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            ctx->warm_accepts += 1; ctx->warm_last_dist = rr;
            *done = true;
            return CTM_OK;
        }
        // refused: the full solve follows; the next attempts back off (x2 per consecutive refusal)
        ctx->warm_rejects += 1;
        hdr[HDR_SFAILS] = std::min(hdr[HDR_SFAILS] + 1.0, 6.0);
        hdr[HDR_SSKIP] = std::ldexp(1.0, (int)hdr[HDR_SFAILS]) - 1.0;
        if (op.have_mid) *op.have_mid = false;
        return CTM_OK;
    };
    if (op.Mi || op.ci[0]) {        // complex128
        if (Ut && Vt && ctx->si_enable && k < n && n >= ctx->si_min_n) {
            bool ok = false, krylov = false;
            MatOp op1 = op;
            if (op.warm_hdr && ctx->lz_enable && k >= ctx->lz_min_k) {      // direct Krylov entry of a full-rank unit, see the real branch below
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(hdr, op.warm_hdr, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
                CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                { bool done = false; CTM_TRY(stationary_attempt(&done)); if (done) return CTM_OK; }
                if (hdr[HDR_SIDE] >= 1.0) {      // the workspace holds LEFT vectors (kept by the fast path): the warm starts below expect right vectors
                    op1.warm = nullptr; op1.warm_hdr = nullptr;
                    if (hdr[HDR_SKIP] < 1.0) hdr[HDR_SKIP] = 1.0;
                }
                if (hdr[HDR_SKIP] >= 1.0 && hdr[HDR_STEPS] >= 1.0) {
                    CTM_TRY(fill_f64(ctx, op.warm_hdr, 1, hdr[HDR_SKIP] - 1.0)); ctx->si_warm_skips += 1;
                    ctx->lz_last_resid = 1.0;
                    CTM_TRY(svd_lanczos_c(ctx, op, k, S, Ut, Vt, &ok));
                    if (ok) return keep_warm_dist(hdr);
                    op1.warm = nullptr; op1.warm_hdr = nullptr;
                }
            }
            CTM_TRY(svd_iter_c(ctx, op1, k, S, Ut, Vt, &ok, &krylov));
            if (ok) { ctx->si_hits += 1; return keep_warm_dist(hdr); }
            if (krylov) {
                ctx->lz_last_resid = 1.0;
                CTM_TRY(svd_lanczos_c(ctx, op, k, S, Ut, Vt, &ok));
                if (ok) return keep_warm_dist(hdr);
                MatOp op2 = op;
                ArenaScope ws(ctx);
                if (ctx->lz_last_resid <= 1e-11) {
                    double* w2;
                    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)k * n, (void**)&w2));
                    CTM_HIP_CHECK(ctx, hipMemcpyAsync(w2, Vt, sizeof(double) * 2 * (size_t)k * n, hipMemcpyDeviceToDevice, ctx->stream));
                    op2.warm = w2;
                }
                CTM_TRY(svd_iter_c(ctx, op2, k, S, Ut, Vt, &ok, nullptr));
                if (ok) { ctx->si_hits += 1; return keep_warm_dist(hdr); }
            }
            ctx->si_fallbacks += 1;
        }
        if (op.M) {
            if (k == n) return svd_full_c(ctx, op.M, op.Mi, n, k, S, Ut, Vt, op.warm);     // full decomposition: the workspace keeps the left vectors
            CTM_TRY(svd_full_c(ctx, op.M, op.Mi, n, k, S, Ut, Vt)); return keep_warm();
        }
        ArenaScope scope(ctx);
        const size_t nn = (size_t)n * n;
        double *R, *Rt, *M;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * nn, (void**)&R));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * nn, (void**)&Rt));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * nn, (void**)&M));
        const int m0 = op.mid[0] ? op.mid[0] : n, m1 = op.mid[1] ? op.mid[1] : n;
        XM a{op.c[0], op.ci[0], op.t[0] ? n : m0, op.t[0], false}, b{op.c[1], op.ci[1], op.t[1] ? m0 : n, op.t[1], false};
        XM c{op.c[2], op.ci[2], op.t[2] ? n : m1, op.t[2], false}, d{op.c[3], op.ci[3], op.t[3] ? m1 : n, op.t[3], false};
        CTM_TRY(xgemm(ctx, n, n, m0, a, b, R, R + nn, n));
        CTM_TRY(xgemm(ctx, n, n, m1, c, d, Rt, Rt + nn, n));
        XM rT{R, R + nn, n, true, false}, rt{Rt, Rt + nn, n, false, false};
        CTM_TRY(xgemm(ctx, n, n, n, rT, rt, M, M + nn, n));
        CTM_TRY(svd_full_c(ctx, M, M + nn, n, k, S, Ut, Vt));
        return keep_warm_dist(hdr);
    }
    if (Ut && Vt && ctx->si_enable && k < n && n >= ctx->si_min_n) {
        bool ok = false, krylov = false;
        MatOp op1 = op;
        if (op.warm_hdr && ctx->lz_enable && k >= ctx->lz_min_k) {
            // a unit whose last solve needed the Krylov solver and whose warm probe is not due yet goes there directly (no
            // 64-row rank probe either); if that should fail the regular path below starts cold
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(hdr, op.warm_hdr, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            { bool done = false; CTM_TRY(stationary_attempt(&done)); if (done) return CTM_OK; }
            if (hdr[HDR_SIDE] >= 1.0) {
                // the workspace holds LEFT vectors (kept by the fast path): the regular warm starts below expect right vectors
                op1.warm = nullptr; op1.warm_hdr = nullptr;
                if (hdr[HDR_SKIP] < 1.0) hdr[HDR_SKIP] = 1.0;
            }
            if (hdr[HDR_SKIP] >= 1.0 && hdr[HDR_STEPS] >= 1.0) {
                CTM_TRY(fill_f64(ctx, op.warm_hdr, 1, hdr[HDR_SKIP] - 1.0)); ctx->si_warm_skips += 1;
                ctx->lz_last_resid = 1.0;
                CTM_TRY(svd_lanczos(ctx, op, k, S, Ut, Vt, &ok));
                if (ok) return keep_warm_dist(hdr);
                op1.warm = nullptr; op1.warm_hdr = nullptr;
            }
        }
        CTM_TRY(svd_iter(ctx, op1, k, S, Ut, Vt, &ok, &krylov));
        if (ok) { ctx->si_hits += 1; return keep_warm_dist(hdr); }
        if (krylov) {
            ctx->lz_last_resid = 1.0;
            CTM_TRY(svd_lanczos(ctx, op, k, S, Ut, Vt, &ok));
            if (ok) return keep_warm_dist(hdr);
            // not accepted: finish with the subspace iteration (no further switching), started from the Ritz vectors when the
            // Krylov solve got close (their residual only missed the acceptance threshold by rounding)
            MatOp op2 = op;
            ArenaScope ws(ctx);
            if (ctx->lz_last_resid <= 1e-11) {
                double* w2;
                CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&w2));
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(w2, Vt, sizeof(double) * (size_t)k * n, hipMemcpyDeviceToDevice, ctx->stream));
                op2.warm = w2;
            }
            CTM_TRY(svd_iter(ctx, op2, k, S, Ut, Vt, &ok, nullptr));
            if (ok) { ctx->si_hits += 1; return keep_warm_dist(hdr); }
        }
        ctx->si_fallbacks += 1;
    }
    if (op.M) {
        if (k == n && Ut && Vt && ctx->svd_polar && n >= ctx->svd_polar_min_n) return svd_full_polar(ctx, op.M, n, S, Ut, Vt, op.warm);   // workspace: right vectors
        if (k == n) return svd_full(ctx, op.M, n, k, S, Ut, Vt, op.warm);                   // full decomposition: the workspace keeps the left vectors
        CTM_TRY(svd_full(ctx, op.M, n, k, S, Ut, Vt)); return keep_warm();
    }
    // materialise M = R^T Rt for the full decomposition: M = I * M
    ArenaScope scope(ctx);
    double *M, *I;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * n, (void**)&M));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * n, (void**)&I));
    CTM_TRY(set_identity(ctx, I, n, n));
    CTM_TRY(matop_apply(ctx, op, false, I, n, n, M, n));
    CTM_TRY(svd_full(ctx, M, n, k, S, Ut, Vt));
    return keep_warm_dist(hdr);
}

int jacobi_svd_top(ctm_ctx* ctx, const double* M, int n, int k, double* S, double* Ut, double* Vt) {
    MatOp op; op.n = n; op.M = M;
    return jacobi_svd_top_op(ctx, op, k, S, Ut, Vt);
}

int jacobi_svdvals(ctm_ctx* ctx, const double* M, const double* Mi, int n, double* S) {
    if (Mi) return svd_full_c(ctx, M, Mi, n, n, S, nullptr, nullptr);
    return svd_full(ctx, M, n, n, S, nullptr, nullptr);
}

