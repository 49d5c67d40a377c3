// Symmetric / Hermitian chi-truncation (C4v): warm restart with deflated probe, orthogonal iteration with Cholesky-QR steps,
// regular route through the SVD iteration + small Rayleigh-Ritz, full shifted Jacobi.  Split out of jacobi.hip in round 5.
#include "jacobi_internal.h"

// Warm restart of the symmetric leading-|lambda| problem when the matrix has (almost) not changed since the previous call -- the
// regime of a CTM run after its first few sweeps.  `warm` holds kk orthonormal rows (the previous invariant subspace).
//  (a) Rayleigh-Ritz inside the warm subspace: H = Q A Q^T (kk x kk), dense eigendecomposition, rotate, and the residuals
//      |q_i A - lambda_i q_i| of ALL kk pairs must pass the same threshold as the cold iteration.
//  (b) Residuals certify eigenpairs, not that they are the LEADING ones.  A block of 64 fresh pseudo-random rows (a different
//      seed every call) is iterated three times on the operator deflated by the accepted subspace (orthonormalised in between);
//      its largest Ritz singular value must not exceed the smallest accepted |lambda|: a direction the warm subspace misses
//      would show up there exactly as it would among the guard rows of the cold iteration after three applications.
// Not accepted (either test fails, rows missing, rank deficiency) -> the caller runs the regular iteration.  ~100 small launches
// instead of four half steps with a 128-row Jacobi each.
// inv[i] = 1 / x[i] where x[i] > rel * max(x), else 0   (rows <= 64, one wave)
__global__ __launch_bounds__(64) void inv_rel_kernel(const double* __restrict__ x, double* __restrict__ inv, int rows, double rel) {
    const int i = threadIdx.x;
    const double v = (i < rows) ? x[i] : 0.0;
    double m = v;
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if (i < rows) inv[i] = (v > rel * m && v > 0.0) ? 1.0 / v : 0.0;
}

// symmetric positive semi-definite 64 x 64 G: out[0] = |G|_F (a rigorous upper bound of lambda_max); out[1] = rho + |G x - rho x| for
// the unit vector x after `iters` power steps, rho = x^T G x -- G has an eigenvalue in [rho - r, rho + r], the largest one once the
// iteration has turned x towards the leading eigenspace (a cluster or an exact tie at the top only makes r smaller).
// One wave, thread i keeps row i in registers.
__global__ __launch_bounds__(64) void sym64_lmax_kernel(const double* __restrict__ G, int iters, double* __restrict__ out) {
    const int i = threadIdx.x;
    double row[64];
    double f = 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) { row[j] = G[i * 64 + j]; f += row[j] * row[j]; }
    for (int off = 32; off > 0; off >>= 1) f += __shfl_xor(f, off, 64);
    double xi = 1.0 + 0.37 * (double)((i * 29) % 64) / 64.0;       // generic positive start
    double est = 0.0;
    for (int it = 0; it <= iters; ++it) {
        const double q = wave_sum(xi * xi);                     // (DPP moves + four readlanes: no ds_bpermute round trips on the critical path)
        if (!(q > 0.0)) break;
        double rn = __builtin_amdgcn_rsq(q);
        rn = rn * (1.5 - 0.5 * q * rn * rn);
        rn = rn * (1.5 - 0.5 * q * rn * rn);
        const double xn = xi * rn;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int j = 0; j < 64; j += 4) {                       // x_j of lane j by lane broadcast (compile-time lane: v_readlane), no LDS round trip
            a0 += row[j] * lane_bcast(xn, j); a1 += row[j + 1] * lane_bcast(xn, j + 1);
            a2 += row[j + 2] * lane_bcast(xn, j + 2); a3 += row[j + 3] * lane_bcast(xn, j + 3);
        }
        xi = (a0 + a1) + (a2 + a3);                             // (G x)_i
        if (it == iters) {
            const double rho = wave_sum(xn * xi);
            const double r2 = wave_sum((xi - rho * xn) * (xi - rho * xn));
            est = rho + sqrt(r2);
        }
    }
    if (i == 0) { out[0] = sqrt(f); out[1] = est; }
}

// out[i,:] = sign(d[i]) * q[i,:]  (sign(0) = +1)
__global__ void signed_rows_kernel(const double* __restrict__ q, const double* __restrict__ d, int dstride, int rows, int n, double* __restrict__ out) {
    const size_t tot = (size_t)rows * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x)
        out[e] = (d[(e / n) * dstride] < 0.0) ? -q[e] : q[e];
}

// real embedding of a Hermitian matrix for ROW vectors [x y] <-> x + iy:  [x y] [[Ar, Ai], [-Ai, Ar]] = [Re, Im] of (x + iy)(Ar + i Ai)
__global__ void embed_herm_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai, int n, double* __restrict__ Ae) {
    const size_t n2 = 2 * (size_t)n, tot = n2 * n2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / n2, c = e - r * n2;
        const size_t rr = r % n, cc = c % n;
        const bool lo = r >= (size_t)n, ri = c >= (size_t)n;
        Ae[e] = (lo == ri) ? Ar[rr * n + cc] : (lo ? -Ai[rr * n + cc] : Ai[rr * n + cc]);
    }
}

// complex rows x + iy (planar: re plane, im plane, k x n each) -> 2k real rows of length 2n: [x y] and [-y x] (the row times i)
__global__ void embed_rows_kernel(const double* __restrict__ re, const double* __restrict__ im, int k, int n, double* __restrict__ out) {
    const size_t n2 = 2 * (size_t)n, tot = 2 * (size_t)k * n2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / n2, c = e - r * n2;
        const size_t i = r >> 1, cc = c % n;
        const bool second = r & 1, right = c >= (size_t)n;
        const double x = re[i * n + cc], y = im[i * n + cc];
        out[e] = second ? (right ? x : -y) : (right ? y : x);
    }
}

// `embedded`: As is the real 2n x 2n embedding of a Hermitian matrix and `warm` the embedded rows ([x y] and [-y x] per complex row
// x + iy) of eigh_warm_verify_c: only the keep-the-vectors route is taken (a rotation inside the doubly degenerate real spectrum
// would not come back as complex vectors), the workspace is not written and D receives all kk Rayleigh quotients.
// Adaptive state of the orthogonal iteration (contraction rate of the last accepted solve, back-off after a flat spectrum): a property
// of the PROBLEM, i.e. of the caller's warm workspace -- contexts are shared by problems and handed to units dynamically.  It lives in
// the workspace's own header row (the n doubles behind its vectors, include/ctm_hip.h): it is born zero with the workspace and dies
// with it (rounds 3-4 kept it in a process-wide map keyed by the workspace ADDRESS, which a new allocation at the same address
// inherited).  eigh_warm_verify reads it with its first device->host copies (no extra synchronisation) into ctx->orth_cur; a change is
// written back by one tiny launch.
struct OrthState { double rate = 0.0; int skip = 0, backoff = 0; double theta_k = 0.0, theta_0 = 0.0, c_ratio = 0.0; };     // (theta: last kept / largest |Ritz value| of the last accepted look; c_ratio: contraction per application of the last UNSHIFTED solve)
constexpr int ORTH_HDR_WORDS = 6;
__global__ void set_words6_kernel(double* dst, double a0, double a1, double a2, double a3, double a4, double a5) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3; dst[4] = a4; dst[5] = a5; }
}
static OrthState& orth_cur(ctm_ctx* ctx) { return *reinterpret_cast<OrthState*>(ctx->orth_cur_storage); }
static_assert(sizeof(OrthState) <= sizeof(((ctm_ctx*)nullptr)->orth_cur_storage), "ctm_ctx::orth_cur_storage too small");
static OrthState orth_state_get(ctm_ctx* ctx) { return orth_cur(ctx); }
static void orth_state_load(ctm_ctx* ctx, const double* w) {          // w: the ORTH_HDR_WORDS doubles of a header row (host), or nullptr
    OrthState os;
    if (w) { os.rate = w[0]; os.skip = (int)w[1]; os.backoff = (int)w[2]; os.theta_k = w[3]; os.theta_0 = w[4]; os.c_ratio = w[5]; }
    if (!(os.rate >= 0.0 && os.rate <= 1.0) || os.skip < 0 || os.skip > 4096 || os.backoff < 0 || os.backoff > 4096) os = OrthState();   // (not a state: a foreign header)
    orth_cur(ctx) = os;
}
static int orth_state_put(ctm_ctx* ctx, double* hdr, const OrthState& os) {
    orth_cur(ctx) = os;
    if (hdr) CTM_LAUNCH(ctx, set_words6_kernel, dim3(1), dim3(64), 0, hdr, os.rate, (double)os.skip, (double)os.backoff, os.theta_k, os.theta_0, os.c_ratio);
    return CTM_OK;
}

static int eigh_warm_verify(ctm_ctx* ctx, const double* As, int n, int kk, int k_out, double* warm, double* D, double* Ut, bool* accepted,
                            bool embedded = false, bool* norms_ok = nullptr, double* moved = nullptr, const double* state_hdr = nullptr) {
    *accepted = false;
    if (norms_ok) *norms_ok = false;
    if (moved) *moved = 0.0;
    orth_state_load(ctx, nullptr);
    if (kk > n / 4 || kk < 2) return CTM_OK;
    ArenaScope scope(ctx);
    const int pb = 64;
    double *norms, *inv, *Q, *Y, *H, *Dk, *Th, *Q2, *Y2, *res;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * std::max(kk, pb), (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * std::max(kk, pb), (void**)&inv));
    std::vector<double> h(std::max(kk, pb)), hd(kk), hn(kk);
    double hstate[ORTH_HDR_WORDS] = {0.0};
    CTM_TRY(row_norms(ctx, warm, kk, n, n, norms));       // read back with the residuals below (one host synchronisation for both)
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(hn.data(), norms, sizeof(double) * kk, hipMemcpyDeviceToHost, ctx->stream));
    // the adaptive state of this workspace travels with the same synchronisation (a fresh workspace: zeros = the default state)
    if (state_hdr) CTM_HIP_CHECK(ctx, hipMemcpyAsync(hstate, state_hdr, sizeof(hstate), hipMemcpyDeviceToHost, ctx->stream));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * n, (void**)&Q));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * n, (void**)&Y));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * kk, (void**)&H));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * kk, (void**)&Dk));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * kk, (void**)&Th));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * n, (void**)&Q2));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * n, (void**)&Y2));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * kk, (void**)&res));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(Q, warm, sizeof(double) * (size_t)kk * n, hipMemcpyDeviceToDevice, ctx->stream));
    CTM_TRY(reorth_rows(ctx, Q, kk, n, n, 1));                       // rounding drift of many restarts
    CTM_TRY(rows_times(ctx, Q, n, kk, n, n, As, false, Y, n));      // Y = Q A
    // stationary matrix: the previous Ritz vectors ARE the eigenvectors -- Rayleigh quotients d_i = q_i A q_i^T, residuals
    // |q_i A - d_i q_i|, order by |d| unchanged: nothing to rotate.  Otherwise the Rayleigh-Ritz inside the subspace.
    CTM_TRY(row_dots(ctx, Y, Q, kk, n, n, Dk));
    CTM_LAUNCH(ctx, resid_rows_kernel, dim3((kk + 3) / 4), dim3(256), 0, (const double*)Y, (long long)n, (const double*)Q, (long long)n,
               (const double*)Dk, kk, n, res);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), res, sizeof(double) * kk, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(hd.data(), Dk, sizeof(double) * kk, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (state_hdr) orth_state_load(ctx, hstate);
    for (int i = 0; i < kk; ++i) if (!(std::fabs(hn[i] - 1.0) < 1e-6)) {
        if (ctx->jacobi_verbose) fprintf(stderr, "[eigh-warm] n=%d kk=%d: workspace row %d has norm %.3e (no complete previous subspace)\n", n, kk, i, hn[i]);
        return CTM_OK;
    }
    if (norms_ok) *norms_ok = true;
    // The last kept rows may not be eigenvectors: when the kk-th |lambda| is shared by a pair of opposite sign (or a multiplet) that
    // the workspace cuts, its last row is a mixture.  `ke` = the leading rows that are (at least the k_out the caller uses plus one);
    // only those are deflated by the probe and compared with it.
    int ke = 0;
    while (ke < kk && h[ke] <= resid_tol(ctx, n) * std::fabs(hd[0])
           && (ke == 0 || std::fabs(hd[ke]) <= std::fabs(hd[ke - 1]) + 1e-12 * std::fabs(hd[0]))) ++ke;      // ties may sit in either order
    if (embedded) ke &= ~1;
    bool as_is = ke >= std::min(kk, k_out + (embedded ? 2 : 1));
    if (as_is) Q2 = Q;
    else if (embedded) return CTM_OK;
    else {
        ke = kk;
        GemmDesc gh; gh.M = kk; gh.N = kk; gh.K = n; gh.A = Y; gh.sam = n; gh.sak = 1; gh.B = Q; gh.sbk = 1; gh.sbn = n; gh.C = H; gh.ldc = kk;
        CTM_TRY(gemm_f64(ctx, gh));                                      // H = Y Q^T
        // A subspace that is not invariant cannot pass, whatever its Rayleigh-Ritz finds: the residuals of the kk Ritz pairs are the rows of
        // Th (Y - H Q), an orthogonal rotation of R = Y - H Q, so sum_i res_i^2 = |R|_F^2 and max_i res_i >= |R|_F / sqrt(kk).  While the
        // environment moves (every sweep before stationarity) this costs three launches instead of a kk x kk Jacobi eigensolver.
        if (ctx->eigh_warm_early_reject) {
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Y2, Y, sizeof(double) * (size_t)kk * n, hipMemcpyDeviceToDevice, ctx->stream));
            GemmDesc gr; gr.M = kk; gr.N = n; gr.K = kk; gr.A = H; gr.sam = kk; gr.sak = 1; gr.B = Q; gr.sbk = n; gr.sbn = 1; gr.C = Y2; gr.ldc = n;
            gr.alpha = -1.0; gr.beta = 1.0;
            CTM_TRY(gemm_f64(ctx, gr));
            CTM_TRY(row_norms(ctx, Y2, kk, n, n, res));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), res, sizeof(double) * kk, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            double fro2 = 0.0, l0 = 0.0;
            for (int i = 0; i < kk; ++i) { fro2 += h[i] * h[i]; l0 = std::max(l0, std::fabs(hd[i])); }
            const double thr1 = 2.0 * resid_tol(ctx, n) * l0;            // (l0 from the Rayleigh quotients of the rows as they are: factor 2 of slack)
            if (moved && l0 > 0.0) *moved = std::sqrt(fro2) / l0;
            if (!(fro2 <= (double)kk * thr1 * thr1)) {
                if (ctx->jacobi_verbose) fprintf(stderr, "[eigh-warm] n=%d kk=%d: subspace residual |R|_F / |l0| = %.3e (not invariant)\n", n, kk, std::sqrt(fro2) / std::max(l0, 1e-300));
                return CTM_OK;
            }
        }
        const bool save = ctx->si_enable; ctx->si_enable = false;
        const int st = jacobi_eigh_top(ctx, H, kk, kk, Dk, Th, nullptr); // rows of Th = eigenvectors, ordered by |lambda|
        ctx->si_enable = save;
        CTM_TRY(st);
        GemmDesc r1; r1.M = kk; r1.N = n; r1.K = kk; r1.A = Th; r1.sam = kk; r1.sak = 1; r1.B = Q; r1.sbk = n; r1.sbn = 1; r1.C = Q2; r1.ldc = n;
        CTM_TRY(gemm_f64(ctx, r1));
        GemmDesc r2 = r1; r2.B = Y; r2.C = Y2;
        CTM_TRY(gemm_f64(ctx, r2));
        CTM_LAUNCH(ctx, resid_rows_kernel, dim3((kk + 3) / 4), dim3(256), 0, (const double*)Y2, (long long)n, (const double*)Q2, (long long)n,
                   (const double*)Dk, kk, n, res);
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), res, sizeof(double) * kk, hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(hd.data(), Dk, sizeof(double) * kk, hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    const double lam0 = std::fabs(hd[0]), lamk = std::fabs(hd[ke - 1]);
    const double worst = *std::max_element(h.begin(), h.begin() + ke);
    if (ctx->jacobi_verbose) fprintf(stderr, "[eigh-warm] n=%d kk=%d%s (%d rows)  max resid/|l0| = %.3e  |l_ke|/|l0| = %.3e\n", n, kk, as_is ? " (vectors kept)" : "", ke, worst / std::max(lam0, 1e-300), lamk / std::max(lam0, 1e-300));
    if (!(lam0 > 0.0) || !(lamk > ctx->rank_tol * lam0) || !(worst <= resid_tol(ctx, n) * lam0)) return CTM_OK;
    // (b) probe of the deflated operator: Z <- orth(Z A_perp) twice, then the largest singular value of Z A_perp
    double *Z, *Zn, *G, *G64, *Mo, *bnd;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)pb * n, (void**)&Z));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)pb * n, (void**)&Zn));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)pb * kk, (void**)&G));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * pb * pb, (void**)&G64));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * pb * pb, (void**)&Mo));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2, (void**)&bnd));
    CTM_LAUNCH(ctx, hash_fill_kernel, dim3(256), dim3(256), 0, Z, pb, n, (long long)n, 0x51ED270BULL + 0x9E3779B97F4A7C15ULL * (unsigned long long)(++ctx->eigh_probe_calls));
    for (int q = 0; q < 3; ++q) {       // the accepted subspace is projected out after every application (its part of a row grows by |l_0 / l_kk| each time)
        CTM_TRY(rows_times(ctx, Z, n, pb, n, n, As, false, Zn, n));
        std::swap(Z, Zn);
        CTM_TRY(project_out(ctx, Z, pb, n, Q2, ke, G, 1));
        if (q == 2) break;
        // (an orthonormalisation does not change the span: the rows only have to be orthonormal before the LAST application, whose
        //  Gram matrix is read as Ritz values; after the first application the rows stay as they are -- the pivoted factorisation
        //  below is rank revealing relative to the largest pivot, and the direction the test is after is the dominant one)
        if (q == 0 && ctx->eigh_probe_orth_once) continue;
        // orthonormal basis of the significant part of the row space: Gram matrix, pivoted Cholesky stopped at 1e-10 of the largest
        // pivot (rank revealing: the rows of a probe of a fast decaying spectrum are numerically dependent), rows <- L_pp^-1 (pivot rows)
        GemmDesc go; go.M = pb; go.N = pb; go.K = n; go.A = Z; go.sam = n; go.sak = 1; go.B = Z; go.sbk = 1; go.sbn = n; go.C = G64; go.ldc = pb;
        CTM_TRY(gemm_f64(ctx, go));
        CTM_LAUNCH(ctx, pivchol64_inv_kernel, dim3(1), dim3(64), 0, (const double*)G64, 1e-10, Mo, bnd);
        GemmDesc ga; ga.M = pb; ga.N = n; ga.K = pb; ga.A = Mo; ga.sam = pb; ga.sak = 1; ga.B = Z; ga.sbk = n; ga.sbn = 1; ga.C = Zn; ga.ldc = n;
        CTM_TRY(gemm_f64(ctx, ga));
        std::swap(Z, Zn);
    }
    // the largest singular value mu of the last product from its 64 x 64 Gram matrix: mu^2 <= |G|_F, mu^2 ~ rho + r of a power iterate
    GemmDesc gg; gg.M = pb; gg.N = pb; gg.K = n; gg.A = Z; gg.sam = n; gg.sak = 1; gg.B = Z; gg.sbk = 1; gg.sbn = n; gg.C = G64; gg.ldc = pb;
    CTM_TRY(gemm_f64(ctx, gg));
    CTM_LAUNCH(ctx, sym64_lmax_kernel, dim3(1), dim3(64), 0, (const double*)G64, 48, bnd);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), bnd, sizeof(double) * 2, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const double mu_hi = std::sqrt(std::max(h[0], 0.0)), mu_lo = std::sqrt(std::max(h[1], 0.0));
    // the block is orthonormal to ~1e-6 only: same slack in the threshold (8 accepted pairs lie beyond the ones the caller uses)
    const double thr = lamk * (1.0 + 1e-6) + resid_tol(ctx, n) * lam0;
    const double mu = (mu_hi <= thr) ? mu_hi : mu_lo;                // undecided by the Frobenius bound: Rayleigh quotient + residual of the power iterate
    if (ctx->jacobi_verbose) fprintf(stderr, "[eigh-warm] probe: largest Ritz value outside / |l_kk| in [%.6f, %.6f]\n", mu_lo / lamk, mu_hi / lamk);
    if (!(mu <= thr)) { ctx->eigh_warm_rejects += 1; return CTM_OK; }
    if (embedded) {
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(D, Dk, sizeof(double) * ke, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->eigh_warm_hits += 1;
        *accepted = true;
        return CTM_OK;
    }
    if (ke == kk) CTM_HIP_CHECK(ctx, hipMemcpyAsync(warm, Q2, sizeof(double) * (size_t)kk * n, hipMemcpyDeviceToDevice, ctx->stream));
    // the regular iteration keeps the right vectors v_i as the warm basis and returns the left ones, u_i = sign(lambda_i) v_i: same
    // convention here, so that a run does not change the gauge of its environment legs when it switches between the two paths
    CTM_LAUNCH(ctx, signed_rows_kernel, dim3(256), dim3(256), 0, (const double*)Q2, (const double*)Dk, 1, k_out, n, Ut);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(D, Dk, sizeof(double) * k_out, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->eigh_warm_hits += 1;
    *accepted = true;
    return CTM_OK;
}

// Complex Hermitian twin of the warm restart: everything is checked on the real embedding (2n x 2n symmetric, every eigenvalue twice),
// where the real routine applies unchanged: Rayleigh quotients and residuals of the embedded previous vectors, and the deflated probe
// (the largest singular value of the deflated operator is the same number in the embedding).  Only the stationary case -- the
// previous vectors are the eigenvectors -- is taken; they are returned as they are (times sign(lambda), see above).
static int eigh_warm_verify_c(ctm_ctx* ctx, const double* Asr, const double* Asi, int n, int kk, int k_out, double* warm, double* D, double* Ut,
                              bool* accepted) {
    *accepted = false;
    if (2 * kk > (2 * n) / 4 || kk < 2) return CTM_OK;
    ArenaScope scope(ctx);
    const size_t kn = (size_t)kk * n;
    double *Ae, *Qe, *De;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 4 * (size_t)n * n, (void**)&Ae));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 4 * kn, (void**)&Qe));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * kk, (void**)&De));
    CTM_LAUNCH(ctx, embed_herm_kernel, dim3(2048), dim3(256), 0, Asr, Asi, n, Ae);
    CTM_LAUNCH(ctx, embed_rows_kernel, dim3(1024), dim3(256), 0, (const double*)warm, (const double*)(warm + kn), kk, n, Qe);
    CTM_TRY(eigh_warm_verify(ctx, Ae, 2 * n, 2 * kk, 2 * k_out, Qe, De, nullptr, accepted, true));
    if (!*accepted) return CTM_OK;
    const size_t on = (size_t)k_out * n;
    CTM_LAUNCH(ctx, signed_rows_kernel, dim3(256), dim3(256), 0, (const double*)warm, (const double*)De, 2, k_out, n, Ut);
    CTM_LAUNCH(ctx, signed_rows_kernel, dim3(256), dim3(256), 0, (const double*)(warm + kn), (const double*)De, 2, k_out, n, Ut + on);
    CTM_HIP_CHECK(ctx, hipMemcpy2DAsync(D, sizeof(double), De, 2 * sizeof(double), sizeof(double), k_out, hipMemcpyDeviceToDevice, ctx->stream));
    return CTM_OK;
}

// Orthogonal iteration for the symmetric truncation while the matrix still changes from call to call (the C4v corner before
// stationarity: eigh_warm_verify() has just refused the previous subspace).  The regular route -- the SVD block iteration with a
// one-sided Jacobi Rayleigh-Ritz of the 128 rows after EVERY application, then a second Rayleigh-Ritz that separates +-lambda --
// spends 8 of its 9 ms at n = 1024 in ~65 latency-bound launches of the 64 x 64 LDS eigensolver.  A symmetric matrix needs neither the
// left/right bookkeeping nor Ritz values before the test that can accept them:
//   Q_0 = [previous vectors | pseudo-random rows projected off them];   Q_{j+1} = orth(Q_j A)   (block Cholesky-QR: one pass while
//   only the conditioning matters, the full two/three passes for the basis the Rayleigh-Ritz uses);
//   from the fourth application on (the guard rows have seen the operator three times -- the `sound` rule of svd_iter()):
//   T = Q A Q^T (p x p), T = Z^T diag(theta) Z, x_i = z_i Q, residuals |x_i A - theta_i x_i| for the kk leading |theta|.
// Same acceptance threshold as the regular iteration; the orthonormality of Q that the residuals rely on is measured
// (|Q Q^T - I| row norms), not assumed.  Anything unexpected -- no complete previous subspace, numerically low rank inside the block,
// Q not orthonormal to 1e-12, no acceptance after eigh_orth_max applications -- returns with *accepted = false and the regular route runs.
// Returned gauge: rows aligned with the previous vectors (warm_i <- sign<x_i, warm_i> x_i, u_i = sign(theta_i) warm_i), as the
// regular route and the warm restart return them.
static int eigh_orth_iter(ctm_ctx* ctx, const double* As, int n, int kk, int k_out, double* warm, double* D, double* Ut, bool* accepted,
                          bool warm_checked, double moved, double* state_hdr) {
    *accepted = false;
    int p = kk + std::max(32, kk / 2);
    p = ((p + 63) / 64) * 64;
    // more guard rows (whole 64-row blocks): the residual contracts by |lambda_{p+1} / lambda_kk| per application, so a spectrum that
    // decays slowly behind the kept pairs (signed random C4v tensors: 0.25-0.33 with 55 guard rows) needs fewer applications with more
    if (ctx->eigh_orth_extra_blocks > 0 && p + 64 * ctx->eigh_orth_extra_blocks < n / 2) p += 64 * ctx->eigh_orth_extra_blocks;
    if (kk < 2 || p >= n / 2) return CTM_OK;
    ArenaScope scope(ctx);
    const int nb = p / 64, pr = p - kk;
    double *norms, *Q, *Y, *G, *Gp, *Li, *status, *T, *Dp, *Zt, *X, *AX, *res, *E, *dots;
    int* flag3;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * n, (void**)&Q));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * n, (void**)&Y));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 64 * 64, (void**)&G));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * p, (void**)&Gp));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 64 * 64, (void**)&Li));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 16, (void**)&status));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * 4, (void**)&flag3));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * p, (void**)&T));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * p, (void**)&Dp));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * p, (void**)&Zt));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * n, (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * n, (void**)&AX));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * kk, (void**)&res));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * p, (void**)&E));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * kk, (void**)&dots));
    std::vector<double> h(p), hd(p), he(p);
    // Two applications per Cholesky-QR step (option eigh_orth_double): the step costs 8 dependent launches against one for the product.
    // Each 64-row block is factorised on its own after the blocks before it have been projected out, so what matters is the spread
    // INSIDE a block, squared: (|lambda_kk| / |lambda_0|)^2 for the kept rows -- taken only while the previous look of this workspace
    // measured that ratio above eigh_orth_double_min_ratio (Gram matrix of condition <= 1e12: the scaled Cholesky passes cope; never on
    // the quickly decaying spectrum of a positive state, which needs four to six applications anyway).  Value 2 adds the shift
    // Q (A^2 - c^2/2), c = |lambda_kk| x the contraction per application an unshifted solve of this workspace measured (an estimate of
    // the largest |eigenvalue| the block does not hold): |lambda^2 - c^2/2| <= c^2/2 for |lambda| <= c halves what is left of the rest.
    double dbl_shift = 0.0; bool dbl = false; double* Z2 = nullptr;
    if (ctx->eigh_orth_double) {
        const OrthState os0 = orth_state_get(ctx);
        if (os0.theta_0 > 0.0 && os0.theta_k > ctx->eigh_orth_double_min_ratio * os0.theta_0) {
            dbl = true;
            if (ctx->eigh_orth_double >= 2 && os0.c_ratio > 0.0 && os0.c_ratio < 0.7) { const double c = os0.c_ratio * os0.theta_k; dbl_shift = 0.5 * c * c; }
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * n, (void**)&Z2));
        }
    }
    if (!warm_checked) {                     // (the warm restart that has just refused the subspace has looked at the row norms already)
        CTM_TRY(row_norms(ctx, warm, kk, n, n, norms));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * kk, hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < kk; ++i) if (!(std::fabs(h[i] - 1.0) < 1e-6)) return CTM_OK;          // no complete previous subspace
    }
    // start: the previous vectors and pseudo-random guard rows as they are -- the first orthonormalisation (after the first
    // application) is a Gram-Schmidt in this order, which takes the previous vectors out of the guard rows anyway
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(Q, warm, sizeof(double) * (size_t)kk * n, hipMemcpyDeviceToDevice, ctx->stream));
    double* Rn = Q + (size_t)kk * n;
    CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, Rn, pr, n, (long long)n, 0x1234567ULL);
    auto chol_pass = [&](double* Wb, int mode) -> int {
        GemmDesc g; g.M = 64; g.N = 64; g.K = n; g.A = Wb; g.sam = n; g.sak = 1; g.B = Wb; g.sbk = 1; g.sbn = n; g.C = G; g.ldc = 64;
        if (mode == 2) g.skip_all = flag3;
        CTM_TRY(gemm_f64(ctx, g));
        CTM_LAUNCH(ctx, chol64_scaled_inv_kernel<64>, dim3(1), dim3(64), 0, (const double*)G, Li, status + 3 * mode, flag3, mode);
        GemmDesc a; a.M = 64; a.N = n; a.K = 64; a.A = Li; a.sam = 64; a.sak = 1; a.B = Wb; a.sbk = n; a.sbn = 1; a.C = Wb; a.ldc = n;
        if (mode == 2) a.skip_all = flag3;
        return gemm_f64(ctx, a);                          // in place: a workgroup reads its whole column strip before it writes
    };
    auto orth = [&](double* W, bool full) -> int {
        for (int blk = 0; blk < nb; ++blk) {
            double* Wb = W + (size_t)blk * 64 * n;
            if (blk > 0) CTM_TRY(project_out(ctx, Wb, 64, n, W, blk * 64, Gp, 1));
            CTM_TRY(chol_pass(Wb, 0));
            if (!full) continue;
            if (blk > 0) CTM_TRY(project_out(ctx, Wb, 64, n, W, blk * 64, Gp, 1));
            CTM_TRY(chol_pass(Wb, 1));
            CTM_TRY(chol_pass(Wb, 2));
        }
        return CTM_OK;
    };
    // Where to look first: `moved` = |R|_F / |l0| of the previous subspace on this matrix (eigh_warm_verify), the residual contracts by
    // roughly |lambda_{p+1} / lambda_kk| per application -- 0.005 .. 0.01 measured on the C4v corner; 0.01 assumed until a look has
    // measured it -- so a subspace that moved by 4e-3 is looked at after six applications instead of after four AND six (a
    // Rayleigh-Ritz costs as much as three applications).  Never before the fourth application (the `sound` rule).
    const int min_rr = 3, max_it = std::max(min_rr, ctx->eigh_orth_max);
    const double tol = resid_tol(ctx, n);
    int next_rr = min_rr;
    double ref_val = moved; int ref_it = moved > 0.0 ? -1 : -2;       // (-2: nothing to measure the contraction against yet)
    int looks = 0;
    if (moved > 0.0 && ctx->eigh_orth_predict) {
        // (the contraction the previous accepted solve of this context saw from its own `moved` to its accepted residual -- a property
        //  of the spectrum, which changes slowly from sweep to sweep -- places the first look better than the fixed guess: a signed
        //  random C4v state contracts by 0.25-0.33 per application, not 0.01, and paid three looks per solve)
        const double prev_rate = orth_state_get(ctx).rate;
        const double rho = prev_rate > 0.0 ? prev_rate : 1e-2;
        const int need = (int)std::ceil(std::log(0.5 * tol / std::min(moved, 1.0)) / std::log(rho) + (prev_rate > 0.0 ? 0.5 : 0.0));      // applications
        next_rr = std::min(max_it, std::max(min_rr, need - 1));
    }
    for (int it = 0; it <= max_it; ++it) {
        CTM_TRY(rows_times(ctx, Q, n, p, n, n, As, false, Y, n));       // Y = Q A: application it + 1
        if (dbl && it + 2 <= next_rr) {                                  // (no look before application it + 3)
            CTM_TRY(rows_times(ctx, Y, n, p, n, n, As, false, Z2, n));  // application it + 2
            if (dbl_shift > 0.0) CTM_LAUNCH(ctx, axpy_kernel, dim3(1024), dim3(256), 0, Z2, (const double*)Q, -dbl_shift, (size_t)p * n);
            CTM_TRY(orth(Z2, it + 2 >= next_rr));
            std::swap(Q, Z2);
            ctx->eigh_orth_doubled += 1;
            it += 1;
            continue;
        }
        if (it >= next_rr || it == max_it) {
            GemmDesc gt; gt.M = p; gt.N = p; gt.K = n; gt.A = Y; gt.sam = n; gt.sak = 1; gt.B = Q; gt.sbk = 1; gt.sbn = n; gt.C = T; gt.ldc = p;
            CTM_TRY(gemm_f64(ctx, gt));                                  // T = Y Q^T
            GemmDesc ge; ge.M = p; ge.N = p; ge.K = n; ge.A = Q; ge.sam = n; ge.sak = 1; ge.B = Q; ge.sbk = 1; ge.sbn = n; ge.C = E; ge.ldc = p;
            CTM_TRY(gemm_f64(ctx, ge));
            CTM_LAUNCH(ctx, sub_eye_kernel, dim3((p + 255) / 256), dim3(256), 0, E, p);
            CTM_TRY(row_norms(ctx, E, p, p, p, norms));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(he.data(), norms, sizeof(double) * p, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            const double dev = *std::max_element(he.begin(), he.end());
            if (!(dev <= 1e-12)) {          // (NaN included: a numerically rank-deficient block breaks the Cholesky steps) -- not a case for this route
                if (ctx->jacobi_verbose) fprintf(stderr, "[eigh-orth] n=%d p=%d application %d: |QQ^T - I| = %.2e, leaving\n", n, p, it + 1, dev);
                ctx->eigh_orth_fails += 1;
                return CTM_OK;
            }
            const bool save = ctx->si_enable; ctx->si_enable = false;
            ctx->jacobi_quad_exit = ctx->eigh_orth_quad_exit;                   // (the residual test below certifies what this returns)
            const int st = jacobi_eigh_top(ctx, T, p, p, Dp, Zt, nullptr);      // rows of Zt = eigenvectors, ordered by |theta| (synchronises)
            ctx->jacobi_quad_exit = 0.0;
            ctx->si_enable = save;
            CTM_TRY(st);
            GemmDesc r1; r1.M = kk; r1.N = n; r1.K = p; r1.A = Zt; r1.sam = p; r1.sak = 1; r1.B = Q; r1.sbk = n; r1.sbn = 1; r1.C = X; r1.ldc = n;
            CTM_TRY(gemm_f64(ctx, r1));
            GemmDesc r2 = r1; r2.B = Y; r2.C = AX;
            CTM_TRY(gemm_f64(ctx, r2));
            CTM_LAUNCH(ctx, resid_rows_kernel, dim3((kk + 3) / 4), dim3(256), 0, (const double*)AX, (long long)n, (const double*)X, (long long)n,
                       (const double*)Dp, kk, n, res);
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), res, sizeof(double) * kk, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(hd.data(), Dp, sizeof(double) * p, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            const double lam0 = std::fabs(hd[0]), lamk = std::fabs(hd[kk - 1]);
            const double worst = *std::max_element(h.begin(), h.begin() + kk);
            if (ctx->jacobi_verbose) fprintf(stderr, "[eigh-orth] n=%d p=%d application %d: |QQ^T - I| = %.2e  max resid/|l0| = %.3e  |l_kk|/|l0| = %.3e  |l_p|/|l0| = %.3e  double=%d shift/l_kk^2=%.3f\n", n, p, it + 1, dev, worst / std::max(lam0, 1e-300), lamk / std::max(lam0, 1e-300), std::fabs(hd[p - 1]) / std::max(lam0, 1e-300), (int)dbl, dbl_shift / std::max(lamk * lamk, 1e-300));
            if (!(lam0 > 0.0) || !(lamk > ctx->rank_tol * lam0)) return CTM_OK;
            if (worst <= tol * lam0) {
                // gauge: the sign of <x_i, v_i> (previous vectors).  (Rotating whole multiplets onto the previous vectors -- orthogonal
                // Procrustes per cluster of equal |theta|, +-lambda eigenspaces matched by weight -- was tried for the SU(2) multiplets of
                // the RVB state: the movement measure of the next call drops 2-4x, the number of applications does not: not kept.)
                CTM_TRY(row_dots(ctx, X, warm, kk, n, n, dots));
                CTM_LAUNCH(ctx, signed_rows_kernel, dim3(256), dim3(256), 0, (const double*)X, (const double*)dots, 1, kk, n, AX);     // aligned with the previous vectors
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(warm, AX, sizeof(double) * (size_t)kk * n, hipMemcpyDeviceToDevice, ctx->stream));
                CTM_LAUNCH(ctx, signed_rows_kernel, dim3(256), dim3(256), 0, (const double*)AX, (const double*)Dp, 1, k_out, n, Ut);
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(D, Dp, sizeof(double) * k_out, hipMemcpyDeviceToDevice, ctx->stream));
                ctx->si_hits += 1; ctx->eigh_orth_hits += 1;
                {
                    OrthState os = orth_state_get(ctx);
                    os.backoff = 0;
                    os.theta_0 = lam0; os.theta_k = lamk;
                    if (moved > 0.0) os.rate = std::min(0.9, std::max(1e-3, std::pow(std::max(worst / lam0, 1e-16) / std::min(moved, 1.0), 1.0 / (it + 1))));
                    if (moved > 0.0 && dbl_shift == 0.0) os.c_ratio = os.rate;
                    CTM_TRY(orth_state_put(ctx, state_hdr, os));
                }
                ctx->si_last_iters = it + 1; ctx->si_total_iters += it + 1;
                *accepted = true;
                return CTM_OK;
            }
            // next look: from the contraction measured so far (two applications when there is nothing to measure it against)
            int need = 2;
            const double ref = ref_it >= -1 ? ref_val : 0.0;          // residual level `it - ref_it` applications ago
            if (ref > 0.0 && ctx->eigh_orth_predict) {
                // (a flat spectrum behind the kept pairs -- |lambda_{p+1} / lambda_kk| close to 1 -- is not a case for this route: leave at
                //  the first look that can tell, and keep away from it for a growing number of calls)
                const double rate = std::min(0.999, std::max(1e-3, std::pow(worst / (ref * lam0), 1.0 / (it - ref_it))));
                const double needd = std::log(0.5 * tol * lam0 / worst) / std::log(rate);
                if (!(needd <= (double)(max_it - it))) {
                    if (ctx->jacobi_verbose) fprintf(stderr, "[eigh-orth] contraction %.3f per application: %.0f more needed, leaving\n", rate, needd);
                    ctx->eigh_orth_fails += 1;
                    {
                        OrthState os = orth_state_get(ctx);
                        os.backoff = std::min(64, std::max(2, 2 * os.backoff));
                        os.skip = os.backoff;
                        CTM_TRY(orth_state_put(ctx, state_hdr, os));
                    }
                    return CTM_OK;
                }
                need = std::max(1, std::min((int)std::ceil(needd), looks >= 1 ? 12 : 6));      // (the first estimate includes the fast initial drop)
            }
            looks += 1;
            ref_val = worst / lam0; ref_it = it;
            next_rr = std::min(it + need, max_it);
        }
        if (it == max_it) break;
        CTM_TRY(orth(Y, it + 1 >= next_rr));            // the basis a Rayleigh-Ritz may use gets the full passes
        std::swap(Q, Y);
    }
    ctx->eigh_orth_fails += 1;
    return CTM_OK;
}

int jacobi_eigh_top(ctm_ctx* ctx, const double* A, int n, int k, double* D, double* Ut, double* warm) {
    if (n <= 0 || k <= 0 || k > n) { ctx->set_error("jacobi_eigh_top: bad n/k"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    double* As;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * n, (void**)&As));
    CTM_TRY(symmetrize_lower(ctx, A, As, n, 0.0));
    // (1) large problems with k << n: leading |lambda| invariant subspace by the SVD iteration on the symmetric matrix,
    //     then a small symmetric Rayleigh-Ritz on it.
    if (ctx->si_enable && k < n && n >= ctx->si_min_n) {
        // a few extra vectors so that a cluster of equal |lambda| with both signs is never cut inside the RR space
        const int kk = std::min(n, k + 8), k_out = k;
        if (warm && ctx->eigh_warm) {
            bool accepted = false;
            bool norms_ok = false;
            double moved = 0.0;
            double* state_hdr = (n >= 8) ? warm + (size_t)kk * n : nullptr;      // header row behind the kk vectors (include/ctm_hip.h)
            CTM_TRY(eigh_warm_verify(ctx, As, n, kk, k_out, warm, D, Ut, &accepted, false, &norms_ok, &moved, state_hdr));
            if (accepted) return CTM_OK;
            OrthState os = ctx->eigh_orth_iter ? orth_state_get(ctx) : OrthState();
            if (ctx->eigh_orth_iter && os.skip > 0) { os.skip -= 1; CTM_TRY(orth_state_put(ctx, state_hdr, os)); }
            else if (ctx->eigh_orth_iter) {
                CTM_TRY(eigh_orth_iter(ctx, As, n, kk, k_out, warm, D, Ut, &accepted, norms_ok, moved, state_hdr));
                if (accepted) return CTM_OK;
            }
        }
        double *S, *Uk, *Vk;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * kk, (void**)&S));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * n, (void**)&Uk));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)kk * n, (void**)&Vk));
        bool ok = false;
        MatOp aop; aop.n = n; aop.M = As; aop.warm = warm;     // warm: (k + 8) x n rows of the previous invariant subspace
        CTM_TRY(svd_iter(ctx, aop, kk, S, Uk, Vk, &ok));
        if (ok) {
            const int k = kk;
            ctx->si_hits += 1;
            // T = U A U^T (k x k, symmetric, diagonal except inside clusters of equal |lambda|)
            double *Y, *T, *Dk, *Th;
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Y));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * k, (void**)&T));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&Dk));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * k, (void**)&Th));
            GemmDesc g; g.M = k; g.N = n; g.K = n; g.A = Uk; g.sam = n; g.sak = 1; g.B = As; g.sbk = n; g.sbn = 1; g.C = Y; g.ldc = n;
            CTM_TRY(gemm_f64(ctx, g));
            GemmDesc t; t.M = k; t.N = k; t.K = n; t.A = Y; t.sam = n; t.sak = 1; t.B = Uk; t.sbk = 1; t.sbn = n; t.C = T; t.ldc = k;
            CTM_TRY(gemm_f64(ctx, t));
            const bool save = ctx->si_enable; ctx->si_enable = false;
            const int st = jacobi_eigh_top(ctx, T, k, k, Dk, Th, nullptr);      // full small problem (rows of Th = eigenvectors)
            ctx->si_enable = save;
            CTM_TRY(st);
            // eigen-pairs of T come ordered by |lambda|: keep the leading k_out.  The workspace keeps ALL kk eigenvectors (after this
            // Rayleigh-Ritz: inside a cluster of equal |lambda| with both signs the singular vectors of the iteration are mixtures),
            // as v_i = sign(lambda_i) u_i -- the right vectors the iteration would have kept
            GemmDesc r; r.M = k; r.N = n; r.K = k; r.A = Th; r.sam = k; r.sak = 1; r.B = Uk; r.sbk = n; r.sbn = 1; r.C = Y; r.ldc = n;
            CTM_TRY(gemm_f64(ctx, r));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Ut, Y, sizeof(double) * (size_t)k_out * n, hipMemcpyDeviceToDevice, ctx->stream));
            if (warm) CTM_LAUNCH(ctx, signed_rows_kernel, dim3(256), dim3(256), 0, (const double*)Y, (const double*)Dk, 1, k, n, warm);
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(D, Dk, sizeof(double) * k_out, hipMemcpyDeviceToDevice, ctx->stream));
            return CTM_OK;
        }
        ctx->si_fallbacks += 1;
    }
    // (2) full path: one-sided Jacobi on A + shift*I (positive definite)
    const int b = choose_block(ctx, n), np = padded(n, b);
    const long long ld = (long long)n + np;
    double *X, *norms;
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)np * ld, (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * np, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * np, (void**)&d_idx));
    std::vector<double> h;
    int st;
    const double fro = host_fro(ctx, As, n, n, n, norms, h, &st);     // >= spectral norm
    CTM_TRY(st);
    const double shift = fro * 1.0009765625 + 1e-300;
    CTM_TRY(symmetrize_lower(ctx, A, As, n, shift));
    // Warm start of the FULL decomposition (k == n, the differentiable route of an optimisation: the same matrix comes back, slightly
    // changed, epoch after epoch): with W = the previous eigenvector rows, the rows of W (A + shift I) are already almost
    // orthogonal (off-diagonal Gram entries of the size of the change), so the sweeps start in the quadratically convergent regime.
    // Any orthonormal W is a valid start -- the result does not depend on it beyond rounding.
    bool warm_full = false;
    if (warm && k == n && ctx->eigh_warm) {
        int st2;
        std::vector<double> hw;
        const double fw = host_fro(ctx, warm, n, n, n, norms, hw, &st2);
        CTM_TRY(st2);
        warm_full = std::fabs(fw - std::sqrt((double)n)) <= 1e-6 * std::sqrt((double)n);
        for (int i = 0; warm_full && i < n; ++i) warm_full = std::fabs(hw[i] - 1.0) <= 1e-6;
    }
    if (warm_full) {
        CTM_TRY(fill_f64(ctx, X, (size_t)np * ld, 0.0));
        GemmDesc gw; gw.M = n; gw.N = n; gw.K = n; gw.A = warm; gw.sam = n; gw.sak = 1; gw.B = As; gw.sbk = n; gw.sbn = 1; gw.C = X; gw.ldc = ld;
        CTM_TRY(gemm_f64(ctx, gw));
        CTM_TRY(copy2d(ctx, warm, n, X + n, ld, n, n));
        ctx->eigh_warm_hits += 1;
    } else
        CTM_LAUNCH(ctx, fill_wq_kernel, dim3(2048), dim3(256), 0, As, n, n, (long long)n, X, np, ld, 1);
    CTM_TRY(jacobi_rows(ctx, X, np, ld, n, (int)ld, b, 0, shift * std::sqrt((double)n), ctx->jacobi_max_sweeps));
    CTM_TRY(row_norms(ctx, X, np, n, ld, norms));
    h.resize(np);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * np, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // the n genuine rows have norm >= shift - fro > 0; padded rows are exactly zero
    std::vector<int> idx;
    for (int i = 0; i < np; ++i) if (h[i] > 0.0) idx.push_back(i);
    if ((int)idx.size() != n) { ctx->set_error("jacobi_eigh_top: rank bookkeeping failed"); return CTM_ERR_NOCONV; }
    std::vector<double> lam(np);
    for (int i : idx) lam[i] = h[i] - shift;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return std::fabs(lam[a]) > std::fabs(lam[c]); });
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, idx.data(), sizeof(int) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    CTM_TRY(gather_rows(ctx, X + n, ld, d_idx, k, n, Ut, n, nullptr));
    CTM_TRY(reorth_rows(ctx, Ut, k, n, n, 2));
    // eigenvalues as Rayleigh quotients u^T A u (drift-free, |error| = O(eps |A|))
    double* Y;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Y));
    CTM_TRY(symmetrize_lower(ctx, A, As, n, 0.0));
    GemmDesc g; g.M = k; g.N = n; g.K = n; g.A = Ut; g.sam = n; g.sak = 1; g.B = As; g.sbk = n; g.sbn = 1; g.C = Y; g.ldc = n;
    CTM_TRY(gemm_f64(ctx, g));
    CTM_TRY(row_dots(ctx, Y, Ut, k, n, n, D));
    if (warm && k == n) CTM_HIP_CHECK(ctx, hipMemcpyAsync(warm, Ut, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream));
    return CTM_OK;
}

// Complex Hermitian twin of jacobi_eigh_top (eig_sym.py:25-34 on a complex128 matrix: torch.linalg.eigh, lower triangle, ordered
// by |lambda| descending).  (1) large n, k << n: leading-|lambda| invariant subspace by the complex block iteration on the
// Hermitian matrix, then a small Hermitian Rayleigh-Ritz; (2) full path: one-sided complex Jacobi on A + shift I (positive
// definite, so the accumulated unitary holds the eigenvectors and lambda = sigma - shift).
int jacobi_eigh_top_c(ctm_ctx* ctx, const double* Ar, const double* Ai, int n, int k, double* D, double* Ut, double* warm) {
    if (n <= 0 || k <= 0 || k > n) { ctx->set_error("jacobi_eigh_top_c: bad n/k"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    const size_t nn = (size_t)n * n;
    double* As;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * nn, (void**)&As));
    CTM_TRY(hermitize_lower_c128(ctx, Ar, Ai, As, As + nn, n, 0.0));
    if (ctx->si_enable && k < n && n >= ctx->si_min_n) {
        const int kk = std::min(n, k + 8);
        const size_t kn = (size_t)kk * n;
        double *S, *Uk, *Vk;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * kk, (void**)&S));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * kn, (void**)&Uk));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * kn, (void**)&Vk));
        if (warm && ctx->eigh_warm) {
            bool accepted = false;
            CTM_TRY(eigh_warm_verify_c(ctx, As, As + nn, n, kk, k, warm, D, Ut, &accepted));
            if (accepted) return CTM_OK;
        }
        bool ok = false;
        MatOp aop; aop.n = n; aop.M = As; aop.Mi = As + nn; aop.warm = warm;      // warm: planar (k + 8) x n rows (re plane, im plane) of the previous subspace
        CTM_TRY(svd_iter_c(ctx, aop, kk, S, Uk, Vk, &ok));
        if (ok) {
            ctx->si_hits += 1;
            // T = U A U^H (kk x kk Hermitian; U rows are q_j^H), T w = mu w, eigenvector rows x^H = w^H U
            double *Y, *T, *Dk, *Th;
            CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * kn, (void**)&Y));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)kk * kk, (void**)&T));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * kk, (void**)&Dk));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)kk * kk, (void**)&Th));
            XM u{Uk, Uk + kn, n, false, false}, a{As, As + nn, n, false, false}, uh{Uk, Uk + kn, n, true, true};
            CTM_TRY(xgemm(ctx, kk, n, n, u, a, Y, Y + kn, n));
            XM y{Y, Y + kn, n, false, false};
            CTM_TRY(xgemm(ctx, kk, kk, n, y, uh, T, T + (size_t)kk * kk, kk));
            const bool save = ctx->si_enable; ctx->si_enable = false;
            const int st = jacobi_eigh_top_c(ctx, T, T + (size_t)kk * kk, kk, kk, Dk, Th, nullptr);
            ctx->si_enable = save;
            CTM_TRY(st);
            XM th{Th, Th + (size_t)kk * kk, kk, false, false};
            // the leading k rows of Th (ordered by |mu|) times U: planar output with k rows
            double* tmp;
            CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * kn, (void**)&tmp));
            CTM_TRY(xgemm(ctx, kk, n, kk, th, u, tmp, tmp + kn, n));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Ut, tmp, sizeof(double) * (size_t)k * n, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Ut + (size_t)k * n, tmp + kn, sizeof(double) * (size_t)k * n, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(D, Dk, sizeof(double) * k, hipMemcpyDeviceToDevice, ctx->stream));
            if (warm) {     // all kk eigenvectors after the Rayleigh-Ritz, v_i = sign(lambda_i) u_i (see jacobi_eigh_top)
                CTM_LAUNCH(ctx, signed_rows_kernel, dim3(256), dim3(256), 0, (const double*)tmp, (const double*)Dk, 1, kk, n, warm);
                CTM_LAUNCH(ctx, signed_rows_kernel, dim3(256), dim3(256), 0, (const double*)(tmp + kn), (const double*)Dk, 1, kk, n, warm + kn);
            }
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            return CTM_OK;
        }
        ctx->si_fallbacks += 1;
    }
    const int np = padded(n, BC);
    const long long ld = (long long)n + np;
    double *X, *norms;
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)2 * np * ld, (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * np, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * 2 * np, (void**)&d_idx));
    std::vector<double> h;
    int st;
    const double fro = host_fro(ctx, As, 2 * n, n, n, norms, h, &st);      // both planes: |A|_F >= spectral norm
    CTM_TRY(st);
    const double shift = fro * 1.0009765625 + 1e-300;
    CTM_TRY(hermitize_lower_c128(ctx, Ar, Ai, As, As + nn, n, shift));
    bool warm_full = false;          // warm start of the full decomposition: see jacobi_eigh_top()
    if (warm && k == n && ctx->eigh_warm) {
        int st2;
        std::vector<double> hw;
        const double fw = host_fro(ctx, warm, 2 * n, n, n, norms, hw, &st2);
        CTM_TRY(st2);
        warm_full = std::fabs(fw - std::sqrt((double)n)) <= 1e-6 * std::sqrt((double)n);
        for (int i = 0; warm_full && i < n; ++i) warm_full = std::fabs(std::sqrt(hw[i] * hw[i] + hw[n + i] * hw[n + i]) - 1.0) <= 1e-6;
    }
    if (warm_full) {
        ArenaScope ws(ctx);
        double* Yw;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * nn, (void**)&Yw));
        XM w{warm, warm + nn, n, false, false}, a{As, As + nn, n, false, false};
        CTM_TRY(xgemm(ctx, n, n, n, w, a, Yw, Yw + nn, n));                       // rows u^H (A + shift I)
        CTM_LAUNCH(ctx, fill_wq_c2_kernel, dim3(2048), dim3(256), 0, (const double*)Yw, (const double*)(Yw + nn), (const double*)warm,
                   (const double*)(warm + nn), n, X, np, ld);
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));                     // Yw is released with the scope
        ctx->eigh_warm_hits += 1;
    } else
        CTM_LAUNCH(ctx, fill_wq_c_kernel, dim3(2048), dim3(256), 0, (const double*)As, (const double*)(As + nn), n, X, np, ld, 1);
    CTM_TRY(jacobi_rows(ctx, X, 2 * np, ld, n, (int)ld, 2 * BC, 0, shift * std::sqrt((double)n), ctx->jacobi_max_sweeps, true));
    std::vector<double> hc;
    CTM_TRY(panel_row_norms(ctx, X, np, n, ld, norms, hc));
    std::vector<int> idx;
    for (int i = 0; i < np; ++i) if (hc[i] > 0.0) idx.push_back(i);
    if ((int)idx.size() != n) { ctx->set_error("jacobi_eigh_top_c: rank bookkeeping failed"); return CTM_ERR_NOCONV; }
    std::vector<double> lam(np);
    for (int i : idx) lam[i] = hc[i] - shift;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return std::fabs(lam[a]) > std::fabs(lam[c]); });
    CTM_TRY(panel_gather(ctx, X + n, ld, idx, k, n, Ut, d_idx));
    CTM_TRY(reorth_rows_c(ctx, Ut, k, n, 2));
    // eigenvalues as Rayleigh quotients Re(u^H A u): rows r = u^H, (r A) . conj(r)
    const size_t kn = (size_t)k * n;
    double *Y, *d2;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * kn, (void**)&Y));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&d2));
    CTM_TRY(hermitize_lower_c128(ctx, Ar, Ai, As, As + nn, n, 0.0));
    XM u{Ut, Ut + kn, n, false, false}, a{As, As + nn, n, false, false};
    CTM_TRY(xgemm(ctx, k, n, n, u, a, Y, Y + kn, n));
    CTM_TRY(row_dots(ctx, Y, Ut, k, n, n, D));
    CTM_TRY(row_dots(ctx, Y + kn, Ut + kn, k, n, n, d2));
    CTM_LAUNCH(ctx, add_inplace_kernel, dim3((k + 255) / 256), dim3(256), 0, D, (const double*)d2, (size_t)k);
    if (warm && k == n) CTM_HIP_CHECK(ctx, hipMemcpyAsync(warm, Ut, sizeof(double) * 2 * nn, hipMemcpyDeviceToDevice, ctx->stream));
    return CTM_OK;
}


