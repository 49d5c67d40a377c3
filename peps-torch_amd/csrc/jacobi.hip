// chi-truncation engine: one-sided block-Jacobi SVD / symmetric eigensolver for gfx950.
//
// Replaces torch.linalg.svd / torch.linalg.eigh at linalg/svd_gesdd.py:91 and linalg/eig_sym.py:25
// of the reference.  LAPACK's bidiagonalisation is BLAS-2 bound and maps
// badly onto the matrix cores; here the O(n^3) work is two batched FP64-MFMA GEMMs per round:
//
//   rows of W (n x n, initially M) are orthogonalised in place:  Q M = Sigma V^T
//   round-robin over pairs (i,j) of row panels of b rows (all n/(2b) disjoint pairs of a round are one
//   batched launch):
//     1. G_ij = [W_i;W_j] [W_i;W_j]^T            (2b x 2b Gram, batched GEMM, K = n)
//     2. G_ij = J diag J^T                        (small two-sided Jacobi in LDS, one workgroup per pair,
//                                                  eigenvalues sorted descending -> de Rijk-like ordering)
//     3. [W_i;W_j] <- J^T [W_i;W_j],  [Q_i;Q_j] <- J^T [Q_i;Q_j]   (batched GEMM into the ping-pong buffer)
//   until every Gram matrix is diagonal to tolerance.  Then sigma_k = |W_k|, v_k = W_k/sigma_k, u_k = Q_k.
//
// U is orthogonal by construction; V (normalised rows) is re-orthonormalised against the LARGER
// triplets only (triangular first-order inverse-Cholesky correction, GEMM-only) so that the result has
// the LAPACK structure: U, V orthonormal to eps, |M - U S V^T| = O(eps |M|).
// The symmetric eigenproblem runs the same machinery on A + shift*I (positive definite, so the
// right-rotation factor IS the eigenvector matrix and lambda = sigma - shift).
#include "ctm_common.h"
#include <algorithm>
#include <cmath>
#include <numeric>

namespace {

constexpr int MAXM = 64;   // largest pair-Gram the LDS solver handles (2 * block)

// ---------------------------------------------------------------------------------------------
// batched small symmetric eigensolver: one workgroup (256 threads) per m x m Gram matrix
// ---------------------------------------------------------------------------------------------
struct SmallEigParams {
    const double* G;     // batch x m x m
    double* J;           // batch x m x m (columns = eigenvectors, eigenvalues descending)
    int m;
    double tol;          // relative off-diagonal tolerance for a rotation
    int max_sweeps;
    double tau2;         // scale floor: a pair (i,j) is measured against max(sqrt(g_ii g_jj), tau2)
    unsigned long long* stat_rel;   // max |g_ij| / max(sqrt(g_ii g_jj), tau2)  (bits of a non-negative double)
    unsigned long long* stat_abs;   // max |g_ij| / sqrt(g_ii g_jj) (classical measure, diagnostics only)
    int* flags;          // per pair: 1 if J != I (the apply GEMM skips the others)
};

__global__ __launch_bounds__(256) void small_eig_kernel(SmallEigParams p) {
    __shared__ double W[MAXM][MAXM + 1];
    __shared__ double Jm[MAXM][MAXM + 1];
    __shared__ double cs_c[MAXM / 2], cs_s[MAXM / 2];
    __shared__ int pr_p[MAXM / 2], pr_q[MAXM / 2];
    __shared__ double red[4];
    __shared__ int rot_flag;
    __shared__ int rank_of[MAXM];

    const int m = p.m, tid = threadIdx.x;
    const double* G = p.G + (size_t)blockIdx.x * m * m;
    double* Jout = p.J + (size_t)blockIdx.x * m * m;

    for (int q = tid; q < m * m; q += 256) {
        const int r = q / m, c = q - r * m;
        W[r][c] = G[q];
        Jm[r][c] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();

    // convergence statistics of the INCOMING Gram matrix
    {
        double srel = 0.0, sabs = 0.0;
        for (int q = tid; q < m * m; q += 256) {
            const int r = q / m, c = q - r * m;
            if (r < c) {
                const double g = fabs(W[r][c]), a = W[r][r], b = W[c][c];
                if (g > 0.0) {
                    const double sc = sqrt(fabs(a * b));
                    srel = fmax(srel, g / fmax(sc, p.tau2));
                    if (sc > 0.0) sabs = fmax(sabs, g / sc);
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            srel = fmax(srel, __shfl_down(srel, off, 64));
            sabs = fmax(sabs, __shfl_down(sabs, off, 64));
        }
        if ((tid & 63) == 0) red[tid >> 6] = srel;
        __syncthreads();
        if (tid == 0) {
            double v = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
            atomicMax(p.stat_rel, (unsigned long long)__double_as_longlong(v));
            red[0] = v;
        }
        __syncthreads();
        srel = red[0];
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = sabs;
        __syncthreads();
        if (tid == 0) {
            double v = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
            atomicMax(p.stat_abs, (unsigned long long)__double_as_longlong(v));
        }
        __syncthreads();
        if (srel <= p.tol) {
            // already diagonal to tolerance: rows stay untouched, the apply GEMM skips this pair
            if (tid == 0) p.flags[blockIdx.x] = 0;
            return;
        }
        if (tid == 0) p.flags[blockIdx.x] = 1;
    }

    const int half = m / 2, mm1 = m - 1;
    for (int sweep = 0; sweep < p.max_sweeps; ++sweep) {
        if (tid == 0) rot_flag = 0;
        __syncthreads();
        for (int r = 0; r < mm1; ++r) {
            if (tid < half) {
                int pi, qi;
                if (tid == 0) { pi = mm1; qi = r % mm1; }
                else { pi = (r + tid) % mm1; qi = (r - tid + mm1) % mm1; }
                if (pi > qi) { const int t = pi; pi = qi; qi = t; }
                const double a = W[pi][pi], b = W[qi][qi], g = W[pi][qi];
                double c = 1.0, s = 0.0;
                if (g != 0.0 && fabs(g) > p.tol * fmax(sqrt(fabs(a * b)), p.tau2)) {
                    const double zeta = (b - a) / (2.0 * g);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                    rot_flag = 1;
                }
                cs_c[tid] = c; cs_s[tid] = s; pr_p[tid] = pi; pr_q[tid] = qi;
            }
            __syncthreads();
            // column update of W and J:  [x_p x_q] <- [c x_p - s x_q,  s x_p + c x_q]
            for (int q = tid; q < half * m; q += 256) {
                const int k = q / m, i = q - k * m;
                const double c = cs_c[k], s = cs_s[k];
                if (s != 0.0) {
                    const int pi = pr_p[k], qi = pr_q[k];
                    const double wp = W[i][pi], wq = W[i][qi];
                    W[i][pi] = c * wp - s * wq; W[i][qi] = s * wp + c * wq;
                    const double jp = Jm[i][pi], jq = Jm[i][qi];
                    Jm[i][pi] = c * jp - s * jq; Jm[i][qi] = s * jp + c * jq;
                }
            }
            __syncthreads();
            // row update of W
            for (int q = tid; q < half * m; q += 256) {
                const int k = q / m, j = q - k * m;
                const double c = cs_c[k], s = cs_s[k];
                if (s != 0.0) {
                    const int pi = pr_p[k], qi = pr_q[k];
                    const double wp = W[pi][j], wq = W[qi][j];
                    W[pi][j] = c * wp - s * wq; W[qi][j] = s * wp + c * wq;
                }
            }
            __syncthreads();
        }
        if (rot_flag == 0) break;
        __syncthreads();
    }

    // sort eigenvalues (diagonal) descending; ties broken by index
    if (tid < m) {
        const double d = W[tid][tid];
        int rk = 0;
        for (int j = 0; j < m; ++j) {
            const double dj = W[j][j];
            rk += (dj > d) || (dj == d && j < tid);
        }
        rank_of[tid] = rk;
    }
    __syncthreads();
    for (int q = tid; q < m * m; q += 256) {
        const int r = q / m, c = q - r * m;
        Jout[r * m + rank_of[c]] = Jm[r][c];
    }
}

// build the per-round batched-GEMM offset tables on the host (uploaded once per problem size)
struct RRTables {
    int nbk = 0, np = 0, b = 0;
    GemmOff* d_gram = nullptr;    // [rounds][pairs]
    GemmOff* d_apply = nullptr;   // [rounds][2*pairs]   (W panels then Q panels)
};

std::map<long long, RRTables>& tables() { static std::map<long long, RRTables> t; return t; }

int get_tables(ctm_ctx* ctx, int nbk, int np, int b, RRTables** out) {
    const long long key = ((long long)nbk << 40) ^ ((long long)np << 8) ^ b ^ ((long long)ctx->device << 60);
    auto& T = tables();
    auto it = T.find(key);
    if (it != T.end()) { *out = &it->second; return CTM_OK; }
    const int rounds = nbk - 1, pairs = nbk / 2, m = 2 * b;
    std::vector<GemmOff> gram((size_t)rounds * pairs), app((size_t)rounds * 2 * pairs);
    const long long ld = np, qbase = (long long)np * np;
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < pairs; ++k) {
            int i, j;
            if (k == 0) { i = nbk - 1; j = r % (nbk - 1); }
            else { i = (r + k) % (nbk - 1); j = (r - k + (nbk - 1)) % (nbk - 1); }
            if (i > j) std::swap(i, j);
            const long long oi = (long long)i * b * ld, oj = (long long)j * b * ld;
            GemmOff g; g.a0 = oi; g.a1 = oj; g.b0 = oi; g.b1 = oj; g.c0 = g.c1 = (long long)k * m * m;
            gram[(size_t)r * pairs + k] = g;
            GemmOff a; a.a0 = a.a1 = (long long)k * m * m; a.b0 = oi; a.b1 = oj; a.c0 = oi; a.c1 = oj;
            app[(size_t)r * 2 * pairs + k] = a;
            a.b0 += qbase; a.b1 += qbase; a.c0 += qbase; a.c1 += qbase;
            app[(size_t)r * 2 * pairs + pairs + k] = a;
        }
    }
    RRTables t; t.nbk = nbk; t.np = np; t.b = b;
    if (hipMalloc(&t.d_gram, gram.size() * sizeof(GemmOff)) != hipSuccess ||
        hipMalloc(&t.d_apply, app.size() * sizeof(GemmOff)) != hipSuccess) {
        ctx->set_error("jacobi: table alloc"); return CTM_ERR_NOMEM;
    }
    CTM_HIP_CHECK(ctx, hipMemcpy(t.d_gram, gram.data(), gram.size() * sizeof(GemmOff), hipMemcpyHostToDevice));
    CTM_HIP_CHECK(ctx, hipMemcpy(t.d_apply, app.data(), app.size() * sizeof(GemmOff), hipMemcpyHostToDevice));
    T[key] = t;
    *out = &T[key];
    return CTM_OK;
}

// Core: X holds [W (np x np); Q (np x np)] and is transformed IN PLACE (each workgroup of the apply GEMM
// owns a column strip of all 2b rows of its pair and finishes reading it before it writes).
// ktop > 0: only the ktop largest rows need full relative accuracy -- pairs of smaller rows are measured
// against tau = (ktop-th largest row norm), which still bounds the spectral norm of the tail by tau(1+n tol).
int jacobi_core(ctm_ctx* ctx, double* X, int np, int b, bool with_q, int ktop, double fro) {
    const int nbk = np / b, rounds = nbk - 1, pairs = nbk / 2, m = 2 * b;
    RRTables* T;
    CTM_TRY(get_tables(ctx, nbk, np, b, &T));
    ArenaScope scope(ctx);
    double *G, *J, *norms;
    int* flags;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * pairs * m * m, (void**)&G));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * pairs * m * m, (void**)&J));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * np, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * pairs, (void**)&flags));
    unsigned long long* stat = (unsigned long long*)ctx->d_scratch;   // [0]=scaled, [1]=classical
    std::vector<double> h(np);
    const double floor2 = (1e-14 * fro) * (1e-14 * fro);
    ctx->last_sweeps = 0;
    for (int sweep = 0; sweep < ctx->jacobi_max_sweeps; ++sweep) {
        double tau2 = floor2;
        if (ktop > 0 && ktop < np) {
            CTM_TRY(row_norms(ctx, X, np, np, np, norms));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * np, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            std::nth_element(h.begin(), h.begin() + (ktop - 1), h.end(), std::greater<double>());
            tau2 = std::max(floor2, h[ktop - 1] * h[ktop - 1]);
        }
        CTM_HIP_CHECK(ctx, hipMemsetAsync(stat, 0, 2 * sizeof(double), ctx->stream));
        for (int r = 0; r < rounds; ++r) {
            GemmDesc g;
            g.M = m; g.N = m; g.K = np;
            g.A = X; g.sam = np; g.sak = 1; g.splitA = b;
            g.B = X; g.sbk = 1; g.sbn = np; g.splitB = b; g.splitB_dim = 2;
            g.C = G; g.ldc = m;
            g.batch = pairs; g.offs = T->d_gram + (size_t)r * pairs;
            CTM_TRY(gemm_f64(ctx, g));
            SmallEigParams sp;
            sp.G = G; sp.J = J; sp.m = m; sp.tol = ctx->jacobi_tol * 0.1; sp.max_sweeps = ctx->jacobi_inner_sweeps;
            sp.tau2 = tau2; sp.stat_rel = stat; sp.stat_abs = stat + 1; sp.flags = flags;
            hipLaunchKernelGGL(small_eig_kernel, dim3(pairs), dim3(256), 0, ctx->stream, sp);
            GemmDesc a;
            a.M = m; a.N = np; a.K = m;
            a.A = J; a.sam = 1; a.sak = m;                       // J^T
            a.B = X; a.sbk = np; a.sbn = 1; a.splitB = b; a.splitB_dim = 1;
            a.C = X; a.ldc = np; a.splitC = b;
            a.batch = pairs; a.offs = T->d_apply + (size_t)r * 2 * pairs; a.skip_flags = flags;
            CTM_TRY(gemm_f64(ctx, a));
            if (with_q) { a.offs = T->d_apply + (size_t)r * 2 * pairs + pairs; CTM_TRY(gemm_f64(ctx, a)); }
        }
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scratch, stat, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const double srel = ctx->h_scratch[0];
        ctx->last_sweeps = sweep + 1;
        ctx->last_offnorm = srel;
        if (ctx->jacobi_verbose) fprintf(stderr, "[jacobi] np=%d sweep %d  scaled=%.3e classical=%.3e tau=%.3e\n", np, sweep + 1, srel, ctx->h_scratch[1], std::sqrt(tau2));
        if (srel <= ctx->jacobi_tol) break;
    }
    ctx->total_sweeps += ctx->last_sweeps; ctx->jacobi_calls += 1;
    return CTM_OK;
}

int choose_block(ctm_ctx* ctx, int n) {
    int b = ctx->jacobi_block;
    if (b > MAXM / 2) b = MAXM / 2;
    if (n <= 64 && b > 16) b = 16;
    return b;
}

inline int padded(int n, int b) {
    int nbk = (n + b - 1) / b;
    if (nbk < 2) nbk = 2;
    if (nbk & 1) ++nbk;
    return nbk * b;
}

__global__ void pad_copy_kernel(const double* src, int n, double* dst, int np) {
    const size_t tot = (size_t)np * np;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const size_t r = q / np, c = q - r * np;
        dst[q] = (r < (size_t)n && c < (size_t)n) ? src[r * n + c] : 0.0;
    }
}

__global__ void scale_rows_kernel(double* x, int rows, int cols, const double* rs) {
    const size_t tot = (size_t)rows * cols;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) x[q] *= rs[q / cols];
}

__global__ void inv_or_zero_kernel(const double* s, double* out, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) out[i] = s[i] > 0.0 ? 1.0 / s[i] : 0.0;
}

// re-orthonormalise the rows of V (k x n) against the rows above them: V <- (I - tril(E) - diag(E)/2) V, E = V V^T - I
int reorth_rows(ctm_ctx* ctx, double* V, int k, int n, long long ld, int iters) {
    ArenaScope scope(ctx);
    double *E, *tmp;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k * k, (void**)&E));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&tmp));
    for (int it = 0; it < iters; ++it) {
        GemmDesc g;
        g.M = k; g.N = k; g.K = n; g.A = V; g.sam = ld; g.sak = 1; g.B = V; g.sbk = 1; g.sbn = ld; g.C = E; g.ldc = k;
        CTM_TRY(gemm_f64(ctx, g));
        CTM_TRY(tril_correction(ctx, E, k));
        GemmDesc a;
        a.M = k; a.N = n; a.K = k; a.A = E; a.sam = k; a.sak = 1; a.B = V; a.sbk = ld; a.sbn = 1; a.C = tmp; a.ldc = n;
        CTM_TRY(gemm_f64(ctx, a));
        CTM_TRY(copy2d(ctx, tmp, n, V, ld, k, n));
    }
    return CTM_OK;
}

}  // namespace

int jacobi_svd_top(ctm_ctx* ctx, const double* M, int n, int k, double* S, double* Ut, double* Vt) {
    if (n <= 0 || k <= 0 || k > n) { ctx->set_error("jacobi_svd_top: bad n/k"); return CTM_ERR_BADARG; }
    const int b = choose_block(ctx, n), np = padded(n, b);
    ArenaScope scope(ctx);
    double *X, *norms;
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)np * np, (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * np, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * np, (void**)&d_idx));
    hipLaunchKernelGGL(pad_copy_kernel, dim3(2048), dim3(256), 0, ctx->stream, M, n, X, np);
    const bool with_q = (Ut != nullptr);
    if (with_q) CTM_TRY(set_identity(ctx, X + (size_t)np * np, np, np));
    std::vector<double> h(np);
    CTM_TRY(row_norms(ctx, X, np, np, np, norms));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * np, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double fro = 0.0;
    for (int i = 0; i < np; ++i) fro += h[i] * h[i];
    fro = std::sqrt(fro);
    CTM_TRY(jacobi_core(ctx, X, np, b, with_q, (k < n) ? k : 0, fro));
    CTM_TRY(row_norms(ctx, X, np, np, np, norms));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * np, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> idx(np);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return h[a] > h[c]; });
    std::vector<double> hs(k);
    for (int i = 0; i < k; ++i) hs[i] = h[idx[i]];
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, idx.data(), sizeof(int) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, hs.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // host vectors go out of scope
    if (!Ut) {   // singular values only: row norms of the converged W
        return CTM_OK;
    }
    // U = accumulated rotations (orthonormalised against drift); then Sigma V^T = U^T M is recomputed by one
    // k x n x n GEMM so that S and V carry no accumulated rounding of the sweeps (|error| = O(eps |M|)).
    CTM_TRY(gather_rows(ctx, X + (size_t)np * np, np, d_idx, k, n, Ut, n, nullptr));
    CTM_TRY(reorth_rows(ctx, Ut, k, n, n, 2));
    if (Vt) {
        double* inv;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&inv));
        GemmDesc g; g.M = k; g.N = n; g.K = n; g.A = Ut; g.sam = n; g.sak = 1; g.B = M; g.sbk = n; g.sbn = 1; g.C = Vt; g.ldc = n;
        CTM_TRY(gemm_f64(ctx, g));
        CTM_TRY(row_norms(ctx, Vt, k, n, n, S));
        hipLaunchKernelGGL(inv_or_zero_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, S, inv, k);
        hipLaunchKernelGGL(scale_rows_kernel, dim3(1024), dim3(256), 0, ctx->stream, Vt, k, n, inv);
        CTM_TRY(reorth_rows(ctx, Vt, k, n, n, 2));
    }
    return CTM_OK;
}

int jacobi_svdvals(ctm_ctx* ctx, const double* M, int n, double* S) {
    return jacobi_svd_top(ctx, M, n, n, S, nullptr, nullptr);
}

int jacobi_eigh_top(ctm_ctx* ctx, const double* A, int n, int k, double* D, double* Ut) {
    if (n <= 0 || k <= 0 || k > n) { ctx->set_error("jacobi_eigh_top: bad n/k"); return CTM_ERR_BADARG; }
    const int b = choose_block(ctx, n), np = padded(n, b);
    ArenaScope scope(ctx);
    double *X, *norms, *As;
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)np * np, (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * n, (void**)&As));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * np, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * np, (void**)&d_idx));
    // shift = Frobenius norm of sym(lower(A)) >= spectral norm
    CTM_TRY(symmetrize_lower(ctx, A, As, n, 0.0));
    CTM_TRY(row_norms(ctx, As, n, n, n, norms));
    std::vector<double> h(np);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double fro = 0.0;
    for (int i = 0; i < n; ++i) fro += h[i] * h[i];
    fro = std::sqrt(fro);
    const double shift = fro * 1.0009765625 + 1e-300;
    CTM_TRY(symmetrize_lower(ctx, A, As, n, shift));
    hipLaunchKernelGGL(pad_copy_kernel, dim3(2048), dim3(256), 0, ctx->stream, As, n, X, np);
    CTM_TRY(set_identity(ctx, X + (size_t)np * np, np, np));
    CTM_TRY(jacobi_core(ctx, X, np, b, true, 0, shift * std::sqrt((double)n)));
    CTM_TRY(row_norms(ctx, X, np, np, np, norms));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * np, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // the n genuine rows are those with norm >= shift - fro > 0; padded rows are exactly zero
    std::vector<int> idx;
    for (int i = 0; i < np; ++i) if (h[i] > 0.0) idx.push_back(i);
    if ((int)idx.size() != n) { ctx->set_error("jacobi_eigh_top: rank bookkeeping failed"); return CTM_ERR_NOCONV; }
    std::vector<double> lam(np);
    for (int i : idx) lam[i] = h[i] - shift;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return std::fabs(lam[a]) > std::fabs(lam[c]); });
    std::vector<double> hd(k);
    for (int i = 0; i < k; ++i) hd[i] = lam[idx[i]];
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, idx.data(), sizeof(int) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(D, hd.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    CTM_TRY(gather_rows(ctx, X + (size_t)np * np, np, d_idx, k, n, Ut, n, nullptr));
    CTM_TRY(reorth_rows(ctx, Ut, k, n, n, 2));
    // eigenvalues as Rayleigh quotients u^T A u (drift-free, |error| = O(eps |A|))
    {
        double* Y;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Y));
        CTM_TRY(symmetrize_lower(ctx, A, As, n, 0.0));
        GemmDesc g; g.M = k; g.N = n; g.K = n; g.A = Ut; g.sam = n; g.sak = 1; g.B = As; g.sbk = n; g.sbn = 1; g.C = Y; g.ldc = n;
        CTM_TRY(gemm_f64(ctx, g));
        CTM_TRY(row_dots(ctx, Y, Ut, k, n, n, D));
    }
    return CTM_OK;
}
