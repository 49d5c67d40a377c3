// Shared internals of the chi-truncation engine (jacobi_core.hip: dense one-sided block Jacobi; svd_leading.hip: leading-k solvers of
// the implicit operator -- block power iteration, block Golub-Kahan-Lanczos, stationary Rayleigh-Ritz; eigh.hip: symmetric / Hermitian
// truncation with warm restart and orthogonal iteration).  Small kernels used by more than one of them are defined here with
// internal linkage (one copy per translation unit); host functions that cross translation units are declared at the end.
#pragma once
#include "ctm_common.h"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <mutex>
#include <type_traits>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int MAXM = 64;   // largest pair-Gram the LDS solver handles (2 * block)

// acceptance threshold of the leading-k solvers on the residual / s_0: the configured tolerance, but never below the rounding
// floor of applying an n-dimensional operator (a few ulps times sqrt(n): 5.7e-14 at n = 16384)
inline double resid_tol(const ctm_ctx* ctx, int n) { return std::max(ctx->si_tol, 4.0 * 1.1102230246251565e-16 * std::sqrt((double)n)); }

__global__ void axpy_kernel(double* x, const double* y, double a, size_t n) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) x[q] += a * y[q];
}

__global__ void sub_eye_kernel(double* G, int m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) G[(size_t)i * m + i] -= 1.0;
}

// dst (R x ld) <- [ src (rows x cols, lds) zero padded to R rows | identity (R x R) if with_eye ] ; other columns untouched
__global__ void fill_wq_kernel(const double* src, int rows, int cols, long long lds, double* dst, int R, long long ld, int with_eye) {
    const long long W = cols + (with_eye ? R : 0);
    const size_t tot = (size_t)R * W;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const long long r = q / W, c = q - r * W;
        double v;
        if (c < cols) v = (r < rows) ? src[r * lds + c] : 0.0;
        else v = ((c - cols) == r) ? 1.0 : 0.0;
        dst[r * ld + c] = v;
    }
}

__global__ void scale_rows_kernel(double* x, int rows, int cols, long long ld, const double* rs) {
    const size_t tot = (size_t)rows * cols;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const size_t r = q / cols, c = q - r * cols;
        x[r * ld + c] *= rs[r];
    }
}

__global__ void inv_or_zero_kernel(const double* s, double* out, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) out[i] = s[i] > 0.0 ? 1.0 / s[i] : 0.0;
}

// deterministic pseudo-random fill in (-0.5, 0.5) (splitmix64 hash of the element index)
__global__ void hash_fill_kernel(double* x, int rows, int cols, long long ld, unsigned long long seed) {
    const size_t tot = (size_t)rows * cols;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (q + 1) * 0x9E3779B97F4A7C15ULL + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= (z >> 31);
        const size_t r = q / cols, c = q - r * cols;
        x[r * ld + c] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}

// out[r] = | a[r,:] - s[r] * b[r,:] |
__global__ void resid_rows_kernel(const double* a, long long lda, const double* b, long long ldb, const double* s, int rows, int cols,
                                  double* out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        double acc = 0.0;
        const double sr = s[r];
        for (int c = lane; c < cols; c += 64) { const double d = a[(long long)r * lda + c] - sr * b[(long long)r * ldb + c]; acc += d * d; }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) out[r] = sqrt(acc);
    }
}

// =============================================================================================
// complex128 decomposition.  Working matrices hold complex rows in the panel layout of small_eig_c_kernel
// (16 real-part rows, then the 16 imaginary-part rows of the same complex rows); operators and results are planar.
// Row factors follow the real convention with ^T -> ^H:  Ut rows = u_k^H, Vt rows = v_k^H,  M = Ut^H diag(S) Vt.
// =============================================================================================
constexpr int BC = 16;
inline int crow_re(int cr) { return (cr / BC) * (2 * BC) + (cr % BC); }

// X (2*np real rows x ld) <- panel layout of [ M (n x n complex, planar) | identity (np complex columns) if with_eye ]
__global__ void fill_wq_c_kernel(const double* Mr, const double* Mi, int n, double* X, int np, long long ld, int with_eye) {
    const long long W = n + (with_eye ? np : 0);
    const size_t tot = (size_t)np * W;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const long long r = q / W, c = q - r * W;
        double vr, vi = 0.0;
        if (c < n) { vr = (r < n) ? Mr[r * n + c] : 0.0; vi = (r < n) ? Mi[r * n + c] : 0.0; }
        else vr = ((c - n) == r) ? 1.0 : 0.0;
        const long long rr = (r / BC) * (2 * BC) + (r % BC);
        X[rr * ld + c] = vr; X[(rr + BC) * ld + c] = vi;
    }
}

// X (2*np real rows x ld) <- panel layout of [ Y (n x n complex, planar) | W (n x n complex, planar) ], everything else zero
__global__ void fill_wq_c2_kernel(const double* Yr, const double* Yi, const double* Wr, const double* Wi, int n, double* X, int np, long long ld) {
    const long long W = n + np;
    const size_t tot = (size_t)np * W;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const long long r = q / W, c = q - r * W;
        double vr = 0.0, vi = 0.0;
        if (r < n) {
            if (c < n) { vr = Yr[r * n + c]; vi = Yi[r * n + c]; }
            else if (c - n < n) { vr = Wr[r * n + (c - n)]; vi = Wi[r * n + (c - n)]; }
        }
        const long long rr = (r / BC) * (2 * BC) + (r % BC);
        X[rr * ld + c] = vr; X[(rr + BC) * ld + c] = vi;
    }
}

// dst = i * src on panel rows: real-part rows <- -imag rows, imag rows <- real-part rows
__global__ void panel_times_i_kernel(const double* src, long long lds, double* dst, long long ldd, int R, int cols) {
    const size_t tot = (size_t)R * cols;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const long long r = q / cols, c = q - r * cols;
        const bool is_im = ((r / BC) & 1) != 0;
        dst[r * ldd + c] = is_im ? src[(r - BC) * lds + c] : -src[(r + BC) * lds + c];
    }
}

// panel rows [row0, row0 + rows) of dst <- planar complex rows (re, im: rows x cols, leading dim lds)
__global__ void planar_to_panel_kernel(const double* re, const double* im, long long lds, int rows, int cols, double* dst, long long ldd, int row0) {
    const size_t tot = (size_t)rows * cols;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const long long r = q / cols, c = q - r * cols;
        const long long cr = row0 + r, rr = (cr / BC) * (2 * BC) + (cr % BC);
        dst[rr * ldd + c] = re[r * lds + c]; dst[(rr + BC) * ldd + c] = im[r * lds + c];
    }
}

// (Pr + i Pi) <- 1 - (Pr + i Pi)
__global__ void eye_minus_kernel(double* Pr, double* Pi, int n) {
    const size_t tot = (size_t)n * n;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const size_t r = q / n, c = q - r * n;
        Pr[q] = (r == c ? 1.0 : 0.0) - Pr[q];
        Pi[q] = -Pi[q];
    }
}

__global__ void add_inplace_kernel(double* x, const double* y, size_t n) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) x[q] += y[q];
}

__global__ void sub_inplace_kernel(double* x, const double* y, size_t n) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) x[q] -= y[q];
}

// per real row r of a panel matrix: out[r] = norm of the complex row it belongs to (both of its real rows get the value)
__global__ void panel_combine_kernel(const double* nr, double* out, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) {
        const int re = ((r / BC) & 1) ? r - BC : r;
        out[r] = sqrt(nr[re] * nr[re] + nr[re + BC] * nr[re + BC]);
    }
}

// ---------------------------------------------------------------------------------------------
// leading-k decomposition by block Golub-Kahan-Lanczos with full re-orthogonalisation, for operators whose spectrum does
// NOT collapse inside a small block (the subspace iteration above then needs 15-25 half steps with a Rayleigh-Ritz on
// p = k + k/2 long rows each).  Row bases U_1..U_j, V_1..V_j (blocks of 64 rows):
//     W = V_j M^T  - (projection on U_1..U_{j-1})  ->  rows orthonormalised  ->  U_j        (so M V_j^T lies in span U_1..U_j)
//     Z = U_j M    - (projection on V_1..V_j)      ->  rows orthonormalised  ->  V_{j+1}
// Ritz extraction from the SMALL matrix T = U_all M V_all^T (jb x jb, assembled from the stored raw products U_i M), one
// dense Jacobi SVD of it; the coupling E = (U_all M) V_{j+1}^T gives the residual estimate |x_i^T E|; when it passes, the
// Ritz triplets are formed and BOTH relations are verified with the operator itself (same acceptance as svd_iter).
// ---------------------------------------------------------------------------------------------
// Cholesky factor of a 64 x 64 Gram matrix and the inverse of its lower factor, one workgroup, everything in LDS:
// G = L L^T, out = L^-1 (lower triangular, row-major); status[0] = smallest pivot met (<= 0: not positive definite).
__global__ __launch_bounds__(64) void chol64_inv_kernel(const double* G, double* Linv, double* status, int m = 64) {
    // ONE wave, thread i owns row i of L (kept in LDS, row stride 65: a column access by the wave is conflict free, a pivot-row
    // access is a broadcast).  Left-looking factorisation: every thread recomputes the pivot itself, so a column costs one
    // barrier; then L X = I by forward substitution, thread c owning column c of X.  m <= 64: order of the (dense, leading dimension m) matrix.
    constexpr int M = 64;
    __shared__ double L[M][M + 1];
    __shared__ double X[M][M + 1];
    const int i = threadIdx.x;
    const bool act = i < m;
    for (int c = 0; c < m; ++c) L[i][c] = act ? G[c * m + i] : 0.0;          // G is symmetric: column i read as row i, coalesced
    __syncthreads();
    double pmin = 1e300;
    for (int j = 0; j < m; ++j) {
        // the LDS reads do not depend on the accumulators: eight iterations' loads are issued together (four accumulator chains)
        double dot0 = 0.0, dot1 = 0.0, pd0 = 0.0, pd1 = 0.0;
        int t = 0;
        for (; t + 8 <= j; t += 8) {
            double a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u] = L[i][t + u]; b[u] = L[j][t + u]; }
#pragma unroll
            for (int u = 0; u < 8; u += 2) { dot0 += a[u] * b[u]; dot1 += a[u + 1] * b[u + 1]; pd0 += b[u] * b[u]; pd1 += b[u + 1] * b[u + 1]; }
        }
        for (; t < j; ++t) { const double ljt = L[j][t]; dot0 += L[i][t] * ljt; pd0 += ljt * ljt; }
        const double dot = dot0 + dot1, pd = pd0 + pd1;
        const double p = L[j][j] - pd;
        pmin = fmin(pmin, p);
        const double l = sqrt(fmax(p, 1e-300));
        const double v = (i == j) ? l : (L[i][j] - dot) / l;
        __syncthreads();                                          // everybody has read the old L[j][j]
        if (i >= j && act) L[i][j] = v;
        __syncthreads();
    }
    const int c = i;
    for (int r = 0; r < m; ++r) {       // X[t][c] = 0 for t < c: the sum may start at 0 for every thread (uniform trip count)
        double acc0 = (r == c) ? 1.0 : 0.0, acc1 = 0.0;
        int t = 0;
        for (; t + 8 <= r; t += 8) {
            double a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u] = L[r][t + u]; b[u] = X[t + u][c]; }
#pragma unroll
            for (int u = 0; u < 8; u += 2) { acc0 -= a[u] * b[u]; acc1 -= a[u + 1] * b[u + 1]; }
        }
        for (; t < r; ++t) acc0 -= L[r][t] * X[t][c];
        X[r][c] = (r >= c) ? (acc0 + acc1) / L[r][r] : 0.0;
    }
    if (act) for (int r = 0; r < m; ++r) Linv[r * m + c] = X[r][c];
    if (i == 0) status[0] = pmin;
}

// One wave, register resident (see chol64_scaled_inv_kernel below for the scheme): a value another lane needs travels through
// v_readlane, wave-wide maxima through DPP moves inside the rows of 16 lanes and four readlanes across them.
__device__ __forceinline__ double lane_bcast(double v, int lane) {       // `lane` must be wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) {                   // every lane receives the maximum over the 64 lanes
    v = fmax(v, dpp_move<0xB1>(v));        // quad_perm [1,0,3,2]
    v = fmax(v, dpp_move<0x4E>(v));        // quad_perm [2,3,0,1]
    v = fmax(v, dpp_move<0x141>(v));       // row_half_mirror
    v = fmax(v, dpp_move<0x140>(v));       // row_mirror: every lane of a row of 16 holds the row maximum
    return fmax(fmax(lane_bcast(v, 0), lane_bcast(v, 16)), fmax(lane_bcast(v, 32), lane_bcast(v, 48)));
}
__device__ __forceinline__ double wave_sum(double v) {                   // every lane receives the sum over the 64 lanes (same value on all)
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}

// Rank-revealing companion of chol64_inv_kernel: pivoted Cholesky of a 64 x 64 Gram matrix G = Z Z^T (lazy, left-looking: only the
// pivot columns are ever formed), stopped at the first pivot below rel * (largest diagonal entry).  Output Mo (64 x 64, row-major):
// rows k < rank hold row k of L_pp^-1 scattered to the pivot positions, so that Mo Z has orthonormal rows 0..rank-1 (Gram-Schmidt
// of the pivot rows in pivot order) and zero rows beyond; status[0] = rank.  One wave; lane j keeps row j of L in registers (entry
// t = pivot step t: the loops are fully unrolled, so the index is static) and reads the pivot row's entries from lane p by
// v_readlane; G stays in LDS for the one row per step that is read.  (91 us -> 44 us against the LDS version: six ds_bpermute
// rounds of the argmax and an LDS round trip per term were on the critical path of every step.)
__global__ __launch_bounds__(64) void pivchol64_inv_kernel(const double* __restrict__ G, double rel, double* __restrict__ Mo, double* __restrict__ status) {
    constexpr int M = 64;
    __shared__ double GS[M][M];
    const int j = threadIdx.x;
#pragma unroll 8
    for (int r = 0; r < M; ++r) GS[r][j] = G[r * M + j];
    __syncthreads();
    double d = GS[j][j];
    bool chosen = false;
    double d0 = 0.0, rinv = 0.0;            // lane k: 1 / L_pp[k][k]
    int rank = 0, myperm = 0;               // lane k remembers the pivot of step k
    double L[M];
    bool stop = false;                      // (no `break`: the loops must unroll completely for L[] and x[] to stay in registers)
#pragma unroll
    for (int k = 0; k < M; ++k) {
        if (stop) continue;
        const double cand = chosen ? -1.0 : d;
        const double best = wave_max(cand);
        if (k == 0) d0 = best;
        if (!(best > rel * d0) || !(best > 0.0)) { stop = true; continue; }     // uniform: every lane holds the same maximum
        const int p = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)__ballot(cand == best)) - 1);     // ties: the lowest index
        double acc0 = GS[p][j], acc1 = 0.0;                       // G[j][p] (symmetric)
#pragma unroll
        for (int t = 0; t < k; ++t) {
            const double lp = lane_bcast(L[t], p);
            if (t & 1) acc1 -= L[t] * lp; else acc0 -= L[t] * lp;
        }
        double rl = __builtin_amdgcn_rsq(best);
        rl = rl * (1.5 - 0.5 * best * rl * rl);
        rl = rl * (1.5 - 0.5 * best * rl * rl);
        const double v = (j == p) ? best * rl : (chosen ? 0.0 : (acc0 + acc1) * rl);
        L[k] = v;
        if (j == p) chosen = true; else if (!chosen) d -= v * v;
        if (j == k) { myperm = p; rinv = rl; }
        rank = k + 1;
    }
    // X = L_pp^-1 with L_pp[r][t] = L[t] of lane perm[r] (lower triangular, rank x rank); lane c builds column c
    double x[M];
#pragma unroll
    for (int r = 0; r < M; ++r) {
        if (r >= rank) { x[r] = 0.0; continue; }
        const int pr = __builtin_amdgcn_readlane(myperm, r);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int t = 0; t < r; ++t) {
            const double l = lane_bcast(L[t], pr);
            if ((t & 3) == 0) s0 += l * x[t]; else if ((t & 3) == 1) s1 += l * x[t]; else if ((t & 3) == 2) s2 += l * x[t]; else s3 += l * x[t];
        }
        const double rr = lane_bcast(rinv, r);
        x[r] = j < r ? -((s0 + s1) + (s2 + s3)) * rr : (j == r ? rr : 0.0);
    }
    // Mo[k][perm[t]] = X[k][t]: lane j zeroes column j, then (j < rank) fills column perm[j]
#pragma unroll 8
    for (int k = 0; k < M; ++k) Mo[k * M + j] = 0.0;
    __syncthreads();
    if (j < rank) {
#pragma unroll
        for (int k = 0; k < M; ++k) if (k < rank) Mo[k * M + myperm] = x[k];
    }
    if (j == 0) status[0] = (double)rank;
}

// Sync-free variant of the block orthonormalisation (the Krylov recurrence issues its steps without waiting for the device):
// Cholesky-QR of the 64 rows with the unit-norm scaling folded into the factorisation.  ONE wave, everything in registers, no LDS
// and no barrier: lane i holds row i of the scaled matrix (64 doubles), a value another lane needs travels through v_readlane with
// a compile-time lane index (the loops are fully unrolled), so a step of either phase is two readlanes and one FMA with an SGPR
// operand instead of an LDS round trip behind a barrier:
//   d_i = G_ii (squared row norms), A = D^-1/2 G D^-1/2 (unit diagonal);
//   A = L L^T, right-looking: column j is scaled by rsqrt(pivot) (hardware estimate + two Newton steps; the pivot is wave-uniform),
//     then a[k] -= a[j] * L_kj for k > j, L_kj read from lane k;
//   X = L^-1: lane c builds column c, x_r = -(sum_{t<r} L_rt x_t) / L_rr with L_rt read from lane r (x_t = 0 for t < c by construction);
//   out = X D^-1/2  so that  out * W  has orthonormal rows.
// 108 us -> 41 us per call against the 256-thread LDS version it replaces (2100 cycles per column and 2000 per row of the inverse there:
// barrier, LDS latency, ds_bpermute reductions and IEEE divisions on the critical path).
// mode 0 (first pass): when a pivot of A falls below 1e-10 (rows nearly dependent: cond(W) > ~1e5, beyond two Cholesky-QR passes)
//   the factorisation is repeated on A + 1e-10 I (shifted Cholesky-QR: the result is only roughly orthonormal, cond ~ 1e-5 cond(W))
//   and *flag3 is set: a third pass then finishes.  mode 1: plain.  mode 2 (third pass): returns at once unless *flag3.
// status[0] = smallest pivot of the accepted factorisation (NEGATED when the first pass shifted), [1] = smallest, [2] = largest row norm.  Nothing is decided on the
// host here: the caller reads the status words of all its steps at its next host synchronisation.
#ifdef CTM_KERNEL_CLOCKS
__device__ double ctm_dbg_clocks[4];    // phase clocks of the single-wave kernels (tools/bench_small_kernels.hip); NOT in the callers' status words
#endif
template <int M>      // M = 64, or 32 (block Krylov recurrence on 32-row blocks: lanes >= M carry empty rows; 128 instead of 256 VGPRs of rows)
__global__ __launch_bounds__(64) void chol64_scaled_inv_kernel(const double* __restrict__ G, double* __restrict__ out, double* __restrict__ status,
                                                               int* __restrict__ flag3, int mode) {
    const int lane = threadIdx.x;
    const bool act = lane < M;
    if (mode == 2 && *flag3 == 0) { if (lane == 0) { status[0] = 1.0; status[1] = -1.0; status[2] = -1.0; } return; }   // skipped: marked by the -1
#ifdef CTM_KERNEL_CLOCKS
    const long long clk0 = clock64();
#endif
    const double dg = act ? G[lane * M + lane] : 1.0;
    const double dinv = dg > 0.0 ? 1.0 / sqrt(dg) : 0.0;
    double a[M];                    // row `lane` of A, then of L
    double rinv = 0.0;              // lane j: 1 / L_jj
    double pmin = 1e300;
    bool shifted = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const double shift = attempt == 0 ? 0.0 : 1e-10;
        shifted = attempt == 1;
#pragma unroll
        for (int k = 0; k < M; ++k)     // G is symmetric: column `lane` read along rows (coalesced) is row `lane`
            a[k] = act ? G[k * M + lane] * dinv * lane_bcast(dinv, k) + ((k == lane) ? shift : 0.0) : 0.0;
        pmin = 1e300;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const double piv = lane_bcast(a[j], j);
            pmin = fmin(pmin, piv);
            const double ps = fmax(piv, 1e-300);
            double rl = __builtin_amdgcn_rsq(ps);
            rl = rl * (1.5 - 0.5 * ps * rl * rl);
            rl = rl * (1.5 - 0.5 * ps * rl * rl);
            a[j] *= rl;                                  // L_ij for every row i (lane j: L_jj)
            if (lane == j) rinv = rl;
#pragma unroll
            for (int k = j + 1; k < M; ++k) a[k] -= a[j] * lane_bcast(a[j], k);
        }
        if (mode != 0 || attempt == 1) break;
        if (pmin > 1e-10) { if (lane == 0) *flag3 = 0; break; }      // (uniform: every lane tracked the same pivots)
        if (lane == 0) *flag3 = 1;
    }
#ifdef CTM_KERNEL_CLOCKS
    const long long clk1 = clock64();
#endif
    double x[M];                    // column `lane` of X = L^-1
#pragma unroll
    for (int r = 0; r < M; ++r) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int t = 0; t < r; ++t) {
            const double l = lane_bcast(a[t], r);       // L_rt
            if ((t & 3) == 0) s0 += l * x[t]; else if ((t & 3) == 1) s1 += l * x[t]; else if ((t & 3) == 2) s2 += l * x[t]; else s3 += l * x[t];
        }
        const double rr = lane_bcast(rinv, r);
        const double sum = (s0 + s1) + (s2 + s3);
        x[r] = lane < r ? -sum * rr : (lane == r ? rr : 0.0);
    }
#ifdef CTM_KERNEL_CLOCKS
    const long long clk2 = clock64();
    if (lane == 0) { ctm_dbg_clocks[0] = (double)(clk1 - clk0); ctm_dbg_clocks[1] = (double)(clk2 - clk1); }
#endif
    if (act) {
#pragma unroll
        for (int r = 0; r < M; ++r) out[r * M + lane] = x[r] * dinv;
    }
    double mn = act ? (dg > 0.0 ? sqrt(dg) : 0.0) : 1e300, mx = act ? (dg > 0.0 ? sqrt(dg) : 0.0) : 0.0;
    for (int off = 32; off > 0; off >>= 1) { mn = fmin(mn, __shfl_down(mn, off, 64)); mx = fmax(mx, __shfl_down(mx, off, 64)); }
    if (lane == 0) { status[0] = shifted ? -pmin : pmin; status[1] = mn; status[2] = mx; }      // (negative: the first pass had to shift -- a third pass is due)
}

// ---------------------------------------------------------------------------------------------
// complex128 block Golub-Kahan-Lanczos (planar data): the algorithm of svd_lanczos() with ^T -> ^H.  Row bases hold u^H, v^H.
// ---------------------------------------------------------------------------------------------
// Cholesky factor of a 64 x 64 Hermitian Gram matrix (planar) and the inverse of its lower factor
__global__ __launch_bounds__(256) void chol64_inv_c_kernel(const double* Gr, const double* Gi, double* Lr, double* Li, double* status, int m = 64) {
    constexpr int M = 64;                 // capacity; m <= 64 is the order of the (dense, leading dimension m) matrices
    __shared__ double Ar[M][M + 1], Ai[M][M + 1];
    __shared__ double Xr[M][M + 1], Xi[M][M + 1];
    __shared__ double piv_min;
    const int tid = threadIdx.x;
    for (int q = tid; q < m * m; q += 256) {
        const int i = q / m, c = q - i * m;
        Ar[i][c] = Gr[q]; Ai[i][c] = (i == c) ? 0.0 : Gi[q];
        Xr[i][c] = (i == c) ? 1.0 : 0.0; Xi[i][c] = 0.0;
    }
    if (tid == 0) piv_min = 1e300;
    __syncthreads();
    for (int j = 0; j < m; ++j) {
        const double d = Ar[j][j];
        if (tid == 0) piv_min = fmin(piv_min, d);
        const double l = sqrt(fmax(d, 1e-300));
        __syncthreads();
        if (tid < m) {
            if (tid == j) { Ar[j][j] = l; Ai[j][j] = 0.0; }
            else if (tid > j) { Ar[tid][j] /= l; Ai[tid][j] /= l; }
        }
        __syncthreads();
        for (int q = tid; q < m * m; q += 256) {            // A[i][c] -= A[i][j] conj(A[c][j])   (lower triangle)
            const int i = q / m, c = q - i * m;
            if (c > j && i >= c) {
                const double xr = Ar[i][j], xi = Ai[i][j], yr = Ar[c][j], yi = Ai[c][j];
                Ar[i][c] -= xr * yr + xi * yi;
                Ai[i][c] -= xi * yr - xr * yi;
            }
        }
        __syncthreads();
    }
    if (tid < m) {                                            // L X = I, one column per thread
        const int c = tid;
        for (int i = c; i < m; ++i) {
            double ar = (i == c) ? 1.0 : 0.0, ai = 0.0;
            for (int t = c; t < i; ++t) {
                ar -= Ar[i][t] * Xr[t][c] - Ai[i][t] * Xi[t][c];
                ai -= Ar[i][t] * Xi[t][c] + Ai[i][t] * Xr[t][c];
            }
            Xr[i][c] = ar / Ar[i][i]; Xi[i][c] = ai / Ar[i][i];
        }
    }
    __syncthreads();
    for (int q = tid; q < m * m; q += 256) { const int i = q / m, c = q - i * m; Lr[q] = (c <= i) ? Xr[i][c] : 0.0; Li[q] = (c <= i) ? Xi[i][c] : 0.0; }
    if (tid == 0) status[0] = piv_min;
}

struct CRows { double* re; double* im; };      // planar complex row block (rows x n, leading dimension n)

// number of rows after padding to an even number of b-row panels (>= 2)
inline int padded(int n, int b) {
    int nbk = (n + b - 1) / b;
    if (nbk < 2) nbk = 2;
    if (nbk & 1) ++nbk;
    return nbk * b;
}

}  // namespace

// ---- host functions shared between the translation units --------------------------------------------------------
struct RRTables;
int get_tables(ctm_ctx* ctx, int nbk, long long ld, int b, int Cg, RRTables** out);
int jacobi_rows(ctm_ctx* ctx, double* X, int R, long long ld, int Cg, int Ctot, int b, int ktop, double fro, int max_sweeps, bool cplx = false,
                   bool tau_both = false, double null_rel = 0.0);
int choose_block(ctm_ctx* ctx, int n);
int reorth_rows(ctm_ctx* ctx, double* V, int k, int n, long long ld, int iters);
double host_fro(ctm_ctx* ctx, const double* M, int rows, int cols, long long ld, double* d_tmp, std::vector<double>& h, int* status);
int complete_null_rows(ctm_ctx* ctx, double* Vt, int kg, int k, int n);
int svd_full_rot_rows(ctm_ctx* ctx, int n);
int svd_full(ctm_ctx* ctx, const double* M, int n, int k, double* S, double* Ut, double* Vt, double* warm = nullptr, double* rot = nullptr, bool rot_valid = false);
int svd_full_polar(ctm_ctx* ctx, const double* M, int n, double* S, double* Ut, double* Vt, double* warm);
int panel_row_norms(ctm_ctx* ctx, const double* X, int np, int cols, long long ld, double* norms, std::vector<double>& hc);
int panel_gather(ctm_ctx* ctx, const double* X, long long ld, const std::vector<int>& idx, int k, int cols, double* out, int* d_idx2);
int scale_planar_rows(ctm_ctx* ctx, double* V, int k, int n, const double* inv);
int reorth_rows_c(ctm_ctx* ctx, double* V, int k, int n, int iters);
int svd_full_c(ctm_ctx* ctx, const double* Mr, const double* Mi, int n, int k, double* S, double* Ut, double* Vt, double* warm = nullptr, double* rot = nullptr, bool rot_valid = false);
int rows_times(ctm_ctx* ctx, const double* X, long long ldx, int p, int kin, int nout, const double* Z, bool transZ, double* Y, long long ldy);
int matop_apply(ctm_ctx* ctx, const MatOp& op, bool transpose, const double* B, long long ldb, int p, double* C, long long ldc,
                   double* mid = nullptr);
int svd_iter(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged, bool* want_krylov = nullptr);
int rows_times_c(ctm_ctx* ctx, const double* X, long long ldx, int R, int kin, int nout, const double* Zr, const double* Zi, bool trans, bool conj,
                    double* Y, long long ldy, double* scratch);
int matop_apply_c(ctm_ctx* ctx, const MatOp& op, bool adjoint, const double* B, long long ldb, int R, double* C, long long ldc);
int svd_iter_c(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged, bool* want_krylov = nullptr);
int orthonormalise_block(ctm_ctx* ctx, double* W, int rows, int n, double* norms, double* inv, double* min_norm, double* max_norm);
int orthonormalise_block_async(ctm_ctx* ctx, double* W, int b, int n, double* G, double* Li, double* status, int* flag3, int npass = 3);
int project_out(ctm_ctx* ctx, double* W, int b, int n, const double* B, int m, double* G, int reps = 2, int local = 0);
int svd_lanczos(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged);
int orthonormalise_block_c(ctm_ctx* ctx, CRows W, int rows, int n, double* norms, double* inv, double* min_norm, double* max_norm, bool* ok);
int project_out_c(ctm_ctx* ctx, CRows W, int b, int n, const double* Bre, const double* Bim, int m, double* G, double* T);
int matop_apply_planar(ctm_ctx* ctx, const MatOp& op, bool adjoint, const double* Bre, const double* Bim, int rows, double* Cre, double* Cim);
int svd_lanczos_c(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged);
int svd_stationary(ctm_ctx* ctx, const MatOp& op, int k, int side0, double* S, double* Ut, double* Vt, bool* accepted, double* resid_rel);
int svd_stationary_c(ctm_ctx* ctx, const MatOp& op, int k, int side0, double* S, double* Ut, double* Vt, bool* accepted, double* resid_rel);
int spectrum_movement(ctm_ctx* ctx, double* hdr_row, int n, const double* S, int k, double* moved);
