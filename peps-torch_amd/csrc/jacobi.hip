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
#include <mutex>
#include <type_traits>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int MAXM = 64;   // largest pair-Gram the LDS solver handles (2 * block)

// acceptance threshold of the leading-k solvers on the residual / s_0: the configured tolerance, but never below the rounding
// floor of applying an n-dimensional operator (a few ulps times sqrt(n): 5.7e-14 at n = 16384)
inline double resid_tol(const ctm_ctx* ctx, int n) { return std::max(ctx->si_tol, 4.0 * 1.1102230246251565e-16 * std::sqrt((double)n)); }

// ---------------------------------------------------------------------------------------------
// batched small symmetric eigensolver: one workgroup (256 threads) per m x m Gram matrix
// ---------------------------------------------------------------------------------------------
struct SmallEigParams {
    const double* G;     // nsplit x batch x m x m partial Gram matrices (summed on load)
    int nsplit; long long split_stride;
    double* J;           // batch x m x m (columns = eigenvectors, eigenvalues descending)
    int m;
    double tol;          // relative off-diagonal tolerance for a rotation
    int max_sweeps;
    double tau2;         // scale floor: a pair (i,j) is measured against max(sqrt(g_ii g_jj), tau2)
    int tau_both;        // 1: the floor applies only when BOTH rows are below it (g_ii, g_jj < tau2); pairs with a leading row keep full relative accuracy
    unsigned long long* stat_rel;   // max |g_ij| / max(sqrt(g_ii g_jj), tau2)  (bits of a non-negative double)
    unsigned long long* stat_abs;   // max |g_ij| / sqrt(g_ii g_jj) (classical measure, diagnostics only)
    int* flags;          // per pair: 1 if J != I (the apply GEMM skips the others)
    int cross = 0;       // small_eig64_kernel: 1 = rotate only the 32 x 32 pairs (row of panel 0, row of panel 1), 32 rounds instead of 63
};

// floor of the pair measure: with `both`, a pair that contains a row at or above the floor is measured relative to its own rows only
// both == 2 (absolute accuracy, full decompositions of the differentiable route): tau2 holds s_0 (the largest row norm) and the pair
// is measured against s_0 * max(|x_i|, |x_j|) instead of |x_i| |x_j| -- the orthogonality a pair needs for the reconstruction
// U S V^H = M to hold to tol * s_0 (what LAPACK's bidiagonal SVD delivers), not for every small singular value to be relatively exact
__device__ __forceinline__ double tau_floor(double a, double b, double tau2, int both) {
    if (both == 2) return tau2 * sqrt(fmax(fabs(a), fabs(b)));
    return (both && (a >= tau2 || b >= tau2)) ? 0.0 : tau2;
}


// Two-sided cyclic Jacobi on one m x m (m <= 64, even) symmetric matrix per workgroup, everything in LDS.
// Round-robin ordering: m/2 disjoint rotations per round; per round the rotation parameters are computed by
// m/2 lanes, then every thread transforms whole 2x2 blocks W[{p1,q1}][{p2,q2}] <- R1^T B R2 (row and column
// update fused: each block is owned by exactly one thread) and the eigenvector columns -- two barriers per round.
// MX = 64 or 32: capacity of the LDS images (m <= MX).  The 32 variant holds 18 KB of LDS: a workgroup of it fits on a CU beside the
// chip-filling kernels of another unit (row-block GEMM: 2 x 50 KB), which is what lets a Ritz extraction on 16-row panels proceed
// while another unit's corner passes are resident.
template <int MX>
__global__ __launch_bounds__(MX == 64 ? 1024 : 256) void small_eig_kernel(SmallEigParams p) {
    const int NTH = blockDim.x;      // 1024 for m = 64 (one 2x2 block + two eigenvector rows per thread), 256 for m <= 32
    __shared__ double W[MX][MX + 1];
    __shared__ double Jm[MX][MX + 1];
    __shared__ double cs_c[MX / 2], cs_s[MX / 2];
    __shared__ int pr_p[MX / 2], pr_q[MX / 2];
    __shared__ double red[16];
    __shared__ int rot_flag;
    __shared__ int round_rot[2];
    __shared__ unsigned char pair_tab[MX - 1][MX / 2][2];
    __shared__ int rank_of[MX];

    const int m = p.m, tid = threadIdx.x;
    const double* G = p.G + (size_t)blockIdx.x * m * m;
    double* Jout = p.J + (size_t)blockIdx.x * m * m;

    {   // sum the split-K partial Grams: 16 independent accumulators per thread keep the loads in flight
        double acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.0;
        for (int s = 0; s < p.nsplit; ++s) {
            const double* Gs = G + (size_t)s * p.split_stride;
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int q = tid + u * NTH; if (q < m * m) acc[u] += Gs[q]; }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = tid + u * NTH;
            if (q < m * m) { const int r = q / m, c = q - r * m; W[r][c] = acc[u]; Jm[r][c] = (r == c) ? 1.0 : 0.0; }
        }
    }
    __syncthreads();

    // convergence statistics of the INCOMING Gram matrix
    {
        double srel = 0.0, sabs = 0.0;
        for (int q = tid; q < m * m; q += NTH) {
            const int r = q / m, c = q - r * m;
            if (r < c) {
                const double g = fabs(W[r][c]), a = W[r][r], b = W[c][c];
                if (g > 0.0) {
                    const double sc = sqrt(fabs(a * b));
                    srel = fmax(srel, g / fmax(sc, tau_floor(a, b, p.tau2, p.tau_both)));
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
            double v = 0.0;
            for (int w = 0; w < (NTH >> 6); ++w) v = fmax(v, red[w]);
            atomicMax(p.stat_rel, (unsigned long long)__double_as_longlong(v));
            red[0] = v;
        }
        __syncthreads();
        srel = red[0];
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = sabs;
        __syncthreads();
        if (tid == 0) {
            double v = 0.0;
            for (int w = 0; w < (NTH >> 6); ++w) v = fmax(v, red[w]);
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
    for (int q = tid; q < mm1 * half; q += NTH) {      // round-robin schedule of all rounds, once
        const int r = q / half, k = q - r * half;
        int pi, qi;
        if (k == 0) { pi = mm1; qi = r % mm1; }
        else { pi = (r + k) % mm1; qi = (r - k + mm1) % mm1; }
        if (pi > qi) { const int t = pi; pi = qi; qi = t; }
        pair_tab[r][k][0] = (unsigned char)pi; pair_tab[r][k][1] = (unsigned char)qi;
    }
    __syncthreads();
    const int k2 = tid % half;            // column pair owned by this thread
    const int g0 = tid / half;            // first row-pair / row group
    const int ngrp = NTH / half;          // thread groups along the other dimension
    for (int sweep = 0; sweep < p.max_sweeps; ++sweep) {
        if (tid == 0) { rot_flag = 0; round_rot[0] = 0; }
        __syncthreads();
        for (int r = 0; r < mm1; ++r) {
            if (tid == 0) round_rot[(r + 1) & 1] = 0;      // slot of the NEXT round (nobody reads it before the next barrier pair)
            if (tid < half) {
                const int pi = pair_tab[r][tid][0], qi = pair_tab[r][tid][1];
                const double a = W[pi][pi], b = W[qi][qi], g = W[pi][qi];
                double c = 1.0, s = 0.0;
                if (g != 0.0 && fabs(g) > p.tol * fmax(sqrt(fabs(a * b)), tau_floor(a, b, p.tau2, p.tau_both))) {
                    // tan of the rotation angle: t = 2g / (d + sign(d) hypot(d, 2g)), d = b - a.  Only c^2 + s^2 = 1 has to
                    // hold to machine precision (orthogonality); the angle itself may carry the ~1e-8 error of the
                    // hardware rsq/rcp seeds, so t uses the fast seeds and c = rsqrt(1 + t^2) gets two Newton steps.
                    const double d = b - a, g2 = 2.0 * g;
                    const double hh = d * d + g2 * g2;
                    double rh = __builtin_amdgcn_rsq(hh);
                    rh = rh * (1.5 - 0.5 * hh * rh * rh);
                    const double h = hh * rh;                                  // hypot
                    const double den = d + (d >= 0.0 ? h : -h);
                    double rd = __builtin_amdgcn_rcp(den);
                    rd = rd * (2.0 - den * rd);
                    const double t = g2 * rd;
                    const double x = 1.0 + t * t;
                    double y = __builtin_amdgcn_rsq(x);
                    y = y * (1.5 - 0.5 * x * y * y);
                    y = y * (1.5 - 0.5 * x * y * y);
                    c = y;
                    s = t * c;
                    rot_flag = 1;
                    round_rot[r & 1] = 1;
                }
                cs_c[tid] = c; cs_s[tid] = s; pr_p[tid] = pi; pr_q[tid] = qi;
            }
            __syncthreads();
            if (round_rot[r & 1]) {      // uniform: skip the whole update when no pair of this round rotates
                const double c2 = cs_c[k2], s2 = cs_s[k2];
                const int p2 = pr_p[k2], q2 = pr_q[k2];
                // stage 1: all LDS loads (2x2 blocks (k1,k2) and the eigenvector columns of pair k2)
                double b00[4], b01[4], b10[4], b11[4], c1[4], s1[4], jp[8], jq[8];
                int p1[4], q1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k1 = g0 + u * ngrp;
                    if (k1 < half) {
                        c1[u] = cs_c[k1]; s1[u] = cs_s[k1]; p1[u] = pr_p[k1]; q1[u] = pr_q[k1];
                        b00[u] = W[p1[u]][p2]; b01[u] = W[p1[u]][q2]; b10[u] = W[q1[u]][p2]; b11[u] = W[q1[u]][q2];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = g0 + u * ngrp;
                    if (i < m) { jp[u] = Jm[i][p2]; jq[u] = Jm[i][q2]; }
                }
                // stage 2: B <- R1^T B R2 ; J columns <- J R2 ; stage 3: stores (each element is owned by one thread)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k1 = g0 + u * ngrp;
                    if (k1 < half) {
                        const double t00 = c2 * b00[u] - s2 * b01[u], t01 = s2 * b00[u] + c2 * b01[u];
                        const double t10 = c2 * b10[u] - s2 * b11[u], t11 = s2 * b10[u] + c2 * b11[u];
                        W[p1[u]][p2] = c1[u] * t00 - s1[u] * t10; W[p1[u]][q2] = c1[u] * t01 - s1[u] * t11;
                        W[q1[u]][p2] = s1[u] * t00 + c1[u] * t10; W[q1[u]][q2] = s1[u] * t01 + c1[u] * t11;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = g0 + u * ngrp;
                    if (i < m) { Jm[i][p2] = c2 * jp[u] - s2 * jq[u]; Jm[i][q2] = s2 * jp[u] + c2 * jq[u]; }
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
    for (int q = tid; q < m * m; q += NTH) {
        const int r = q / m, c = q - r * m;
        Jout[r * m + rank_of[c]] = Jm[r][c];
    }
}

// m = 64 variant with ONE barrier per round: the matrix ping-pongs between two LDS images, every thread derives the two
// rotations it needs (row pair k1, column pair k2) itself from the source image -- bitwise identical on all threads that share
// a pair -- transforms its own 2x2 block into the other image and its two eigenvector rows in place.  1024 threads.
// round-robin pairing of M = 64 indices: round r (0..62), slot k (0..31) -> the pair (p < q); same schedule as the table the
// other kernels build, computed instead of read (six LDS byte loads per thread and round less)
__device__ __forceinline__ void rr_pair64(int r, int k, int& p, int& q) {
    int a = r + k; a = (a >= 63) ? a - 63 : a;
    int b = r - k + 63; b = (b >= 63) ? b - 63 : b;
    if (k == 0) { a = 63; b = r; }
    p = min(a, b); q = max(a, b);
}
// cross pairs only: round r (0..31), slot k -> (row k of the first panel, row (k + r) mod 32 of the second).  The rows inside a panel
// were made orthogonal by the one full round of the sweep (jacobi_rows: round 0) and are not rotated against each other again:
// the block sweep then is the classical cyclic sweep -- every row pair once -- instead of 27 repetitions of the intra-panel pairs
__device__ __forceinline__ void cross_pair64(int r, int k, int& p, int& q) { p = k; q = 32 + ((k + r) & 31); }

__device__ __forceinline__ void jacobi_cs(double a, double b, double g, double tol, double tau2in, int both, double& c, double& s, bool& rot) {
    c = 1.0; s = 0.0; rot = false;
    const double tau2 = tau_floor(a, b, tau2in, both);
    // |g| > tol * max(sqrt|a b|, tau2)  <=>  g^2 > tol^2 * max(|a b|, tau2^2): no square root on the critical path
    if (g != 0.0 && g * g > tol * tol * fmax(fabs(a * b), tau2 * tau2)) {
        const double d = b - a, g2 = 2.0 * g;
        const double hh = d * d + g2 * g2;
        double rh = __builtin_amdgcn_rsq(hh);
        rh = rh * (1.5 - 0.5 * hh * rh * rh);
        const double h = hh * rh;
        const double den = d + (d >= 0.0 ? h : -h);
        double rd = __builtin_amdgcn_rcp(den);
        rd = rd * (2.0 - den * rd);
        const double t = g2 * rd;
        const double x = 1.0 + t * t;
        double y = __builtin_amdgcn_rsq(x);
        y = y * (1.5 - 0.5 * x * y * y);
        y = y * (1.5 - 0.5 * x * y * y);
        c = y; s = t * c; rot = true;
    }
}

template <int BPT, bool CROSS = false>     // 2x2 blocks per thread: 1024 / BPT threads per workgroup; CROSS: cross-pair rounds (compile time: a
// run-time choice of the pairing inside the round loop cost 10 % of the kernel -- 600 -> 1130 clocks of address arithmetic + LDS loads per round)
__global__ __launch_bounds__(1024 / BPT) void small_eig64_kernel(SmallEigParams p) {
    constexpr int M = 64, H = 32, NTH = 1024 / BPT, NW = NTH / 64, KS = H / BPT, EPT = (M * M) / NTH;
    __shared__ double Wb[2][M][M + 1];
    __shared__ double Jm[M][M + 1];
    __shared__ double red[16];
    __shared__ int rot_flag;
    __shared__ int rank_of[M];
    const int tid = threadIdx.x;
    const double* G = p.G + (size_t)blockIdx.x * M * M;
    double* Jout = p.J + (size_t)blockIdx.x * M * M;
    {
        double acc[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) acc[u] = 0.0;
        for (int s = 0; s < p.nsplit; ++s) {
            const double* Gs = G + (size_t)s * p.split_stride;
#pragma unroll
            for (int u = 0; u < EPT; ++u) acc[u] += Gs[tid + u * NTH];
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u) { const int q = tid + u * NTH, r = q >> 6, c = q & 63; Wb[0][r][c] = acc[u]; Jm[r][c] = (r == c) ? 1.0 : 0.0; }
    }
    __syncthreads();
    {
        double srel = 0.0, sabs = 0.0;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int q = tid + u * NTH, r = q >> 6, c = q & 63;
            if (r < c) {
                const double g = fabs(Wb[0][r][c]), a = Wb[0][r][r], b = Wb[0][c][c];
                if (g > 0.0) {
                    const double sc = sqrt(fabs(a * b));
                    srel = fmax(srel, g / fmax(sc, tau_floor(a, b, p.tau2, p.tau_both)));
                    if (sc > 0.0) sabs = fmax(sabs, g / sc);
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) { srel = fmax(srel, __shfl_down(srel, off, 64)); sabs = fmax(sabs, __shfl_down(sabs, off, 64)); }
        if ((tid & 63) == 0) red[tid >> 6] = srel;
        __syncthreads();
        if (tid == 0) {
            double v = 0.0;
            for (int w = 0; w < NW; ++w) v = fmax(v, red[w]);
            atomicMax(p.stat_rel, (unsigned long long)__double_as_longlong(v));
            red[0] = v;
        }
        __syncthreads();
        srel = red[0];
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = sabs;
        __syncthreads();
        if (tid == 0) {
            double v = 0.0;
            for (int w = 0; w < NW; ++w) v = fmax(v, red[w]);
            atomicMax(p.stat_abs, (unsigned long long)__double_as_longlong(v));
        }
        if (srel <= p.tol) { if (tid == 0) p.flags[blockIdx.x] = 0; return; }
        if (tid == 0) p.flags[blockIdx.x] = 1;
    }
    if (tid == 0) rot_flag = 0;
    __syncthreads();
    const int k2 = tid & 31, kb = tid >> 5;          // column pair; first of the BPT row pairs kb, kb + KS, ...
    int par = 0;
#ifdef CTM_KERNEL_CLOCKS
    long long ck_load = 0, ck_cs = 0, ck_upd = 0, ck_bar = 0;
#endif
    constexpr bool cross = CROSS;
    constexpr int nrounds = CROSS ? H : M - 1;
    for (int sweep = 0; sweep < p.max_sweeps; ++sweep) {
        for (int r = 0; r < nrounds; ++r) {
#ifdef CTM_KERNEL_CLOCKS
            const long long c0 = clock64();
#endif
            const double (*S)[M + 1] = Wb[par];
            double (*D)[M + 1] = Wb[par ^ 1];
            int p2, q2;
            if (cross) cross_pair64(r, k2, p2, q2); else rr_pair64(r, k2, p2, q2);
            const double a2 = S[p2][p2], d2 = S[q2][q2], g2 = S[p2][q2];
            int p1[BPT], q1[BPT];
            double b00[BPT], b01[BPT], b10[BPT], b11[BPT], jp0[BPT], jq0[BPT], jp1[BPT], jq1[BPT];
#pragma unroll
            for (int u = 0; u < BPT; ++u) {
                const int k1 = kb + u * KS;
                if (cross) cross_pair64(r, k1, p1[u], q1[u]); else rr_pair64(r, k1, p1[u], q1[u]);
                b00[u] = S[p1[u]][p2]; b01[u] = S[p1[u]][q2]; b10[u] = S[q1[u]][p2]; b11[u] = S[q1[u]][q2];
                jp0[u] = Jm[k1][p2]; jq0[u] = Jm[k1][q2]; jp1[u] = Jm[k1 + 32][p2]; jq1[u] = Jm[k1 + 32][q2];
            }
#ifdef CTM_KERNEL_CLOCKS
            __builtin_amdgcn_s_waitcnt(0); const long long c1 = clock64();
#endif
            double c2, s2; bool r2;
            jacobi_cs(a2, d2, g2, p.tol, p.tau2, p.tau_both, c2, s2, r2);
#ifdef CTM_KERNEL_CLOCKS
            __builtin_amdgcn_sched_barrier(0); const long long c2k = clock64(); __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int u = 0; u < BPT; ++u) {
                const int k1 = kb + u * KS;
                // the rotation of row pair k1 is the one the lane with k2 == k1 of this half-wave just computed
                const int src = (tid & 32) | k1;
                const double c1 = __shfl(c2, src, 64), s1 = __shfl(s2, src, 64);
                const double t00 = c2 * b00[u] - s2 * b01[u], t01 = s2 * b00[u] + c2 * b01[u];
                const double t10 = c2 * b10[u] - s2 * b11[u], t11 = s2 * b10[u] + c2 * b11[u];
                D[p1[u]][p2] = c1 * t00 - s1 * t10; D[p1[u]][q2] = c1 * t01 - s1 * t11;
                D[q1[u]][p2] = s1 * t00 + c1 * t10; D[q1[u]][q2] = s1 * t01 + c1 * t11;
                if (r2) {
                    Jm[k1][p2] = c2 * jp0[u] - s2 * jq0[u]; Jm[k1][q2] = s2 * jp0[u] + c2 * jq0[u];
                    Jm[k1 + 32][p2] = c2 * jp1[u] - s2 * jq1[u]; Jm[k1 + 32][q2] = s2 * jp1[u] + c2 * jq1[u];
                }
                if (r2 && k1 == k2) rot_flag = 1;
            }
            par ^= 1;
#ifdef CTM_KERNEL_CLOCKS
            __builtin_amdgcn_s_waitcnt(0); const long long c3 = clock64();
#endif
            __syncthreads();
#ifdef CTM_KERNEL_CLOCKS
            const long long c4 = clock64();
            ck_load += c1 - c0; ck_cs += c2k - c1; ck_upd += c3 - c2k; ck_bar += c4 - c3;
#endif
        }
        const int any = rot_flag;
        __syncthreads();
        if (!any) break;
        if (tid == 0) rot_flag = 0;
        __syncthreads();
    }
#ifdef CTM_KERNEL_CLOCKS
    if (tid == 0) { p.stat_rel[4] = ck_load; p.stat_rel[5] = ck_cs; p.stat_rel[6] = ck_upd; p.stat_rel[7] = ck_bar; }
#endif
    const double (*W)[M + 1] = Wb[par];
    if (tid < M) {
        const double d = W[tid][tid];
        int rk = 0;
        for (int j = 0; j < M; ++j) { const double dj = W[j][j]; rk += (dj > d) || (dj == d && j < tid); }
        rank_of[tid] = rk;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < EPT; ++u) { const int q = tid + u * NTH, r = q >> 6, c = q & 63; Jout[r * M + rank_of[c]] = Jm[r][c]; }
}

// ---------------------------------------------------------------------------------------------
// complex128 variant.  A panel of bc = 16 complex rows is stored as 16 real rows (real parts) followed by 16 real
// rows (imaginary parts), so a pair of panels is 64 REAL rows and the Gram / apply GEMMs of jacobi_rows() run
// unchanged on real data.  This kernel turns the 64 x 64 real Gram of such a pair into the 32 x 32 Hermitian Gram
//   G = W W^H :  Re G[a][c] = <x_a,x_c> + <y_a,y_c>,   Im G[a][c] = <y_a,x_c> - <x_a,y_c>
// diagonalises it by two-sided cyclic Jacobi with complex rotations R = diag(1, e^{-i phi}) [[c, s], [-s, c]]
// (phi = arg g_pq), and writes the unitary row transformation Q = J^H as the 64 x 64 real matrix the apply GEMM
// consumes (in[.] x out[.], real/imag rows interleaved per panel like the data).
// ---------------------------------------------------------------------------------------------
constexpr int MC = 32;    // complex Gram order (2 panels of 16)
__device__ __forceinline__ int re_row(int a) { return a + (a >= 16 ? 16 : 0); }

__global__ __launch_bounds__(256) void small_eig_c_kernel(SmallEigParams p) {
    __shared__ double Gs[2 * MC][2 * MC + 1];
    __shared__ double Wr[MC][MC + 1], Wi[MC][MC + 1], Jr[MC][MC + 1], Ji[MC][MC + 1];
    __shared__ double cs_c[MC / 2], cs_s[MC / 2], ph_r[MC / 2], ph_i[MC / 2];
    __shared__ int pr_p[MC / 2], pr_q[MC / 2];
    __shared__ double red[4];
    __shared__ int rot_flag;
    __shared__ int round_rot[2];
    __shared__ unsigned char pair_tab[MC - 1][MC / 2][2];
    __shared__ int rank_of[MC];
    const int tid = threadIdx.x, NTH = 256, mh = 2 * MC;
    const double* G = p.G + (size_t)blockIdx.x * mh * mh;
    double* Jout = p.J + (size_t)blockIdx.x * mh * mh;
    {
        double acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.0;
        for (int s = 0; s < p.nsplit; ++s) {
            const double* Gq = G + (size_t)s * p.split_stride;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] += Gq[tid + u * NTH];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int q = tid + u * NTH; Gs[q / mh][q % mh] = acc[u]; }
    }
    __syncthreads();
    for (int q = tid; q < MC * MC; q += NTH) {
        const int a = q / MC, c = q - a * MC;
        const int ra = re_row(a), rc = re_row(c);
        Wr[a][c] = Gs[ra][rc] + Gs[ra + 16][rc + 16];
        Wi[a][c] = (a == c) ? 0.0 : (Gs[ra + 16][rc] - Gs[ra][rc + 16]);
        Jr[a][c] = (a == c) ? 1.0 : 0.0; Ji[a][c] = 0.0;
    }
    __syncthreads();
    {
        double srel = 0.0, sabs = 0.0;
        for (int q = tid; q < MC * MC; q += NTH) {
            const int r = q / MC, c = q - r * MC;
            if (r < c) {
                const double g = sqrt(Wr[r][c] * Wr[r][c] + Wi[r][c] * Wi[r][c]), a = Wr[r][r], b = Wr[c][c];
                if (g > 0.0) {
                    const double sc = sqrt(fabs(a * b));
                    srel = fmax(srel, g / fmax(sc, tau_floor(a, b, p.tau2, p.tau_both)));
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
            const double v = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
            atomicMax(p.stat_rel, (unsigned long long)__double_as_longlong(v));
            red[0] = v;
        }
        __syncthreads();
        srel = red[0];
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = sabs;
        __syncthreads();
        if (tid == 0) atomicMax(p.stat_abs, (unsigned long long)__double_as_longlong(fmax(fmax(red[0], red[1]), fmax(red[2], red[3]))));
        __syncthreads();
        if (srel <= p.tol) { if (tid == 0) p.flags[blockIdx.x] = 0; return; }
        if (tid == 0) p.flags[blockIdx.x] = 1;
    }
    const int half = MC / 2, mm1 = MC - 1;
    for (int q = tid; q < mm1 * half; q += NTH) {
        const int r = q / half, k = q - r * half;
        int pi, qi;
        if (k == 0) { pi = mm1; qi = r % mm1; }
        else { pi = (r + k) % mm1; qi = (r - k + mm1) % mm1; }
        if (pi > qi) { const int t = pi; pi = qi; qi = t; }
        pair_tab[r][k][0] = (unsigned char)pi; pair_tab[r][k][1] = (unsigned char)qi;
    }
    __syncthreads();
    const int k2 = tid % half, k1 = tid / half;     // this thread owns the 2x2 block (pair k1, pair k2) and rows k1, k1+16 of J
    for (int sweep = 0; sweep < p.max_sweeps; ++sweep) {
        if (tid == 0) { rot_flag = 0; round_rot[0] = 0; }
        __syncthreads();
        for (int r = 0; r < mm1; ++r) {
            if (tid == 0) round_rot[(r + 1) & 1] = 0;
            if (tid < half) {
                const int pi = pair_tab[r][tid][0], qi = pair_tab[r][tid][1];
                const double a = Wr[pi][pi], b = Wr[qi][qi], gr = Wr[pi][qi], gi = Wi[pi][qi];
                const double g = sqrt(gr * gr + gi * gi);
                double c = 1.0, s = 0.0, er = 1.0, ei = 0.0;
                if (g != 0.0 && g > p.tol * fmax(sqrt(fabs(a * b)), tau_floor(a, b, p.tau2, p.tau_both))) {
                    er = gr / g; ei = gi / g;
                    const double d = b - a, g2 = 2.0 * g;
                    const double h = sqrt(d * d + g2 * g2);
                    const double t = g2 / (d + (d >= 0.0 ? h : -h));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                    rot_flag = 1;
                    round_rot[r & 1] = 1;
                }
                cs_c[tid] = c; cs_s[tid] = s; ph_r[tid] = er; ph_i[tid] = ei; pr_p[tid] = pi; pr_q[tid] = qi;
            }
            __syncthreads();
            if (round_rot[r & 1]) {
                const double c2 = cs_c[k2], s2 = cs_s[k2], e2r = ph_r[k2], e2i = ph_i[k2];
                const double c1 = cs_c[k1], s1 = cs_s[k1], e1r = ph_r[k1], e1i = ph_i[k1];
                const int p2 = pr_p[k2], q2 = pr_q[k2], p1 = pr_p[k1], q1 = pr_q[k1];
                // loads
                double b00r = Wr[p1][p2], b00i = Wi[p1][p2], b01r = Wr[p1][q2], b01i = Wi[p1][q2];
                double b10r = Wr[q1][p2], b10i = Wi[q1][p2], b11r = Wr[q1][q2], b11i = Wi[q1][q2];
                double jpr[2], jpi[2], jqr[2], jqi[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { const int i = k1 + u * 16; jpr[u] = Jr[i][p2]; jpi[u] = Ji[i][p2]; jqr[u] = Jr[i][q2]; jqi[u] = Ji[i][q2]; }
                // column q2 *= e^{-i phi2}
                double x;
                x = b01r * e2r + b01i * e2i; b01i = b01i * e2r - b01r * e2i; b01r = x;
                x = b11r * e2r + b11i * e2i; b11i = b11i * e2r - b11r * e2i; b11r = x;
                // column rotation
                const double t00r = c2 * b00r - s2 * b01r, t00i = c2 * b00i - s2 * b01i, t01r = s2 * b00r + c2 * b01r, t01i = s2 * b00i + c2 * b01i;
                double t10r = c2 * b10r - s2 * b11r, t10i = c2 * b10i - s2 * b11i, t11r = s2 * b10r + c2 * b11r, t11i = s2 * b10i + c2 * b11i;
                // row q1 *= e^{+i phi1}
                x = t10r * e1r - t10i * e1i; t10i = t10i * e1r + t10r * e1i; t10r = x;
                x = t11r * e1r - t11i * e1i; t11i = t11i * e1r + t11r * e1i; t11r = x;
                // row rotation + stores
                const bool dg = (k1 == k2);
                Wr[p1][p2] = c1 * t00r - s1 * t10r; Wi[p1][p2] = dg ? 0.0 : (c1 * t00i - s1 * t10i);
                Wr[p1][q2] = c1 * t01r - s1 * t11r; Wi[p1][q2] = c1 * t01i - s1 * t11i;
                Wr[q1][p2] = s1 * t00r + c1 * t10r; Wi[q1][p2] = s1 * t00i + c1 * t10i;
                Wr[q1][q2] = s1 * t01r + c1 * t11r; Wi[q1][q2] = dg ? 0.0 : (s1 * t01i + c1 * t11i);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = k1 + u * 16;
                    const double qr = jqr[u] * e2r + jqi[u] * e2i, qi_ = jqi[u] * e2r - jqr[u] * e2i;
                    Jr[i][p2] = c2 * jpr[u] - s2 * qr; Ji[i][p2] = c2 * jpi[u] - s2 * qi_;
                    Jr[i][q2] = s2 * jpr[u] + c2 * qr; Ji[i][q2] = s2 * jpi[u] + c2 * qi_;
                }
            }
            __syncthreads();
        }
        if (rot_flag == 0) break;
        __syncthreads();
    }
    if (tid < MC) {
        const double d = Wr[tid][tid];
        int rk = 0;
        for (int j = 0; j < MC; ++j) { const double dj = Wr[j][j]; rk += (dj > d) || (dj == d && j < tid); }
        rank_of[tid] = rk;
    }
    __syncthreads();
    // Jhat[in][out]: out_re(k) <- +Jr in_re(a), +Ji in_im(a) ; out_im(k) <- -Ji in_re(a), +Jr in_im(a)
    for (int q = tid; q < MC * MC; q += NTH) {
        const int a = q / MC, k = q - a * MC;
        const int ia = re_row(a), ok = re_row(rank_of[k]);
        const double jr = Jr[a][k], ji = Ji[a][k];
        Jout[(size_t)ia * mh + ok] = jr;          Jout[(size_t)(ia + 16) * mh + ok] = ji;
        Jout[(size_t)ia * mh + ok + 16] = -ji;    Jout[(size_t)(ia + 16) * mh + ok + 16] = jr;
    }
}

// per-round batched-GEMM offset tables (built on the host once per (blocks, leading dim, block size))
struct RRTables {
    int nbk = 0, b = 0, nsplit = 1, klen = 0; long long ld = 0;
    GemmOff* d_gram = nullptr;    // [rounds][pairs]
    GemmOff* d_apply = nullptr;   // [rounds][pairs]
    int* d_pairs = nullptr;       // [rounds][pairs][2]: the panels of every pair (jacobi_sweep_kernel)
};

std::map<std::string, RRTables>& tables() { static std::map<std::string, RRTables> t; return t; }
std::mutex& tables_mutex() { static std::mutex m; return m; }   // contexts of several host threads share the tables

int get_tables(ctm_ctx* ctx, int nbk, long long ld, int b, int Cg, RRTables** out) {
    const std::string key = std::to_string(ctx->device) + ":" + std::to_string(nbk) + ":" + std::to_string(ld) + ":" + std::to_string(b) + ":" + std::to_string(Cg) + ":" + std::to_string(ctx->jacobi_gram_kmin) + ":" + std::to_string(ctx->jacobi_gram_kmin_short);
    std::lock_guard<std::mutex> lock(tables_mutex());
    auto& T = tables();
    auto it = T.find(key);
    if (it != T.end()) { *out = &it->second; return CTM_OK; }
    const int rounds = nbk - 1, pairs = nbk / 2, m = 2 * b;
    // split the long K (= Cg) of the pair Grams over enough workgroups to fill the chip (>= ~512 WGs per launch)
    // (short rows -- the dense SVD of a Ritz matrix -- are split down to jacobi_gram_kmin_short columns per workgroup: the round is
    // latency bound and the eigensolver's prologue adds the partial Grams; long rows keep >= 256 columns, their partials are 32 KB each)
    const int kmin = (Cg <= 2048) ? ctx->jacobi_gram_kmin_short : ctx->jacobi_gram_kmin;
    int nsplit = std::max(1, std::min(512 / std::max(pairs, 1), Cg / std::max(16, kmin)));
    int klen = (((Cg + nsplit - 1) / nsplit) + 15) / 16 * 16;
    nsplit = (Cg + klen - 1) / klen;
    std::vector<GemmOff> gram((size_t)rounds * pairs * nsplit), app((size_t)rounds * pairs);
    std::vector<int> plist((size_t)rounds * pairs * 2);
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < pairs; ++k) {
            int i, j;
            if (k == 0) { i = nbk - 1; j = r % (nbk - 1); }
            else { i = (r + k) % (nbk - 1); j = (r - k + (nbk - 1)) % (nbk - 1); }
            if (i > j) std::swap(i, j);
            plist[((size_t)r * pairs + k) * 2] = i; plist[((size_t)r * pairs + k) * 2 + 1] = j;
            const long long oi = (long long)i * b * ld, oj = (long long)j * b * ld;
            for (int s = 0; s < nsplit; ++s) {
                const long long k0 = (long long)s * klen;
                GemmOff g; g.a0 = oi + k0; g.a1 = oj + k0; g.b0 = oi + k0; g.b1 = oj + k0;
                g.c0 = g.c1 = ((long long)s * pairs + k) * m * m;
                g.klen = (int)std::min<long long>(klen, Cg - k0);
                gram[((size_t)r * nsplit + s) * pairs + k] = g;
            }
            GemmOff a; a.a0 = a.a1 = (long long)k * m * m; a.b0 = oi; a.b1 = oj; a.c0 = oi; a.c1 = oj; a.klen = 0;
            app[(size_t)r * pairs + k] = a;
        }
    }
    RRTables t; t.nbk = nbk; t.ld = ld; t.b = b; t.nsplit = nsplit; t.klen = klen;
    if (hipMalloc(&t.d_gram, gram.size() * sizeof(GemmOff)) != hipSuccess ||
        hipMalloc(&t.d_apply, app.size() * sizeof(GemmOff)) != hipSuccess ||
        hipMalloc(&t.d_pairs, plist.size() * sizeof(int)) != hipSuccess) {
        ctx->set_error("jacobi: table alloc"); return CTM_ERR_NOMEM;
    }
    CTM_HIP_CHECK(ctx, hipMemcpy(t.d_pairs, plist.data(), plist.size() * sizeof(int), hipMemcpyHostToDevice));
    CTM_HIP_CHECK(ctx, hipMemcpy(t.d_gram, gram.data(), gram.size() * sizeof(GemmOff), hipMemcpyHostToDevice));
    CTM_HIP_CHECK(ctx, hipMemcpy(t.d_apply, app.data(), app.size() * sizeof(GemmOff), hipMemcpyHostToDevice));
    T[key] = t;
    *out = &T[key];
    return CTM_OK;
}

// Core: orthogonalise the rows of the R x Cg matrix held in the first Cg columns of X (R x ld, row-major),
// applying the same row rotations to ALL Ctot columns (the extra columns carry the accumulated left factor
// or any companion basis).  In place: each workgroup of the apply GEMM owns a column strip of all 2b rows of
// its pair and finishes reading it before it writes.
// ktop > 0: only the ktop largest rows need full relative accuracy -- pairs of smaller rows are measured against
// tau = (ktop-th largest row norm), which still bounds the spectral norm of the remaining rows by tau(1 + R tol).
// null_rel > 0 (full decompositions whose numerically null rows are rebuilt afterwards, see svd_full): pairs of rows that are BOTH below
// null_rel x the largest row norm are not rotated against each other -- such rows are rounding noise of the big ones, their mutual
// overlaps never settle (every rotation with a big row re-injects noise of their own size) and the accumulated rotations stay
// orthogonal whether or not they are touched.
int jacobi_rows(ctm_ctx* ctx, double* X, int R, long long ld, int Cg, int Ctot, int b, int ktop, double fro, int max_sweeps, bool cplx = false,
                bool tau_both = false, double null_rel = 0.0) {
    const int nbk = R / b, rounds = nbk - 1, pairs = nbk / 2, m = 2 * b;
    if (cplx && b != 32) { ctx->set_error("jacobi_rows: complex panels are 16 + 16 real rows"); return CTM_ERR_BADARG; }
    RRTables* T;
    CTM_TRY(get_tables(ctx, nbk, ld, b, Cg, &T));
    ArenaScope scope(ctx);
    double *G, *J, *norms;
    int* flags;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)T->nsplit * pairs * m * m, (void**)&G));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * pairs * m * m, (void**)&J));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * R, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * pairs, (void**)&flags));
    unsigned long long* stat = (unsigned long long*)ctx->d_scratch;   // [0]=scaled, [1]=classical
    std::vector<double> h(R);
    const double floor2 = (1e-14 * fro) * (1e-14 * fro);
    ctx->last_sweeps = 0;
    bool abs_mode = false;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        double tau2 = floor2;
        if (ctx->jacobi_tau_relax && ktop > 0 && ktop < (cplx ? R / 2 : R)) {
            CTM_TRY(row_norms(ctx, X, R, Cg, ld, norms));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * R, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            int nr = R;
            if (cplx) {      // squared norm of a complex row = its real-part row + its imaginary-part row (16 + 16 per panel)
                nr = R / 2;
                std::vector<double> hc(nr);
                for (int cr = 0; cr < nr; ++cr) { const int rr = (cr / 16) * 32 + (cr % 16); hc[cr] = std::sqrt(h[rr] * h[rr] + h[rr + 16] * h[rr + 16]); }
                std::copy(hc.begin(), hc.end(), h.begin());
            }
            std::nth_element(h.begin(), h.begin() + (ktop - 1), h.begin() + nr, std::greater<double>());
            tau2 = std::max(floor2, h[ktop - 1] * h[ktop - 1]);
        }
        if (null_rel > 0.0) {
            CTM_TRY(row_norms(ctx, X, R, Cg, ld, norms));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * R, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            double mx2 = 0.0;
            if (cplx) { for (int cr = 0; cr < R / 2; ++cr) { const int rr = (cr / 16) * 32 + (cr % 16); mx2 = std::max(mx2, h[rr] * h[rr] + h[rr + 16] * h[rr + 16]); } }
            else for (int i = 0; i < R; ++i) mx2 = std::max(mx2, h[i] * h[i]);
            tau2 = std::max(tau2, null_rel * null_rel * mx2);
            tau_both = true;
            if (ctx->svd_abs_accuracy) { tau2 = std::sqrt(mx2); abs_mode = true; }
        } else if (ctx->force_abs && !cplx) {
            // experiment knob (lz_abs_accuracy): the Ritz extraction of the block Krylov solver with the absolute criterion
            CTM_TRY(row_norms(ctx, X, R, Cg, ld, norms));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * R, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            tau2 = *std::max_element(h.begin(), h.begin() + R); abs_mode = true;
        }
        CTM_HIP_CHECK(ctx, hipMemsetAsync(stat, 0, 2 * sizeof(double), ctx->stream));
        for (int r = 0; r < rounds; ++r) {
            GemmDesc g;
            g.M = m; g.N = m; g.K = Cg;
            g.A = X; g.sam = ld; g.sak = 1; g.splitA = b;
            g.B = X; g.sbk = 1; g.sbn = ld; g.splitB = b; g.splitB_dim = 2;
            g.C = G; g.ldc = m;
            g.batch = pairs * T->nsplit; g.offs = T->d_gram + (size_t)r * pairs * T->nsplit;
            CTM_TRY(gemm_f64(ctx, g));
            SmallEigParams sp;
            sp.G = G; sp.nsplit = T->nsplit; sp.split_stride = (long long)pairs * m * m; sp.J = J; sp.m = m; sp.tol = ctx->jacobi_tol * 0.1; sp.max_sweeps = (pairs == 1) ? 12 : (pairs >= 4 ? ctx->jacobi_inner_sweeps_many : ctx->jacobi_inner_sweeps);
            sp.tau2 = tau2; sp.tau_both = abs_mode ? 2 : (tau_both ? 1 : 0); sp.stat_rel = stat; sp.stat_abs = stat + 1; sp.flags = flags;
            // cross-only rotations in every round but the first of a sweep (which pairs every panel once and solves the full 64 x 64
            // problems: the intra-panel pairs); many-panel problems only (the dense SVD of a Ritz matrix, full-block Rayleigh-Ritz)
            sp.cross = (ctx->jacobi_cross_only && !cplx && m == 64 && pairs >= 4 && r > 0) ? 1 : 0;
            if (cplx) CTM_LAUNCH(ctx, small_eig_c_kernel, dim3(pairs), dim3(256), 0, sp);
            else if (m == 64) {
                if (sp.cross) CTM_LAUNCH(ctx, (small_eig64_kernel<2, true>), dim3(pairs), dim3(512), 0, sp);
                else CTM_LAUNCH(ctx, small_eig64_kernel<2>, dim3(pairs), dim3(512), 0, sp);
            }
            else if (m > 32) CTM_LAUNCH(ctx, small_eig_kernel<64>, dim3(pairs), dim3(1024), 0, sp);
            else CTM_LAUNCH(ctx, small_eig_kernel<32>, dim3(pairs), dim3(256), 0, sp);
            GemmDesc a;
            a.M = m; a.N = Ctot; a.K = m;
            a.A = J; a.sam = 1; a.sak = m;                       // J^T
            a.B = X; a.sbk = ld; a.sbn = 1; a.splitB = b; a.splitB_dim = 1;
            a.C = X; a.ldc = ld; a.splitC = b;
            a.batch = pairs; a.offs = T->d_apply + (size_t)r * pairs; a.skip_flags = flags;
            CTM_TRY(gemm_f64(ctx, a));
        }
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scratch, stat, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const double srel = ctx->h_scratch[0];
        ctx->last_sweeps = sweep + 1;
        ctx->last_offnorm = srel;
        if (ctx->jacobi_verbose > 1) fprintf(stderr, "[jacobi] R=%d sweep %d  scaled=%.3e classical=%.3e tau=%.3e\n", R, sweep + 1, srel, ctx->h_scratch[1], std::sqrt(tau2));
        if (srel <= ctx->jacobi_tol) break;
        // `srel` is the measure of the Gram matrices the sweep FOUND; in the quadratic regime the sweep leaves ~ srel^2 / gap.  A caller
        // that verifies the result itself (the Ritz extraction of the block Krylov solver: residuals of both relations on the returned
        // triplets) does not pay for a twelfth sweep that finds 1e-15 and rotates nothing
        if (ctx->jacobi_quad_exit > 0.0 && srel <= ctx->jacobi_quad_exit) break;
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

double host_fro(ctm_ctx* ctx, const double* M, int rows, int cols, long long ld, double* d_tmp, std::vector<double>& h, int* status) {
    *status = row_norms(ctx, M, rows, cols, ld, d_tmp);
    if (*status != CTM_OK) return 0.0;
    h.resize(rows);
    if (hipMemcpyAsync(h.data(), d_tmp, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { *status = CTM_ERR_HIP; return 0.0; }
    double f = 0.0;
    for (int i = 0; i < rows; ++i) f += h[i] * h[i];
    return std::sqrt(f);
}

// ---------------------------------------------------------------------------------------------
// full decomposition: every row pair is orthogonalised (O(n^3) per sweep)
// ---------------------------------------------------------------------------------------------
// Rows kg .. k-1 of the k x n row matrix Vt (orthonormal rows expected) <- an orthonormal basis of the orthogonal complement of rows
// 0 .. kg-1 (or of part of it when k < n).  See svd_full().
int complete_null_rows(ctm_ctx* ctx, double* Vt, int kg, int k, int n) {
    ArenaScope cscope(ctx);
        CTM_TRY(reorth_rows(ctx, Vt, kg, n, n, 2));
        double *Pm, *Dn, *Wn;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * n, (void**)&Pm));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (k - kg), (void**)&Dn));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(k - kg) * n, (void**)&Wn));
        CTM_TRY(set_identity(ctx, Pm, n, n));
        GemmDesc gp; gp.M = n; gp.N = n; gp.K = kg; gp.A = Vt; gp.sam = 1; gp.sak = n; gp.B = Vt; gp.sbk = n; gp.sbn = 1; gp.C = Pm; gp.ldc = n;
        gp.alpha = -1.0; gp.beta = 1.0;
        CTM_TRY(gemm_f64(ctx, gp));
        // Full complement (k == n): the n - kg rows of the projector with the largest norm (|P e_j|^2 = P_jj: pivoting) span it unless
        // they happen to be dependent; a row Jacobi on those m rows (m^2 n work instead of the n^3 of the projector's
        // eigendecomposition) orthogonalises them, and row norms that stay O(1) certify the span.  Otherwise: eigenvectors.
        bool done = false;
        const int m = k - kg;
        if (k == n) {
            std::vector<double> pd(n);
            CTM_HIP_CHECK(ctx, hipMemcpy2DAsync(pd.data(), sizeof(double), Pm, sizeof(double) * ((size_t)n + 1), sizeof(double), n,
                                                hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            std::vector<int> jd(n);
            std::iota(jd.begin(), jd.end(), 0);
            std::stable_sort(jd.begin(), jd.end(), [&](int a, int c) { return pd[a] > pd[c]; });
            const int b2 = choose_block(ctx, n), mp = padded(m, b2);
            ArenaScope zs(ctx);
            double *Z, *zn;
            int* dj;
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)mp * n, (void**)&Z));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * mp, (void**)&zn));
            CTM_TRY(arena_alloc(ctx, sizeof(int) * mp, (void**)&dj));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(dj, jd.data(), sizeof(int) * m, hipMemcpyHostToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            CTM_TRY(fill_f64(ctx, Z, (size_t)mp * n, 0.0));
            CTM_TRY(gather_rows(ctx, Pm, n, dj, m, n, Z, n, nullptr));
            // orthonormalise the m rows: Newton-Schulz iteration for the polar factor, Z <- Z - (Z Z^T - I) Z / 2 (two small GEMMs per
            // step; singular values of the pivoted rows lie in (0, 1], each step moves them towards 1, quadratically at the end)
            double *G2, *Z2;
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)m * m, (void**)&G2));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)mp * n, (void**)&Z2));
            std::vector<double> hz(m);
            double dev = 1.0;
            for (int it = 0; it < 48; ++it) {
                GemmDesc gg; gg.M = m; gg.N = m; gg.K = n; gg.A = Z; gg.sam = n; gg.sak = 1; gg.B = Z; gg.sbk = 1; gg.sbn = n; gg.C = G2; gg.ldc = m;
                CTM_TRY(gemm_f64(ctx, gg));
                CTM_LAUNCH(ctx, sub_eye_kernel, dim3((m + 255) / 256), dim3(256), 0, G2, m);
                CTM_TRY(row_norms(ctx, G2, m, m, m, zn));
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(hz.data(), zn, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
                CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                const double prev = dev;
                dev = *std::max_element(hz.begin(), hz.end());
                if (!(dev == dev) || dev <= 1e-13 || (it > 0 && dev < 1e-10 && dev > 0.5 * prev)) break;    // converged / at the rounding floor
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(Z2, Z, sizeof(double) * (size_t)m * n, hipMemcpyDeviceToDevice, ctx->stream));
                GemmDesc gz; gz.M = m; gz.N = n; gz.K = m; gz.A = G2; gz.sam = m; gz.sak = 1; gz.B = Z; gz.sbk = n; gz.sbn = 1; gz.C = Z2; gz.ldc = n;
                gz.alpha = -0.5; gz.beta = 1.0;
                CTM_TRY(gemm_f64(ctx, gz));
                std::swap(Z, Z2);
            }
            if (dev == dev && dev <= 1e-10) {
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vt + (size_t)kg * n, Z, sizeof(double) * (size_t)m * n, hipMemcpyDeviceToDevice, ctx->stream));
                CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                done = true; ctx->svd_polar_completions += 1;
            }
        }
        if (!done) {
            ctx->svd_eig_completions += 1;
            const bool save = ctx->si_enable; ctx->si_enable = false;      // a projector's spectrum is flat: the leading-k iteration cannot converge on it
            const int st3 = jacobi_eigh_top(ctx, Pm, n, m, Dn, Wn, nullptr);
            ctx->si_enable = save;
            CTM_TRY(st3);
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vt + (size_t)kg * n, Wn, sizeof(double) * (size_t)m * n, hipMemcpyDeviceToDevice, ctx->stream));
        }
        CTM_TRY(reorth_rows(ctx, Vt, k, n, n, done ? 2 : 1));
    return CTM_OK;
}

// warm (optional, k == n only): n x n workspace with the left vectors u_i^T of the previous decomposition of a nearby matrix.  The rows
// of W M are then almost orthogonal already and the sweeps start in the quadratically convergent regime (the differentiable route
// of an optimisation decomposes the same sequence of matrices again and again); any orthonormal W is a valid start.  Updated.
int svd_full(ctm_ctx* ctx, const double* M, int n, int k, double* S, double* Ut, double* Vt, double* warm = nullptr) {
    const int b = choose_block(ctx, n), np = padded(n, b);
    ArenaScope scope(ctx);
    const bool with_q = (Ut != nullptr);
    const long long ld = (long long)n + (with_q ? np : 0);
    double *X, *norms;
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)np * ld, (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * np, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * np, (void**)&d_idx));
    std::vector<double> h;
    int st;
    bool warm_full = false;
    if (warm && with_q && k == n && ctx->eigh_warm) {
        const double fw = host_fro(ctx, warm, n, n, n, norms, h, &st);
        CTM_TRY(st);
        warm_full = std::fabs(fw - std::sqrt((double)n)) <= 1e-6 * std::sqrt((double)n);
        for (int i = 0; warm_full && i < n; ++i) warm_full = std::fabs(h[i] - 1.0) <= 1e-6;
    }
    if (warm_full) {
        CTM_TRY(fill_f64(ctx, X, (size_t)np * ld, 0.0));
        GemmDesc gw; gw.M = n; gw.N = n; gw.K = n; gw.A = warm; gw.sam = n; gw.sak = 1; gw.B = M; gw.sbk = n; gw.sbn = 1; gw.C = X; gw.ldc = ld;
        CTM_TRY(gemm_f64(ctx, gw));
        CTM_TRY(copy2d(ctx, warm, n, X + n, ld, n, n));
        ctx->eigh_warm_hits += 1;
    } else
        CTM_LAUNCH(ctx, fill_wq_kernel, dim3(2048), dim3(256), 0, M, n, n, (long long)n, X, np, ld, with_q ? 1 : 0);
    const double fro = host_fro(ctx, X, np, n, ld, norms, h, &st);
    CTM_TRY(st);
    // full decomposition with vectors: rows below 0.1 svd_null_tol s_0 are rebuilt as an orthonormal complement below anyway
    const double null_rel = (Ut && Vt && k == n) ? 0.1 * ctx->svd_null_tol : 0.0;
    CTM_TRY(jacobi_rows(ctx, X, np, ld, n, (int)ld, b, (k < n) ? k : 0, fro, ctx->jacobi_max_sweeps, false, false, null_rel));
    CTM_TRY(row_norms(ctx, X, np, n, ld, norms));
    h.resize(np);
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
    if (!Ut) return CTM_OK;   // singular values only: row norms of the converged W
    // U = accumulated rotations (orthonormalised against drift); then Sigma V^T = U^T M is recomputed by one
    // k x n x n GEMM so that S and V carry no accumulated rounding of the sweeps (|error| = O(eps |M|)).
    CTM_TRY(gather_rows(ctx, X + n, ld, d_idx, k, n, Ut, n, nullptr));
    CTM_TRY(reorth_rows(ctx, Ut, k, n, n, 2));
    if (warm && k == n) CTM_HIP_CHECK(ctx, hipMemcpyAsync(warm, Ut, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream));
    if (Vt) {
        double* inv;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&inv));
        GemmDesc g; g.M = k; g.N = n; g.K = n; g.A = Ut; g.sam = n; g.sak = 1; g.B = M; g.sbk = n; g.sbn = 1; g.C = Vt; g.ldc = n;
        CTM_TRY(gemm_f64(ctx, g));
        CTM_TRY(row_norms(ctx, Vt, k, n, n, S));
        CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3((k + 255) / 256), dim3(256), 0, S, inv, k);
        CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, Vt, k, n, (long long)n, inv);
        // Rows whose singular value sits at the rounding level of M (s_i <= null_tol s_0): U^T M is noise there and the 1/s scaling
        // turns it into O(1) garbage that the re-orthonormalisation cannot repair.  Like LAPACK, return an ORTHONORMAL V for a
        // rank-deficient matrix: those rows become an orthonormal basis of the orthogonal complement of the others (eigenvectors
        // with eigenvalue 1 of the projector 1 - Vg^T Vg).  The differentiable full decomposition (linalg/svd_gesdd.py) needs it.
        int kg = k;
        while (kg > 0 && !(hs[kg - 1] > ctx->svd_null_tol * hs[0])) --kg;
        if (kg > 0 && kg < k) {
            CTM_TRY(complete_null_rows(ctx, Vt, kg, k, n));
        } else
            CTM_TRY(reorth_rows(ctx, Vt, k, n, n, 2));
    }
    return CTM_OK;
}

__global__ void sym_avg_kernel(double* H, int n) {
    const size_t tot = (size_t)n * n;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const size_t r = q / n, c = q - r * n;
        if (r < c) { const double v = 0.5 * (H[r * n + c] + H[c * n + r]); H[r * n + c] = v; H[c * n + r] = v; }
    }
}

// Full SVD with vectors of a real n x n matrix through its polar decomposition (differentiable route; `svd_polar`):
//   X_0 = M / |M|_F,  X <- X - (X X^T - I) X / 2   (Newton-Schulz: X keeps M's singular vectors, its singular values go to 1 -- a value
//                                                   s needs log_1.5(|M|_F / s) steps, so 80 steps resolve everything above 1e-14 |M|_F)
//   H = X^T M = V S V^T  (symmetric positive semi-definite),  eigenvectors by the one-sided Jacobi on H + shift I -- a matrix WITHOUT small
//   singular values, where the sweeps converge in a handful (and in 3-4 from the previous call's eigenvectors, `warm`), instead of
//   the ~20 sweeps the row Jacobi needs on M itself when its spectrum is graded over many orders of magnitude;
//   u_i = M v_i / |M v_i|, rows at the rounding level completed orthonormally (complete_null_rows).
// Accuracy: absolute, eps |M| on values and on U S V^T = M (what a bidiagonalisation-based SVD delivers).
int svd_full_polar(ctm_ctx* ctx, const double* M, int n, double* S, double* Ut, double* Vt, double* warm) {
    ArenaScope scope(ctx);
    const size_t nn = (size_t)n * n;
    double *X, *X2, *E, *H, *Dv, *norms, *inv;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * nn, (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * nn, (void**)&X2));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * nn, (void**)&E));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * nn, (void**)&H));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * n, (void**)&Dv));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * n, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * n, (void**)&inv));
    std::vector<double> h;
    int st;
    const double fro = host_fro(ctx, M, n, n, n, norms, h, &st);
    CTM_TRY(st);
    if (!(fro > 0.0)) return svd_full(ctx, M, n, n, S, Ut, Vt, nullptr);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(X, M, sizeof(double) * nn, hipMemcpyDeviceToDevice, ctx->stream));
    CTM_LAUNCH(ctx, axpy_kernel, dim3(1024), dim3(256), 0, X, (const double*)M, 1.0 / fro - 1.0, nn);       // X = M / fro
    for (int it = 0; it < 80; ++it) {
        GemmDesc ge; ge.M = n; ge.N = n; ge.K = n; ge.A = X; ge.sam = n; ge.sak = 1; ge.B = X; ge.sbk = 1; ge.sbn = n; ge.C = E; ge.ldc = n;
        CTM_TRY(gemm_f64(ctx, ge));                                                                       // X X^T
        CTM_LAUNCH(ctx, sub_eye_kernel, dim3((n + 255) / 256), dim3(256), 0, E, n);
        if (it % 8 == 7) {                                   // a full-rank, well conditioned matrix is done early
            CTM_TRY(row_norms(ctx, E, n, n, n, norms));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (*std::max_element(h.begin(), h.begin() + n) <= 1e-13) break;
        }
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(X2, X, sizeof(double) * nn, hipMemcpyDeviceToDevice, ctx->stream));
        GemmDesc gx; gx.M = n; gx.N = n; gx.K = n; gx.A = E; gx.sam = n; gx.sak = 1; gx.B = X; gx.sbk = n; gx.sbn = 1; gx.C = X2; gx.ldc = n;
        gx.alpha = -0.5; gx.beta = 1.0;
        CTM_TRY(gemm_f64(ctx, gx));
        std::swap(X, X2);
    }
    GemmDesc gh; gh.M = n; gh.N = n; gh.K = n; gh.A = X; gh.sam = 1; gh.sak = n; gh.B = M; gh.sbk = n; gh.sbn = 1; gh.C = H; gh.ldc = n;       // H = X^T M
    CTM_TRY(gemm_f64(ctx, gh));
    CTM_LAUNCH(ctx, sym_avg_kernel, dim3(1024), dim3(256), 0, H, n);
    CTM_TRY(jacobi_eigh_top(ctx, H, n, n, Dv, Vt, warm));                  // rows of Vt: eigenvectors, ordered by |eigenvalue| descending
    // U rows: (M v_i)^T, their norms are the singular values
    GemmDesc gu; gu.M = n; gu.N = n; gu.K = n; gu.A = Vt; gu.sam = n; gu.sak = 1; gu.B = M; gu.sbk = 1; gu.sbn = n; gu.C = Ut; gu.ldc = n;       // Vt M^T
    CTM_TRY(gemm_f64(ctx, gu));
    CTM_TRY(row_norms(ctx, Ut, n, n, n, S));
    h.resize(n);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), S, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // order by the singular values (the eigenvalue order can differ among values that agree to rounding)
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return h[a] > h[c]; });
    bool sorted = true;
    for (int i = 0; i < n; ++i) sorted = sorted && idx[i] == i;
    std::vector<double> hs(n);
    for (int i = 0; i < n; ++i) hs[i] = h[idx[i]];
    if (!sorted) {
        int* d_idx;
        CTM_TRY(arena_alloc(ctx, sizeof(int) * n, (void**)&d_idx));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, idx.data(), sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        CTM_TRY(gather_rows(ctx, Ut, n, d_idx, n, n, X, n, nullptr));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Ut, X, sizeof(double) * nn, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_TRY(gather_rows(ctx, Vt, n, d_idx, n, n, X, n, nullptr));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vt, X, sizeof(double) * nn, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, hs.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3((n + 255) / 256), dim3(256), 0, S, inv, n);
    CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, Ut, n, n, (long long)n, inv);
    int kg = n;
    while (kg > 0 && !(hs[kg - 1] > ctx->svd_null_tol * hs[0])) --kg;
    if (kg > 0 && kg < n) CTM_TRY(complete_null_rows(ctx, Ut, kg, n, n));
    else CTM_TRY(reorth_rows(ctx, Ut, n, n, n, 2));
    ctx->svd_polar_solves += 1;
    return CTM_OK;
}

// Y (p x n, ldy) = X (p x n, ldx) * op(Z) for a dense n x n matrix Z (row-major): op = 'N' or 'T'
int rows_times(ctm_ctx* ctx, const double* X, long long ldx, int p, int kin, int nout, const double* Z, bool transZ, double* Y, long long ldy) {
    // Y (p x nout) = X (p x kin) op(Z); Z is stored kin x nout (transZ == false) or nout x kin (transZ == true)
    GemmDesc g; g.M = p; g.N = nout; g.K = kin; g.A = X; g.sam = ldx; g.sak = 1; g.B = Z;
    if (transZ) { g.sbk = 1; g.sbn = kin; } else { g.sbk = nout; g.sbn = 1; }
    g.C = Y; g.ldc = ldy;
    return gemm_f64(ctx, g);
}

// C = B * M (transpose == false) or B * M^T (transpose == true) for the operator M of `op`
// mid (optional, p x n, leading dimension n; implicit operators only): receives the half-way product B R^T (transpose == false)
// or B Rt^T (transpose == true)
int matop_apply(ctm_ctx* ctx, const MatOp& op, bool transpose, const double* B, long long ldb, int p, double* C, long long ldc,
                double* mid = nullptr) {
    const int n = op.n;
    if (op.M) return rows_times(ctx, B, ldb, p, n, n, op.M, transpose, C, ldc);
    // implicit M = R^T Rt,  R = opA(cA) opB(cB) (n x m0 x n),  Rt = opC(cC) opD(cD) (n x m1 x n)
    const int m0 = op.mid[0] ? op.mid[0] : n, m1 = op.mid[1] ? op.mid[1] : n, mw = std::max(n, std::max(m0, m1));
    ArenaScope scope(ctx);
    double *t1, *t2 = mid;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * mw, (void**)&t1));
    if (!t2) CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)p * mw, (void**)&t2));
    if (!transpose) {   // B R^T Rt = ((B opB(cB)^T) opA(cA)^T) opC(cC) opD(cD)
        CTM_TRY(rows_times(ctx, B, ldb, p, n, m0, op.c[1], !op.t[1], t1, m0));
        CTM_TRY(rows_times(ctx, t1, m0, p, m0, n, op.c[0], !op.t[0], t2, n));
        CTM_TRY(rows_times(ctx, t2, n, p, n, m1, op.c[2], op.t[2], t1, m1));
        return rows_times(ctx, t1, m1, p, m1, n, op.c[3], op.t[3], C, ldc);
    }
    // B Rt^T R = ((B opD(cD)^T) opC(cC)^T) opA(cA) opB(cB)
    CTM_TRY(rows_times(ctx, B, ldb, p, n, m1, op.c[3], !op.t[3], t1, m1));
    CTM_TRY(rows_times(ctx, t1, m1, p, m1, n, op.c[2], !op.t[2], t2, n));
    CTM_TRY(rows_times(ctx, t2, n, p, n, m0, op.c[0], op.t[0], t1, m0));
    return rows_times(ctx, t1, m0, p, m0, n, op.c[1], op.t[1], C, ldc);
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
       HDR_DIST = 5, HDR_SIDE = 6, HDR_RUN = 7, HDR_SSKIP = 8, HDR_SFAILS = 9, HDR_WORDS = 10,
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
int svd_iter(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged, bool* want_krylov = nullptr) {
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

// row norms of the np complex rows of a panel matrix -> host (hc[np]); `norms` is device scratch of 2*np doubles
int panel_row_norms(ctm_ctx* ctx, const double* X, int np, int cols, long long ld, double* norms, std::vector<double>& hc) {
    std::vector<double> h(2 * np);
    CTM_TRY(row_norms(ctx, X, 2 * np, cols, ld, norms));
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), norms, sizeof(double) * 2 * np, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    hc.resize(np);
    for (int cr = 0; cr < np; ++cr) { const int rr = crow_re(cr); hc[cr] = std::sqrt(h[rr] * h[rr] + h[rr + BC] * h[rr + BC]); }
    return CTM_OK;
}

// gather complex rows idx[0..k) of a panel matrix (column window starting at X) into planar out (re plane, im plane = re + k*cols)
int panel_gather(ctm_ctx* ctx, const double* X, long long ld, const std::vector<int>& idx, int k, int cols, double* out, int* d_idx2) {
    std::vector<int> ir(2 * k);
    for (int i = 0; i < k; ++i) { ir[i] = crow_re(idx[i]); ir[k + i] = ir[i] + BC; }
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(d_idx2, ir.data(), sizeof(int) * 2 * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // the index list is [re rows..., im rows...] and the planes are adjacent: ONE gather fills both
    return gather_rows(ctx, X, ld, d_idx2, 2 * k, cols, out, cols, nullptr);
}

int scale_planar_rows(ctm_ctx* ctx, double* V, int k, int n, const double* inv) {
    CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, V, k, n, (long long)n, inv);
    CTM_LAUNCH(ctx, scale_rows_kernel, dim3(1024), dim3(256), 0, V + (size_t)k * n, k, n, (long long)n, inv);
    return CTM_OK;
}

// re-orthonormalise the rows of planar V (k x n complex, planes k*n apart) against the rows above them
int reorth_rows_c(ctm_ctx* ctx, double* V, int k, int n, int iters) {
    ArenaScope scope(ctx);
    double *E, *tmp;
    const size_t kn = (size_t)k * n;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * k * k, (void**)&E));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * kn, (void**)&tmp));
    for (int it = 0; it < iters; ++it) {
        XM a{V, V + kn, n, false, false}, bh{V, V + kn, n, true, true};
        CTM_TRY(xgemm(ctx, k, k, n, a, bh, E, E + (size_t)k * k, k));              // E = V V^H
        CTM_TRY(tril_correction_c128(ctx, E, E + (size_t)k * k, k));
        XM e{E, E + (size_t)k * k, k, false, false};
        CTM_TRY(xgemm(ctx, k, n, k, e, a, tmp, tmp + kn, n));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(V, tmp, sizeof(double) * 2 * kn, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return CTM_OK;
}

int svd_full_c(ctm_ctx* ctx, const double* Mr, const double* Mi, int n, int k, double* S, double* Ut, double* Vt, double* warm = nullptr) {
    const int np = padded(n, BC);
    ArenaScope scope(ctx);
    const bool with_q = (Ut != nullptr);
    const long long ld = (long long)n + (with_q ? np : 0);
    const size_t nn = (size_t)n * n;
    double *X, *norms;
    int* d_idx;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)2 * np * ld, (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * np, (void**)&norms));
    CTM_TRY(arena_alloc(ctx, sizeof(int) * 2 * np, (void**)&d_idx));
    std::vector<double> h;
    int st;
    bool warm_full = false;          // warm start of the full decomposition (planar rows u_i^H): see svd_full()
    if (warm && with_q && k == n && ctx->eigh_warm) {
        const double fw = host_fro(ctx, warm, 2 * n, n, n, norms, h, &st);
        CTM_TRY(st);
        warm_full = std::fabs(fw - std::sqrt((double)n)) <= 1e-6 * std::sqrt((double)n);
        for (int i = 0; warm_full && i < n; ++i) warm_full = std::fabs(std::sqrt(h[i] * h[i] + h[n + i] * h[n + i]) - 1.0) <= 1e-6;
    }
    if (warm_full) {
        ArenaScope ws(ctx);
        double* Yw;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * nn, (void**)&Yw));
        XM w{warm, warm + nn, n, false, false}, m{Mr, Mi, n, false, false};
        CTM_TRY(xgemm(ctx, n, n, n, w, m, Yw, Yw + nn, n));
        CTM_LAUNCH(ctx, fill_wq_c2_kernel, dim3(2048), dim3(256), 0, (const double*)Yw, (const double*)(Yw + nn), (const double*)warm,
                   (const double*)(warm + nn), n, X, np, ld);
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->eigh_warm_hits += 1;
    } else
        CTM_LAUNCH(ctx, fill_wq_c_kernel, dim3(2048), dim3(256), 0, Mr, Mi, n, X, np, ld, with_q ? 1 : 0);
    const double fro = host_fro(ctx, X, 2 * np, n, ld, norms, h, &st);
    CTM_TRY(st);
    const double null_rel = (Ut && Vt && k == n) ? 0.1 * ctx->svd_null_tol : 0.0;       // see svd_full()
    CTM_TRY(jacobi_rows(ctx, X, 2 * np, ld, n, (int)ld, 2 * BC, (k < n) ? k : 0, fro, ctx->jacobi_max_sweeps, true, false, null_rel));
    std::vector<double> hc;
    CTM_TRY(panel_row_norms(ctx, X, np, n, ld, norms, hc));
    std::vector<int> idx(np);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return hc[a] > hc[c]; });
    std::vector<double> hs(k);
    for (int i = 0; i < k; ++i) hs[i] = hc[idx[i]];
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, hs.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (!Ut) return CTM_OK;
    // rows of the accumulated unitary Q are u_k^H; Sigma V^H = Q M is recomputed by one k x n x n product (drift-free)
    CTM_TRY(panel_gather(ctx, X + n, ld, idx, k, n, Ut, d_idx));
    CTM_TRY(reorth_rows_c(ctx, Ut, k, n, 2));
    if (warm && k == n) CTM_HIP_CHECK(ctx, hipMemcpyAsync(warm, Ut, sizeof(double) * 2 * nn, hipMemcpyDeviceToDevice, ctx->stream));
    if (Vt) {
        const size_t kn = (size_t)k * n;
        double* inv;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&inv));
        XM u{Ut, Ut + kn, n, false, false}, m{Mr, Mi, n, false, false};
        CTM_TRY(xgemm(ctx, k, n, n, u, m, Vt, Vt + kn, n));
        CTM_TRY(row_norms_c128(ctx, Vt, Vt + kn, k, n, n, S));
        CTM_LAUNCH(ctx, inv_or_zero_kernel, dim3((k + 255) / 256), dim3(256), 0, S, inv, k);
        CTM_TRY(scale_planar_rows(ctx, Vt, k, n, inv));
        // rows at the rounding level of M: orthonormal basis of the complement of the others (see svd_full); rows are v_i^H, so the
        // projector is 1 - A^H A with A = the good rows
        int kg = k;
        while (kg > 0 && !(hs[kg - 1] > ctx->svd_null_tol * hs[0])) --kg;
        if (kg > 0 && kg < k) {
            const int kb = k - kg;
            double *Pr, *Pi, *Dn, *Wn, *Vg;
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * n, (void**)&Pr));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * n, (void**)&Pi));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * kb, (void**)&Dn));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)kb * n, (void**)&Wn));
            CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)kg * n, (void**)&Vg));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vg, Vt, sizeof(double) * (size_t)kg * n, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vg + (size_t)kg * n, Vt + kn, sizeof(double) * (size_t)kg * n, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_TRY(reorth_rows_c(ctx, Vg, kg, n, 2));
            XM ah{Vg, Vg + (size_t)kg * n, n, true, true}, a{Vg, Vg + (size_t)kg * n, n, false, false};
            CTM_TRY(xgemm(ctx, n, n, kg, ah, a, Pr, Pi, n));
            CTM_LAUNCH(ctx, eye_minus_kernel, dim3(1024), dim3(256), 0, Pr, Pi, n);
            // full complement (k == n): pivoted projector rows + Newton-Schulz polar iteration, as in svd_full(); rows here are v^H (planar)
            bool done = false;
            if (k == n) {
                std::vector<double> pd(n);
                CTM_HIP_CHECK(ctx, hipMemcpy2DAsync(pd.data(), sizeof(double), Pr, sizeof(double) * ((size_t)n + 1), sizeof(double), n,
                                                    hipMemcpyDeviceToHost, ctx->stream));
                CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                std::vector<int> jd(n);
                std::iota(jd.begin(), jd.end(), 0);
                std::stable_sort(jd.begin(), jd.end(), [&](int a, int c) { return pd[a] > pd[c]; });
                const int m = kb;
                ArenaScope zs(ctx);
                double *Z, *Z2, *G2, *zn;
                int* dj;
                const size_t mn = (size_t)m * n, mm = (size_t)m * m;
                CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * mn, (void**)&Z));
                CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * mn, (void**)&Z2));
                CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * mm, (void**)&G2));
                CTM_TRY(arena_alloc(ctx, sizeof(double) * m, (void**)&zn));
                CTM_TRY(arena_alloc(ctx, sizeof(int) * m, (void**)&dj));
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(dj, jd.data(), sizeof(int) * m, hipMemcpyHostToDevice, ctx->stream));
                CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                CTM_TRY(gather_rows(ctx, Pr, n, dj, m, n, Z, n, nullptr));
                CTM_TRY(gather_rows(ctx, Pi, n, dj, m, n, Z + mn, n, nullptr));
                std::vector<double> hz(m);
                double dev = 1.0;
                for (int it = 0; it < 48; ++it) {
                    XM z{Z, Z + mn, n, false, false}, zh{Z, Z + mn, n, true, true};
                    CTM_TRY(xgemm(ctx, m, m, n, z, zh, G2, G2 + mm, m));
                    CTM_LAUNCH(ctx, sub_eye_kernel, dim3((m + 255) / 256), dim3(256), 0, G2, m);
                    CTM_TRY(row_norms_c128(ctx, G2, G2 + mm, m, m, m, zn));
                    CTM_HIP_CHECK(ctx, hipMemcpyAsync(hz.data(), zn, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
                    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                    const double prev = dev;
                    dev = *std::max_element(hz.begin(), hz.end());
                    if (!(dev == dev) || dev <= 1e-13 || (it > 0 && dev < 1e-10 && dev > 0.5 * prev)) break;
                    XM e{G2, G2 + mm, m, false, false};
                    CTM_TRY(xgemm(ctx, m, n, m, e, z, Z2, Z2 + mn, n));                       // (Z Z^H - I) Z
                    CTM_LAUNCH(ctx, axpy_kernel, dim3(1024), dim3(256), 0, Z, (const double*)Z2, -0.5, 2 * mn);
                }
                if (dev == dev && dev <= 1e-10) {
                    CTM_HIP_CHECK(ctx, hipMemcpyAsync(Wn, Z, sizeof(double) * 2 * mn, hipMemcpyDeviceToDevice, ctx->stream));
                    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                    done = true; ctx->svd_polar_completions += 1;
                }
            }
            if (!done) {
                ctx->svd_eig_completions += 1;
                const bool save = ctx->si_enable; ctx->si_enable = false;      // a projector's spectrum is flat: the leading-k iteration cannot converge on it
                const int st3 = jacobi_eigh_top_c(ctx, Pr, Pi, n, kb, Dn, Wn, nullptr);
                ctx->si_enable = save;
                CTM_TRY(st3);
            }
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vt, Vg, sizeof(double) * (size_t)kg * n, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vt + kn, Vg + (size_t)kg * n, sizeof(double) * (size_t)kg * n, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vt + (size_t)kg * n, Wn, sizeof(double) * (size_t)kb * n, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vt + kn + (size_t)kg * n, Wn + (size_t)kb * n, sizeof(double) * (size_t)kb * n, hipMemcpyDeviceToDevice, ctx->stream));
            CTM_TRY(reorth_rows_c(ctx, Vt, k, n, done ? 2 : 1));
        } else
            CTM_TRY(reorth_rows_c(ctx, Vt, k, n, 2));
    }
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
int svd_iter_c(ctm_ctx* ctx, const MatOp& op, int k, double* S, double* Ut, double* Vt, bool* converged, bool* want_krylov = nullptr) {
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
// status[0] = smallest pivot of the accepted factorisation, [1] = smallest, [2] = largest row norm.  Nothing is decided on the
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
    for (int attempt = 0; attempt < 2; ++attempt) {
        const double shift = attempt == 0 ? 0.0 : 1e-10;
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
    if (lane == 0) { status[0] = pmin; status[1] = mn; status[2] = mx; }
}

// rows of W (64 x n) -> orthonormal rows spanning the same space: two Cholesky-QR passes, and a third one -- decided on the device --
// when the first had to shift (nearly dependent rows); no host synchronisation.  Status words of the three passes go to
// `status` (9 doubles: pivot, min norm, max norm per pass; a skipped third pass reports 1), `flag3` is a device word.
int orthonormalise_block_async(ctm_ctx* ctx, double* W, int b, int n, double* G, double* Li, double* status, int* flag3) {
    // (Two passes for 32-row blocks were tried: no shifted first pass in any run on random tensors, three idle launches per block saved, < 0.5 % of
    // a sweep -- but on an SU(2)-symmetric state (RVB D = 3 tiled on the 2 x 2 cell, chi = 80: exactly dependent rows inside multiplets) a shifted
    // pass then sends the whole solve to the synchronous path.  The third, flag-skipped pass stays for every block size.)
    const int npass = 3;
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
int project_out(ctm_ctx* ctx, double* W, int b, int n, const double* B, int m, double* G, int reps = 2, int local = 0) {
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
    }
    auto resync = [&]() -> int {          // redo this solve on the synchronous path
        if (ctx->jacobi_verbose) fprintf(stderr, "[lz] n=%d asynchronous recurrence flagged: repeating on the synchronous path\n", n);
        ctx->lz_async_fallbacks += 1;
        ctx->lz_force_sync = true;
        const int st = svd_lanczos(ctx, op, k, S, Ut, Vt, converged);
        ctx->lz_force_sync = false;
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
        jnext = (int)hdr[HDR_STEPS] + (hdr[HDR_EST] <= tol / 30.0 ? -1 : (hdr[HDR_EST] > tol / 3.0 ? 1 : 0));
    else jnext = (int)std::ceil((b == 32 ? ctx->lz_first_factor32 : ctx->lz_first_factor) * k / b);
    jnext = std::max(jmin, std::min(jnext, jmax));
    double est_prev = 0.0; int steps_prev = 0;
    double mn, mx, s0 = 0.0;
    CTM_LAUNCH(ctx, hash_fill_kernel, dim3(1024), dim3(256), 0, Vall, b, n, (long long)n, 0x51f15eedULL);
    if (async) CTM_TRY(orthonormalise_block_async(ctx, Vall, b, n, G, Li, ostat + SW * (nstat - 1), flag3));
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
        if (async) CTM_TRY(orthonormalise_block_async(ctx, Uj, b, n, G, Li, ostat + SW * (2 * j), flag3));
        else {
            CTM_TRY(orthonormalise_block(ctx, Uj, b, n, norms, inv, &mn, &mx));
            s0 = std::max(s0, mx);
            if (mn <= 1e-13 * s0) { if (ctx->jacobi_verbose) fprintf(stderr, "[lz] n=%d breakdown at step %d (U)\n", n, j); return CTM_OK; }
        }
        // raw product and V_{j+1}
        CTM_TRY(matop_apply(ctx, op, false, Uj, n, b, Zj, n, want_mid ? URall + (size_t)j * b * n : nullptr)); applications += b;
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Vn, Zj, sizeof(double) * (size_t)b * n, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_TRY(project_out(ctx, Vn, b, n, Vall, (j + 1) * b, G, 2, ctx->lz_local_project ? b : 0));
        if (async) CTM_TRY(orthonormalise_block_async(ctx, Vn, b, n, G, Li, ostat + SW * (2 * j + 1), flag3));
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
                const double last = third ? q[6] : q[3];                       // pivot of the last pass: rows orthonormal to rounding iff ~1
                if (left) s0a = std::max(s0a, q[2]);
                // breakdown of the recurrence: a new block with (numerically) nothing in it -- the Krylov space has exhausted the range of a
                // rank-deficient operator (symmetric states at small chi: every solve).  Not a case for this solver on either path: leave
                // at once, as the synchronous recurrence does, instead of repeating the whole solve there to find the same thing
                if (q[1] == q[1] && !(q[1] > 1e-13 * std::max(s0a, 1e-300)) && slot != nstat - 1) { broke = true; return; }
                if (!(q[0] > 0.0) || !(last > 0.5) || !(q[1] > 0.0)) bad = true;   // (NaN fails too), zero row
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
        const bool save = ctx->si_enable; ctx->si_enable = false;
        ctx->force_abs = ctx->lz_abs_accuracy != 0;
        ctx->jacobi_quad_exit = ctx->lz_quad_exit;
        const int st = svd_full(ctx, T, m, kq, Ss, Xt, Yt);                      // rows of Xt / Yt: x_i^T, y_i^T
        ctx->jacobi_quad_exit = 0.0;
        ctx->force_abs = false;
        ctx->si_enable = save;
        CTM_TRY(st);
        ctx->lz_extractions += 1;
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
        jnext = (int)hdr[HDR_STEPS] + (hdr[HDR_EST] <= tol / 30.0 ? -1 : (hdr[HDR_EST] > tol / 3.0 ? 1 : 0));
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
        CTM_TRY(svd_full_c(ctx, T, T + mm, m, kq, Ss, Xt, Yt));                      // T = Xt^H diag(Ss) Yt
        ctx->lz_extractions += 1;
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

}  // namespace


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
    CTM_TRY(matop_apply(ctx, op, side0 == 0, B0, n, p, X, ld, M1));
    CTM_TRY(copy2d(ctx, B0, n, X + n, ld, p, n));
    if (want_mid) CTM_TRY(copy2d(ctx, M1, n, X + 2 * (size_t)n, ld, p, n));
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
        if (op.warm && op.warm_hdr && !op.M && !(op.Mi || op.ci[0]) && (ctx->warm_accept_tol > 0.0 || hdr_old[HDR_SIDE] >= 1.0)) {
            double dist = 0.0;
            if (ctx->warm_accept_tol > 0.0) CTM_TRY(spectrum_movement(ctx, op.warm_hdr, n, S, k, &dist));
            const double w[5] = {dist, 0.0, 0.0, hdr_old[HDR_SSKIP], hdr_old[HDR_SFAILS]};     // HDR_DIST, HDR_SIDE, HDR_RUN, HDR_SSKIP, HDR_SFAILS
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm_hdr + HDR_DIST, w, sizeof(w), hipMemcpyHostToDevice, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            ctx->warm_last_dist = dist;
        }
        return keep_warm();
    };
    if (op.Mi || op.ci[0]) {        // complex128
        if (Ut && Vt && ctx->si_enable && k < n && n >= ctx->si_min_n) {
            bool ok = false, krylov = false;
            MatOp op1 = op;
            if (op.warm_hdr && ctx->lz_enable && k >= ctx->lz_min_k) {      // direct Krylov entry of a full-rank unit, see the real branch below
                double hdr[HDR_WORDS];
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(hdr, op.warm_hdr, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
                CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                if (hdr[HDR_SKIP] >= 1.0 && hdr[HDR_STEPS] >= 1.0) {
                    CTM_TRY(fill_f64(ctx, op.warm_hdr, 1, hdr[HDR_SKIP] - 1.0)); ctx->si_warm_skips += 1;
                    ctx->lz_last_resid = 1.0;
                    CTM_TRY(svd_lanczos_c(ctx, op, k, S, Ut, Vt, &ok));
                    if (ok) return keep_warm();
                    op1.warm = nullptr; op1.warm_hdr = nullptr;
                }
            }
            CTM_TRY(svd_iter_c(ctx, op1, k, S, Ut, Vt, &ok, &krylov));
            if (ok) { ctx->si_hits += 1; return keep_warm(); }
            if (krylov) {
                ctx->lz_last_resid = 1.0;
                CTM_TRY(svd_lanczos_c(ctx, op, k, S, Ut, Vt, &ok));
                if (ok) return keep_warm();
                MatOp op2 = op;
                ArenaScope ws(ctx);
                if (ctx->lz_last_resid <= 1e-11) {
                    double* w2;
                    CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)k * n, (void**)&w2));
                    CTM_HIP_CHECK(ctx, hipMemcpyAsync(w2, Vt, sizeof(double) * 2 * (size_t)k * n, hipMemcpyDeviceToDevice, ctx->stream));
                    op2.warm = w2;
                }
                CTM_TRY(svd_iter_c(ctx, op2, k, S, Ut, Vt, &ok, nullptr));
                if (ok) { ctx->si_hits += 1; return keep_warm(); }
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
        return keep_warm();
    }
    if (Ut && Vt && ctx->si_enable && k < n && n >= ctx->si_min_n) {
        bool ok = false, krylov = false;
        MatOp op1 = op;
        if (op.warm_hdr && ctx->lz_enable && k >= ctx->lz_min_k) {
            // a unit whose last solve needed the Krylov solver and whose warm probe is not due yet goes there directly (no
            // 64-row rank probe either); if that should fail the regular path below starts cold
            CTM_HIP_CHECK(ctx, hipMemcpyAsync(hdr, op.warm_hdr, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
            CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->warm_accept_tol > 0.0 && op.warm && !op.M && hdr[HDR_STEPS] >= 1.0) {
                // stationary fast path: one Rayleigh-Ritz half step from the previous basis when the last solve found it close
                if (hdr[HDR_SSKIP] >= 1.0) { hdr[HDR_SSKIP] -= 1.0; CTM_TRY(fill_f64(ctx, op.warm_hdr + HDR_SSKIP, 1, hdr[HDR_SSKIP])); }
                else if (hdr[HDR_DIST] > 0.0 && hdr[HDR_DIST] <= ctx->warm_try_factor * ctx->warm_accept_tol &&
                         (ctx->warm_accept_max_run <= 0 || hdr[HDR_RUN] < ctx->warm_accept_max_run)) {
                    bool acc = false; double rr = 0.0;
                    CTM_TRY(svd_stationary(ctx, op, k, hdr[HDR_SIDE] >= 1.0 ? 1 : 0, S, Ut, Vt, &acc, &rr));
                    if (acc) {
                        double mv = 0.0;
                        CTM_TRY(spectrum_movement(ctx, op.warm_hdr, n, S, k, &mv));
                        const double w[5] = {std::max(mv, 1e-300), hdr[HDR_SIDE] >= 1.0 ? 0.0 : 1.0, hdr[HDR_RUN] + 1.0, 0.0, 0.0};
                        CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm_hdr + HDR_DIST, w, sizeof(w), hipMemcpyHostToDevice, ctx->stream));
                        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                        ctx->warm_accepts += 1; ctx->warm_last_dist = rr;
                        return CTM_OK;
                    }
                    // refused: the full solve below; the next attempts back off (x2 per consecutive refusal)
                    ctx->warm_rejects += 1;
                    hdr[HDR_SFAILS] = std::min(hdr[HDR_SFAILS] + 1.0, 6.0);
                    hdr[HDR_SSKIP] = std::ldexp(1.0, (int)hdr[HDR_SFAILS]) - 1.0;
                    if (op.have_mid) *op.have_mid = false;
                }
            }
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

