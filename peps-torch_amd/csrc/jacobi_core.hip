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
#include "jacobi_internal.h"

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
        if (tid == 0) atomicAdd(p.stat_rel + 2, 1ULL);         // diagnostics: inner sweeps of this launch ([jacobi] line of jacobi_verbose = 2)
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
        if (tid == 0) atomicAdd(p.stat_rel + 2, 1ULL);
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
int jacobi_rows(ctm_ctx* ctx, double* X, int R, long long ld, int Cg, int Ctot, int b, int ktop, double fro, int max_sweeps, bool cplx,
                bool tau_both, double null_rel) {
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
        CTM_HIP_CHECK(ctx, hipMemsetAsync(stat, 0, 3 * sizeof(double), ctx->stream));
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
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scratch, stat, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const double srel = ctx->h_scratch[0];
        ctx->last_sweeps = sweep + 1;
        ctx->last_offnorm = srel;
        if (ctx->jacobi_verbose > 1) {
            unsigned long long inner; memcpy(&inner, &ctx->h_scratch[2], sizeof(inner));
            fprintf(stderr, "[jacobi] R=%d sweep %d  scaled=%.3e classical=%.3e tau=%.3e  inner sweeps of the LDS eigensolver: %llu in %d launches\n", R, sweep + 1, srel,
                    ctx->h_scratch[1], std::sqrt(tau2), inner, rounds);
        }
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
// rot (optional, any k; needs Ut): np x np workspace, np = svd_full_rot_rows(ctx, n) (n padded to an even number of panels), for the ACCUMULATED
// ROTATIONS of the sweeps on the padded problem [M; 0] (all np rows, in the order the sweeps leave them).  rot_valid: it holds the rotations of
// an earlier decomposition of a nearby matrix (the Ritz matrix of the same unit in the previous CTM sweep, or of the same solve a few block
// steps earlier, extended by the identity): the sweeps then start from rot[:, :n] . M, whose rows are orthogonal up to the change of the
// matrix -- the quadratically convergent regime -- instead of from M.  Any orthogonal rot is a valid start (checked: unit row norms); on
// return it holds this decomposition's rotations.
int svd_full_rot_rows(ctm_ctx* ctx, int n) { return padded(n, choose_block(ctx, n)); }

int svd_full(ctm_ctx* ctx, const double* M, int n, int k, double* S, double* Ut, double* Vt, double* warm, double* rot, bool rot_valid) {
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
    if (rot && (!with_q || warm)) { ctx->set_error("svd_full: rot needs Ut and excludes warm"); return CTM_ERR_BADARG; }
    bool rot_start = false;
    if (rot && rot_valid) {
        const double fw = host_fro(ctx, rot, np, np, np, norms, h, &st);
        CTM_TRY(st);
        rot_start = std::fabs(fw - std::sqrt((double)np)) <= 1e-6 * std::sqrt((double)np);
        for (int i = 0; rot_start && i < np; ++i) rot_start = std::fabs(h[i] - 1.0) <= 1e-6;
        if (rot_start) ctx->ritz_warm_starts += 1;
    }
    if (rot_start) {
        CTM_TRY(fill_f64(ctx, X, (size_t)np * ld, 0.0));
        GemmDesc gw; gw.M = np; gw.N = n; gw.K = n; gw.A = rot; gw.sam = np; gw.sak = 1; gw.B = M; gw.sbk = n; gw.sbn = 1; gw.C = X; gw.ldc = ld;
        CTM_TRY(gemm_f64(ctx, gw));
        CTM_TRY(copy2d(ctx, rot, np, X + n, ld, np, np));
    }
    if (!warm_full && warm && with_q && k == n && ctx->eigh_warm) {
        const double fw = host_fro(ctx, warm, n, n, n, norms, h, &st);
        CTM_TRY(st);
        warm_full = std::fabs(fw - std::sqrt((double)n)) <= 1e-6 * std::sqrt((double)n);
        for (int i = 0; warm_full && i < n; ++i) warm_full = std::fabs(h[i] - 1.0) <= 1e-6;
    }
    if (rot_start) {
    } else if (warm_full) {
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
    if (rot) CTM_TRY(copy2d(ctx, X + n, ld, rot, np, np, np));    // every row of the accumulated rotations, unsorted: the next call's start
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

// rot / rot_valid: as svd_full() -- planar n x n (re plane, im plane) accumulated rotations, rows u_i^H in the order the sweeps leave them
int svd_full_c(ctm_ctx* ctx, const double* Mr, const double* Mi, int n, int k, double* S, double* Ut, double* Vt, double* warm, double* rot, bool rot_valid) {
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
    if (rot && (!with_q || np != n || warm)) { ctx->set_error("svd_full_c: rot needs Ut, excludes warm, and n must be an even number of 16-row panels"); return CTM_ERR_BADARG; }
    if (rot && rot_valid) {
        const double fw = host_fro(ctx, rot, 2 * n, n, n, norms, h, &st);
        CTM_TRY(st);
        bool okw = std::fabs(fw - std::sqrt((double)n)) <= 1e-6 * std::sqrt((double)n);
        for (int i = 0; okw && i < n; ++i) okw = std::fabs(std::sqrt(h[i] * h[i] + h[n + i] * h[n + i]) - 1.0) <= 1e-6;
        if (okw) { warm = rot; warm_full = true; ctx->ritz_warm_starts += 1; }
    }
    if (!warm_full && warm && with_q && k == n && ctx->eigh_warm) {
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
        if (warm != rot) ctx->eigh_warm_hits += 1;
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
    if (rot) {                       // every row of the accumulated rotations, unsorted, planar: the next call's start
        std::vector<int> all(n);
        std::iota(all.begin(), all.end(), 0);
        CTM_TRY(panel_gather(ctx, X + n, ld, all, n, n, rot, d_idx));
    }
    // rows of the accumulated unitary Q are u_k^H; Sigma V^H = Q M is recomputed by one k x n x n product (drift-free)
    CTM_TRY(panel_gather(ctx, X + n, ld, idx, k, n, Ut, d_idx));
    CTM_TRY(reorth_rows_c(ctx, Ut, k, n, 2));
    if (warm && warm != rot && k == n) CTM_HIP_CHECK(ctx, hipMemcpyAsync(warm, Ut, sizeof(double) * 2 * nn, hipMemcpyDeviceToDevice, ctx->stream));
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

