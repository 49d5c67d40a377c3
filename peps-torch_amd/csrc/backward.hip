// Adjoints of the two decompositions of the chi-truncation (first part): the regularised SVD backward of
// linalg/svd_gesdd.py:209-328 (SVDGESDD.backward) and the symmetric / Hermitian eigendecomposition backward of
// linalg/eig_sym.py:57-75 (SYMEIG.backward), restated on the device: every O(n^2 k) term is an FP64-MFMA GEMM, the k x k
// "F, G" weightings are one small elementwise kernel.  The adjoints of the contractions of a move are ctm_einsum calls issued by
// the host layer (peps-torch_amd/linalg/native_einsum.py).
//
//   dA = U K V^H + (1 - U U^H) gU S^-1 V^H + U S^-1 gV^H (1 - V V^H)
//   K  = 1/2 (F + G) o (U^H gU - gU^H U) + 1/2 (F - G) o (V^H gV - gV^H V) + diag(gS) + i diag(Im(U^H gU)_ii / s_i)
//   F_ij = x/(x^2 + e) with x = s_j - s_i,  G_ij = y/(y^2 + e) with y = s_j + s_i,  e = s_0 * eps,  zero diagonals;
//   1/s_i is taken as 0 where |s_i| < s_0 * eps (safe_inverse_2).
#include "contract.h"
#include <algorithm>

namespace {

__global__ void svd_bwd_mid_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai, const double* __restrict__ Br,
                                   const double* __restrict__ Bi, const double* __restrict__ S, const double* __restrict__ gS,
                                   double eps, int k, double* __restrict__ Kr, double* __restrict__ Ki) {
    const double e = S[0] * eps;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < k * k; q += gridDim.x * blockDim.x) {
        const int i = q / k, j = q - i * k;
        double kr = 0.0, ki = 0.0;
        if (i != j) {
            const double x = S[j] - S[i], y = S[j] + S[i];
            const double F = x / (x * x + e), G = y / (y * y + e);
            // (A - A^H)_ij = A_ij - conj(A_ji)
            if (Ar) { kr += 0.5 * (F + G) * (Ar[q] - Ar[j * k + i]); if (Ai) ki += 0.5 * (F + G) * (Ai[q] + Ai[j * k + i]); }
            if (Br) { kr += 0.5 * (F - G) * (Br[q] - Br[j * k + i]); if (Bi) ki += 0.5 * (F - G) * (Bi[q] + Bi[j * k + i]); }
        } else {
            if (gS) kr += gS[i];
            if (Ai) { const double si = (fabs(S[i]) < e) ? 0.0 : 1.0 / S[i]; ki += Ai[q] * si; }      // complex: i Im(U^H gU)_ii / s_i
        }
        Kr[q] = kr;
        if (Ki) Ki[q] = ki;
    }
}

__global__ void eig_bwd_mid_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai, const double* __restrict__ D,
                                   const double* __restrict__ gD, double reg, int k, double* __restrict__ Kr, double* __restrict__ Ki) {
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < k * k; q += gridDim.x * blockDim.x) {
        const int i = q / k, j = q - i * k;
        double kr = 0.0, ki = 0.0;
        if (i != j) {
            const double x = D[j] - D[i], F = x / (x * x + reg);
            if (Ar) kr = F * Ar[q];
            if (Ai) ki = F * Ai[q];
        } else if (gD) kr = gD[i];
        Kr[q] = kr;
        if (Ki) Ki[q] = ki;
    }
}

// x[:, j] *= (|s_j| < s_0 eps ? 0 : 1 / s_j)
__global__ void scale_cols_inv_kernel(double* x, long long rows, int cols, const double* __restrict__ S, double eps) {
    const double e = S[0] * eps;
    const long long tot = rows * cols;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(q % cols);
        x[q] *= (fabs(S[j]) < e) ? 0.0 : 1.0 / S[j];
    }
}

__global__ void axpy_kernel(double* y, const double* __restrict__ x, double a, size_t n) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) y[q] += a * x[q];
}

int nblk(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 2048); }

struct Marsh {       // boundary marshalling as in ctm_ops.hip: interleaved complex128 <-> planes
    ctm_ctx* ctx; std::vector<std::pair<double*, std::pair<double*, size_t>>> outs;
    explicit Marsh(ctm_ctx* c) : ctx(c) {}
    int in(const double* ptr, long long r, long long c, DT* t) {
        *t = DT(ptr, {r, c});
        if (!ptr || !ctx->cplx) return CTM_OK;
        const size_t n = (size_t)(r * c);
        double* buf;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * n, (void**)&buf));
        CTM_TRY(deinterleave_c128(ctx, ptr, buf, buf + n, n));
        t->p = buf; t->q = buf + n;
        return CTM_OK;
    }
    int out(double* user, long long r, long long c, DT* t) {
        *t = DT(user, {r, c});
        if (!ctx->cplx) return CTM_OK;
        const size_t n = (size_t)(r * c);
        double* buf;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * n, (void**)&buf));
        t->p = buf; t->q = buf + n;
        outs.push_back({user, {buf, n}});
        return CTM_OK;
    }
    int finish() {
        for (auto& o : outs) CTM_TRY(interleave_c128(ctx, o.second.first, o.second.first + o.second.second, o.first, o.second.second));
        return CTM_OK;
    }
};

XM xm(const DT& t, bool trans, bool conj) { XM x; x.re = t.p; x.im = t.q; x.ld = t.dims[1]; x.t = trans; x.c = conj; return x; }

int alloc2(ctm_ctx* ctx, long long r, long long c, DT* t) {
    *t = DT(nullptr, {r, c});
    const size_t n = (size_t)(r * c);
    CTM_TRY(arena_alloc(ctx, sizeof(double) * n * (ctx->cplx ? 2 : 1), (void**)&t->p));
    if (ctx->cplx) t->q = t->p + n;
    return CTM_OK;
}

// C = op(A) op(B)
int mm(ctm_ctx* ctx, const DT& A, bool ta, bool ca, const DT& B, bool tb, bool cb, const DT& C) {
    const int M = (int)(ta ? A.dims[1] : A.dims[0]), K = (int)(ta ? A.dims[0] : A.dims[1]), N = (int)(tb ? B.dims[0] : B.dims[1]);
    return xgemm(ctx, M, N, K, xm(A, ta, ca), xm(B, tb, cb), C.p, C.q, C.dims[1]);
}

int add_to(ctm_ctx* ctx, const DT& y, const DT& x, double a) {
    const size_t n = (size_t)(y.dims[0] * y.dims[1]);
    CTM_LAUNCH(ctx, axpy_kernel, dim3(nblk(n)), dim3(256), 0, y.p, (const double*)x.p, a, n);
    if (y.q) CTM_LAUNCH(ctx, axpy_kernel, dim3(nblk(n)), dim3(256), 0, y.q, (const double*)x.q, a, n);
    return CTM_OK;
}

}  // namespace

static int svd_backward_impl(ctm_ctx* ctx, const double* U, const double* S, const double* V, const double* gU, const double* gS,
                             const double* gV, int m, int n, int k, double eps, double* dA) {
    if (m < 1 || n < 1 || k < 1 || k > std::min(m, n) || !U || !S || !V || !dA) { ctx->set_error("svd_backward: bad arguments"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    Marsh io(ctx);
    DT tU, tV, tgU, tgV, tdA;
    CTM_TRY(io.in(U, m, k, &tU));
    CTM_TRY(io.in(V, n, k, &tV));
    CTM_TRY(io.in(gU, m, k, &tgU));
    CTM_TRY(io.in(gV, n, k, &tgV));
    CTM_TRY(io.out(dA, m, n, &tdA));
    DT A, B, K, t1;
    CTM_TRY(alloc2(ctx, k, k, &K));
    CTM_TRY(alloc2(ctx, m, k, &t1));
    if (gU) { CTM_TRY(alloc2(ctx, k, k, &A)); CTM_TRY(mm(ctx, tU, true, true, tgU, false, false, A)); }        // A = U^H gU
    if (gV) { CTM_TRY(alloc2(ctx, k, k, &B)); CTM_TRY(mm(ctx, tV, true, true, tgV, false, false, B)); }        // B = V^H gV
    CTM_LAUNCH(ctx, svd_bwd_mid_kernel, dim3(nblk((size_t)k * k)), dim3(256), 0, (const double*)(gU ? A.p : nullptr),
               (const double*)(gU ? A.q : nullptr), (const double*)(gV ? B.p : nullptr), (const double*)(gV ? B.q : nullptr), S, gS, eps, k, K.p, K.q);
    CTM_TRY(mm(ctx, tU, false, false, K, false, false, t1));                    // U K
    CTM_TRY(mm(ctx, t1, false, false, tV, true, true, tdA));                    // (U K) V^H
    if (gU && m > k) {      // (1 - U U^H) (gU S^-1) V^H
        DT W, C1, T2, T3;
        CTM_TRY(alloc2(ctx, m, k, &W)); CTM_TRY(alloc2(ctx, k, k, &C1)); CTM_TRY(alloc2(ctx, m, k, &T2)); CTM_TRY(alloc2(ctx, m, n, &T3));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(W.p, tgU.p, sizeof(double) * (size_t)m * k * (ctx->cplx ? 2 : 1), hipMemcpyDeviceToDevice, ctx->stream));
        CTM_LAUNCH(ctx, scale_cols_inv_kernel, dim3(nblk((size_t)m * k)), dim3(256), 0, W.p, (long long)m, k, S, eps);
        if (W.q) CTM_LAUNCH(ctx, scale_cols_inv_kernel, dim3(nblk((size_t)m * k)), dim3(256), 0, W.q, (long long)m, k, S, eps);
        CTM_TRY(mm(ctx, tU, true, true, W, false, false, C1));
        CTM_TRY(mm(ctx, tU, false, false, C1, false, false, T2));
        CTM_TRY(add_to(ctx, W, T2, -1.0));
        CTM_TRY(mm(ctx, W, false, false, tV, true, true, T3));
        CTM_TRY(add_to(ctx, tdA, T3, 1.0));
    }
    if (gV && n > k) {      // U S^-1 gV^H (1 - V V^H) = U (Z - V V^H Z)^H with Z = gV S^-1
        DT Z, C1, T2, T3;
        CTM_TRY(alloc2(ctx, n, k, &Z)); CTM_TRY(alloc2(ctx, k, k, &C1)); CTM_TRY(alloc2(ctx, n, k, &T2)); CTM_TRY(alloc2(ctx, m, n, &T3));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(Z.p, tgV.p, sizeof(double) * (size_t)n * k * (ctx->cplx ? 2 : 1), hipMemcpyDeviceToDevice, ctx->stream));
        CTM_LAUNCH(ctx, scale_cols_inv_kernel, dim3(nblk((size_t)n * k)), dim3(256), 0, Z.p, (long long)n, k, S, eps);
        if (Z.q) CTM_LAUNCH(ctx, scale_cols_inv_kernel, dim3(nblk((size_t)n * k)), dim3(256), 0, Z.q, (long long)n, k, S, eps);
        CTM_TRY(mm(ctx, tV, true, true, Z, false, false, C1));
        CTM_TRY(mm(ctx, tV, false, false, C1, false, false, T2));
        CTM_TRY(add_to(ctx, Z, T2, -1.0));
        CTM_TRY(mm(ctx, tU, false, false, Z, true, true, T3));
        CTM_TRY(add_to(ctx, tdA, T3, 1.0));
    }
    CTM_TRY(io.finish());
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
}

static int eigh_backward_impl(ctm_ctx* ctx, const double* D, const double* U, const double* gD, const double* gU, int n, int k, double reg,
                              double* dA) {
    if (n < 1 || k < 1 || k > n || !D || !U || !dA) { ctx->set_error("eigh_backward: bad arguments"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    Marsh io(ctx);
    DT tU, tgU, tdA, A, K, t1;
    CTM_TRY(io.in(U, n, k, &tU));
    CTM_TRY(io.in(gU, n, k, &tgU));
    CTM_TRY(io.out(dA, n, n, &tdA));
    CTM_TRY(alloc2(ctx, k, k, &K));
    CTM_TRY(alloc2(ctx, n, k, &t1));
    if (gU) { CTM_TRY(alloc2(ctx, k, k, &A)); CTM_TRY(mm(ctx, tU, true, true, tgU, false, false, A)); }        // U^H gU
    CTM_LAUNCH(ctx, eig_bwd_mid_kernel, dim3(nblk((size_t)k * k)), dim3(256), 0, (const double*)(gU ? A.p : nullptr),
               (const double*)(gU ? A.q : nullptr), D, gD, reg, k, K.p, K.q);
    CTM_TRY(mm(ctx, tU, false, false, K, false, false, t1));
    CTM_TRY(mm(ctx, t1, false, false, tU, true, true, tdA));
    CTM_TRY(io.finish());
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
}

// The C-ABI entries proper (ctm_entry: one call at a time per context, no C++ exception crosses the boundary).
extern "C" {

int ctm_svd_backward(ctm_ctx* ctx, const double* U, const double* S, const double* V, const double* gU, const double* gS,
                     const double* gV, int m, int n, int k, double eps, double* dA) {
    return ctm_entry(ctx, "ctm_svd_backward", [&]() -> int { return svd_backward_impl(ctx, U, S, V, gU, gS, gV, m, n, k, eps, dA); });
}

int ctm_eigh_backward(ctm_ctx* ctx, const double* D, const double* U, const double* gD, const double* gU, int n, int k, double reg,
                      double* dA) {
    return ctm_entry(ctx, "ctm_eigh_backward", [&]() -> int { return eigh_backward_impl(ctx, D, U, gD, gU, n, k, reg, dA); });
}

}  // extern "C"
