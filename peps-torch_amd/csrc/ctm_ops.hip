// The CTM hot path above the kernels: enlarged corners, halves, projectors, absorb, C4v move, RDMs.
// Each entry point restates one "raw tensor tuple in / raw tensor tuple out" closure of the
// reference (cited in include/ctm_hip.h) as a table-driven sequence of device contractions.
#include "contract.h"
#include <algorithm>
#include <cmath>

namespace {

// ---------------------------------------------------------------------------------------------
// small device helpers for the truncation logic
// ---------------------------------------------------------------------------------------------
// fix_svd_signs (svd_gesdd.py:18-26) on row-stored factors: Ut, Vt are k x n; one workgroup per row:
// argmax of the int64-quantised |U| (first occurrence), then both rows are multiplied by its sign.
// ref (optional, k x n): the rows this decomposition's predecessor returned for the same unit (left rows if ref_left, else right rows).
// A row that still points along its predecessor (|cos| > 1/2) takes the predecessor's orientation instead of the argmax rule: the rule
// flips a vector whenever two of its components of almost equal magnitude swap rank, i.e. it re-gauges legs of a CONVERGED environment
// by signs from sweep to sweep (measured: singular values stationary to 1e-15, previous vectors with residual 3e-2 s_0) -- harmless for
// every gauge-invariant quantity, fatal for restarting from the previous basis (stationary fast path, ctm_args.projector_warm_tol).
__global__ void fix_signs_rows_kernel(double* Ut, double* Vt, int k, int n, double* X1, double* X2, const double* ref = nullptr, int ref_left = 0) {
    const int r = blockIdx.x;
    if (r >= k) return;
    double* u = Ut + (size_t)r * n;
    double* v = Vt + (size_t)r * n;
    __shared__ long long sb[256]; __shared__ int si[256];
    __shared__ double sd[3][256];
    __shared__ double s_sg;
    if (threadIdx.x == 0) s_sg = 0.0;
    if (ref) {
        const double* x = ref_left ? u : v; const double* y = ref + (size_t)r * n;
        double d = 0.0, xx = 0.0, yy = 0.0;
        for (int c = threadIdx.x; c < n; c += blockDim.x) { d += x[c] * y[c]; xx += x[c] * x[c]; yy += y[c] * y[c]; }
        sd[0][threadIdx.x] = d; sd[1][threadIdx.x] = xx; sd[2][threadIdx.x] = yy;
        __syncthreads();
        for (int s = blockDim.x / 2; s > 0; s >>= 1) {
            if (threadIdx.x < s) for (int q = 0; q < 3; ++q) sd[q][threadIdx.x] += sd[q][threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0 && sd[0][0] * sd[0][0] > 0.25 * sd[1][0] * sd[2][0]) s_sg = sd[0][0] < 0.0 ? -1.0 : 1.0;
    }
    __syncthreads();
    long long best = -1; int bi = 0;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const long long a = (long long)(fabs(u[c]) * 1099511627776.0);   // 2^40
        if (a > best) { best = a; bi = c; }                                 // strided scan keeps the first max per thread
    }
    sb[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const long long ob = sb[threadIdx.x + s]; const int oi = si[threadIdx.x + s];
            if (ob > sb[threadIdx.x] || (ob == sb[threadIdx.x] && oi < si[threadIdx.x])) { sb[threadIdx.x] = ob; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    const double ph = u[si[0]];
    const double sg = (s_sg != 0.0) ? s_sg : ((ph < 0.0) ? -1.0 : 1.0);
    __syncthreads();
    if (sg < 0.0) {
        for (int c = threadIdx.x; c < n; c += blockDim.x) { u[c] = -u[c]; v[c] = -v[c]; }
        // rows that are linear images of u (X1) and of v (X2) follow the sign of their triplet
        if (X1) { double* x = X1 + (size_t)r * n; for (int c = threadIdx.x; c < n; c += blockDim.x) x[c] = -x[c]; }
        if (X2) { double* x = X2 + (size_t)r * n; for (int c = threadIdx.x; c < n; c += blockDim.x) x[c] = -x[c]; }
    }
}

// out (n x chi, row-major) = transpose of rows[0:chi] (k x n) with columns > keep zeroed
__global__ void rows_to_cols_kernel(const double* rows, int n, int chi, int keep_last, double* out, double sign) {
    const size_t tot = (size_t)n * chi;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const size_t i = q / chi, j = q - i * chi;
        out[q] = ((int)j <= keep_last) ? sign * rows[j * n + i] : 0.0;
    }
}

// complex fix_svd_signs on planar row factors (rows = u^H, v^H): the phase of the max-|U| entry is divided out, i.e.
// both rows are multiplied by conj(ut)/|ut| with ut the (conjugated) pivot entry of the row.
__global__ void fix_phase_rows_c_kernel(double* Ur, double* Ui, double* Vr, double* Vi, int k, int n, const double* refr = nullptr,
                                        const double* refi = nullptr, int ref_left = 0) {
    const int r = blockIdx.x;
    if (r >= k) return;
    double* ur = Ur + (size_t)r * n; double* ui = Ui + (size_t)r * n;
    double* vr = Vr + (size_t)r * n; double* vi = Vi + (size_t)r * n;
    __shared__ long long sb[256]; __shared__ int si[256];
    __shared__ double sd[4][256];
    __shared__ double s_ph[2];
    if (threadIdx.x == 0) { s_ph[0] = 0.0; s_ph[1] = 0.0; }
    if (refr) {
        // the row its predecessor returned for this unit: a row that still points along it (|cos| > 1/2) keeps ITS phase (see
        // fix_signs_rows_kernel): <x e^{i phi}, ref> real positive
        const double* xr = ref_left ? ur : vr; const double* xi = ref_left ? ui : vi;
        const double* yr = refr + (size_t)r * n; const double* yi = refi + (size_t)r * n;
        double zr = 0.0, zi = 0.0, xx = 0.0, yy = 0.0;
        for (int c = threadIdx.x; c < n; c += blockDim.x) {
            zr += xr[c] * yr[c] + xi[c] * yi[c]; zi += xi[c] * yr[c] - xr[c] * yi[c];
            xx += xr[c] * xr[c] + xi[c] * xi[c]; yy += yr[c] * yr[c] + yi[c] * yi[c];
        }
        sd[0][threadIdx.x] = zr; sd[1][threadIdx.x] = zi; sd[2][threadIdx.x] = xx; sd[3][threadIdx.x] = yy;
        __syncthreads();
        for (int s = blockDim.x / 2; s > 0; s >>= 1) {
            if (threadIdx.x < s) for (int q = 0; q < 4; ++q) sd[q][threadIdx.x] += sd[q][threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const double z2 = sd[0][0] * sd[0][0] + sd[1][0] * sd[1][0];
            if (z2 > 0.25 * sd[2][0] * sd[3][0]) { const double m = sqrt(z2); s_ph[0] = sd[0][0] / m; s_ph[1] = -sd[1][0] / m; }
        }
    }
    __syncthreads();
    long long best = -1; int bi = 0;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const long long a = (long long)(sqrt(ur[c] * ur[c] + ui[c] * ui[c]) * 1099511627776.0);
        if (a > best) { best = a; bi = c; }
    }
    sb[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const long long ob = sb[threadIdx.x + s]; const int oi = si[threadIdx.x + s];
            if (ob > sb[threadIdx.x] || (ob == sb[threadIdx.x] && oi < si[threadIdx.x])) { sb[threadIdx.x] = ob; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    const double a = ur[si[0]], b = ui[si[0]];
    const double m = sqrt(a * a + b * b);
    const bool follow = s_ph[0] != 0.0 || s_ph[1] != 0.0;
    __syncthreads();
    if (follow || m > 0.0) {
        const double pr = follow ? s_ph[0] : a / m, pi = follow ? s_ph[1] : -b / m;
        for (int c = threadIdx.x; c < n; c += blockDim.x) {
            double x = ur[c], y = ui[c]; ur[c] = x * pr - y * pi; ui[c] = x * pi + y * pr;
            x = vr[c]; y = vi[c]; vr[c] = x * pr - y * pi; vi[c] = x * pi + y * pr;
        }
    }
}

// multiplet logic (custom_svd.py:70-86): index of the last kept value
int multiplet_chi(const std::vector<double>& S, int chi, double eps_multiplet, double abs_tol) {
    std::vector<double> g(chi + 1), gaps(chi);
    for (int i = 0; i <= chi; ++i) { g[i] = std::fabs(S[i]); if (g[i] < abs_tol) g[i] = 0.0; }
    for (int i = 0; i < chi; ++i) {
        gaps[i] = (g[i] - std::fabs(S[i + 1])) / (g[i] + 1.0e-16);
        if (gaps[i] > 1.0) gaps[i] = 0.0;
    }
    int chi_new = chi;
    if (gaps[chi - 1] < eps_multiplet) {
        for (int i = chi - 1; i >= 0; --i) if (gaps[i] > eps_multiplet) { chi_new = i; break; }
    }
    return chi_new;
}

const ctm_trunc_cfg kDefaultCfg = {1.0e-8, 1.0e-8, 1.0e-14, 1, 1};

// Boundary marshalling.  Real contexts: a DT aliases the caller's buffer.  Complex128 contexts: the caller's
// interleaved (re,im) tensor is split into two planes in the arena on the way in, and planar results are interleaved
// into the caller's buffer by finish().
struct IO {
    ctm_ctx* ctx; bool cx;
    struct Out { double* user; double* re; size_t n; };
    std::vector<Out> outs;
    explicit IO(ctm_ctx* c) : ctx(c), cx(c->cplx) {}
    int in(const double* ptr, const std::vector<long long>& dims, DT* t) {
        *t = DT(ptr, dims);
        if (!cx) return CTM_OK;
        const size_t n = (size_t)t->numel();
        double* buf;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * n, (void**)&buf));
        CTM_TRY(deinterleave_c128(ctx, ptr, buf, buf + n, n));
        t->p = buf; t->q = buf + n;
        return CTM_OK;
    }
    int out(double* user, size_t n, DT* t) {
        t->p = user; t->q = nullptr; t->cj = false;
        if (!cx) return CTM_OK;
        double* buf;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * n, (void**)&buf));
        t->p = buf; t->q = buf + n;
        outs.push_back({user, buf, n});
        return CTM_OK;
    }
    int finish() {
        for (auto& o : outs) CTM_TRY(interleave_c128(ctx, o.re, o.re + o.n, o.user, o.n));
        outs.clear();
        return CTM_OK;
    }
};

// arena tensor (two planes in complex contexts)
int alloc_dt(ctm_ctx* ctx, const std::vector<long long>& dims, DT* t) {
    *t = DT(nullptr, dims);
    const size_t n = (size_t)t->numel();
    CTM_TRY(arena_alloc(ctx, sizeof(double) * n * (ctx->cplx ? 2 : 1), (void**)&t->p));
    if (ctx->cplx) t->q = t->p + n;
    return CTM_OK;
}

XM xm(const DT& t, long long ld, bool trans, bool conj = false) { XM x; x.re = t.p; x.im = t.q; x.ld = ld; x.t = trans; x.c = conj; return x; }

// x /= max|x| (kind 1, 'inf') or x /= |x|_2 (kind 2)     (move_normalize_c, ctmrg.py:210-230)
int normalize_dt(ctm_ctx* ctx, const DT& t, int kind = 1) {
    const size_t n = (size_t)t.numel();
    double* s = ctx->d_scratch + 8;
    if (kind == 2) {
        ArenaScope scope(ctx);
        const size_t tot = n * (t.q ? 2 : 1);           // planes are adjacent: |z|_2^2 = |re|^2 + |im|^2
        if (t.q && t.q != t.p + n) { ctx->set_error("normalize: planes not adjacent"); return CTM_ERR_BADARG; }
        double* tmp;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (tot / 4096 + 2), (void**)&tmp));
        CTM_TRY(norm2_f64(ctx, t.p, tot, tmp, s));
        return div_by_device_scalar(ctx, t.p, tot, s, 0);
    }
    if (!t.q) { CTM_TRY(absmax_f64(ctx, t.p, n, s)); return div_by_device_scalar(ctx, t.p, n, s, 0); }
    CTM_TRY(absmax_c128(ctx, t.p, t.q, n, s));
    CTM_TRY(div_by_device_scalar(ctx, t.p, n, s, 0));
    return div_by_device_scalar(ctx, t.q, n, s, 0);
}


struct TruncOut { std::vector<double> S; int keep_last; int k; };

// leading triplets of M (n x n) as ROW factors Ut, Vt (k x n), S on host, multiplet-aware keep index
int svd_rows_op(ctm_ctx* ctx, const MatOp& op, int chi, const ctm_trunc_cfg& cfg, double* Ut, double* Vt, double* dS, TruncOut* to) {
    const int n = op.n;
    const int k = (chi < n) ? chi + 1 : n;
    // stationary fast path enabled: the orientation of the returned vectors follows the previous decomposition of this unit (see
    // fix_signs_rows_kernel), whose sign-fixed rows the workspace holds -- kept aside, the solver overwrites them
    ArenaScope ref_scope(ctx);
    double* ref = nullptr; int ref_left = 0;
    const bool follow = (ctx->warm_accept_tol > 0.0 || ctx->sign_follow) && op.warm && op.warm_hdr && !op.M && cfg.fix_signs && k < n;
    const size_t cz = ctx->cplx ? 2 : 1;
    if (follow) {
        double side = 0.0;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * cz * (size_t)k * n, (void**)&ref));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(ref, op.warm, sizeof(double) * cz * (size_t)k * n, hipMemcpyDeviceToDevice, ctx->stream));
        CTM_HIP_CHECK(ctx, hipMemcpyAsync(&side, op.warm_hdr + 6 /* HDR_SIDE, svd_leading.hip */, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        ref_left = side >= 1.0 ? 1 : 0;
    }
    CTM_TRY(jacobi_svd_top_op(ctx, op, k, dS, Ut, Vt));
    to->S.resize(k); to->k = k;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(to->S.data(), dS, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (cfg.fix_signs) {
        const int kf = std::min(k, chi);
        const size_t kn = (size_t)k * n;
        if (ctx->cplx) {
            CTM_LAUNCH(ctx, fix_phase_rows_c_kernel, dim3(kf), dim3(256), 0, Ut, Ut + kn, Vt, Vt + kn, kf, n, (const double*)ref, (const double*)(ref ? ref + kn : nullptr), ref_left);
            if (follow) {      // the workspace keeps the rows AS RETURNED (planar: real plane, imaginary plane)
                double side = 0.0;
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(&side, op.warm_hdr + 6, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                const double* src = side >= 1.0 ? Ut : Vt;
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm, src, sizeof(double) * (size_t)kf * n, hipMemcpyDeviceToDevice, ctx->stream));
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm + kn, src + kn, sizeof(double) * (size_t)kf * n, hipMemcpyDeviceToDevice, ctx->stream));
            }
        } else {
            const bool mids = op.have_mid && *op.have_mid;
            CTM_LAUNCH(ctx, fix_signs_rows_kernel, dim3(kf), dim3(256), 0, Ut, Vt, kf, n, mids ? op.out_uR : (double*)nullptr, mids ? op.out_vRt : (double*)nullptr,
                       (const double*)ref, ref_left);
            if (follow) {      // the workspace keeps the rows AS RETURNED (their orientation is the next call's reference)
                double side = 0.0;
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(&side, op.warm_hdr + 6, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                CTM_HIP_CHECK(ctx, hipMemcpyAsync(op.warm, side >= 1.0 ? Ut : Vt, sizeof(double) * (size_t)kf * n, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
    }
    const int kc = std::min(chi, n);
    to->keep_last = kc - 1;
    if (cfg.keep_multiplets && chi < n) to->keep_last = std::min(kc - 1, multiplet_chi(to->S, chi, cfg.eps_multiplet, cfg.multiplet_abstol));
    return CTM_OK;
}

int svd_rows(ctm_ctx* ctx, const DT& M, int n, int chi, const ctm_trunc_cfg& cfg, double* Ut, double* Vt, double* dS, TruncOut* to,
             double* basis = nullptr) {
    MatOp op; op.n = n; op.M = M.p; op.Mi = M.q;
    if (basis) { const int k = (chi < n) ? chi + 1 : n; op.warm = basis; op.warm_hdr = basis + (size_t)(ctx->cplx ? 2 : 1) * k * n; }
    return svd_rows_op(ctx, op, chi, cfg, Ut, Vt, dS, to);
}

// U (n x kc) = (rows 0..kc of the k x n row factor)^H with the columns beyond keep_last zeroed; planar in complex contexts
int rows_to_cols(ctm_ctx* ctx, const double* rows, int k, int n, int kc, int keep_last, const DT& out) {
    CTM_LAUNCH(ctx, rows_to_cols_kernel, dim3(1024), dim3(256), 0, rows, n, kc, keep_last, out.p, 1.0);
    if (out.q) CTM_LAUNCH(ctx, rows_to_cols_kernel, dim3(1024), dim3(256), 0, rows + (size_t)k * n, n, kc, keep_last, out.q, -1.0);
    return CTM_OK;
}

// S_sqrt = rsqrt(S) where S/S[0] > reltol (ctm_projectors.py:266-270), zero beyond the kept multiplets
void proj_scale(const TruncOut& to, int kc, double reltol, std::vector<double>* Sh, std::vector<double>* sc) {
    Sh->assign(kc, 0.0); sc->assign(kc, 0.0);
    for (int i = 0; i < kc; ++i) (*Sh)[i] = (i <= to.keep_last) ? to.S[i] : 0.0;
    int nz = 0;
    for (int i = 0; i < kc; ++i) if ((*Sh)[0] > 0.0 && (*Sh)[i] / (*Sh)[0] > reltol) { (*sc)[nz] = 1.0 / std::sqrt((*Sh)[i]); ++nz; }
}

// out (n x kc) = opA(cA) * ( opB(cB) * rows^T[^H] ) * diag(scale)   with rows = k x n row factors (kc leading ones used)
// Only the first `ncol` columns carry a non-zero scale (S/S[0] > reltol is a prefix of the descending spectrum): the others
// are exact zeros of the result (as in the reference, ctm_projectors.py:266-283) and are not computed.
// out (n x ldo, first ncol columns) = in^T (in: ncol x n) with column j scaled by scale[j]
__global__ void transpose_scale_kernel(const double* __restrict__ in, int ncol, int n, double* __restrict__ out, long long ldo,
                                       const double* __restrict__ scale) {
    const long long tot = (long long)n * ncol;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (long long)gridDim.x * blockDim.x) {
        const long long i = q / ncol; const int j = (int)(q - i * ncol);
        out[i * ldo + j] = in[(long long)j * n + i] * (scale ? scale[j] : 1.0);
    }
}

int corner_chain_times_rowsT(ctm_ctx* ctx, int n, int mid, int k, int kc, int ncol, const DT& cA, bool tA, const DT& cB, bool tB, const double* rows,
                             bool conj_rows, const double* d_scale, const DT& out) {
    // opA(cA) is n x mid, opB(cB) is mid x n (stored transposed when the flag is set)
    ArenaScope scope(ctx);
    CTM_TRY(fill_f64(ctx, out.p, (size_t)n * kc * (out.q ? 2 : 1), 0.0));      // planes are adjacent
    if (out.q && out.q != out.p + (size_t)n * kc) CTM_TRY(fill_f64(ctx, out.q, (size_t)n * kc, 0.0));
    if (ncol <= 0) return CTM_OK;
    if (!ctx->cplx && ncol <= 64 && ctx->chain_as_strips) {
        // few significant columns: keep them as ROWS so that both corner passes are (ncol x n)(n x n) strips -- the streaming
        // kernel reads each corner once at HBM speed -- and transpose the small result at the end
        //   t1^T (ncol x mid) = rows opB(cB)^T ;  out^T (ncol x n) = t1^T opA(cA)^T
        double *t1t, *ot;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)ncol * mid, (void**)&t1t));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)ncol * n, (void**)&ot));
        auto pass = [&](const double* X, int kin, int nout, const DT& Z, bool transZ, double* Y) {
            // Y (ncol x nout) = X (ncol x kin) op(Z); Z stored kin x nout (transZ == false) or nout x kin
            GemmDesc g; g.M = ncol; g.N = nout; g.K = kin; g.A = X; g.sam = kin; g.sak = 1; g.B = Z.p;
            if (transZ) { g.sbk = 1; g.sbn = kin; } else { g.sbk = nout; g.sbn = 1; }
            g.C = Y; g.ldc = nout;
            return gemm_f64(ctx, g);
        };
        CTM_TRY(pass(rows, n, mid, cB, !tB, t1t));
        CTM_TRY(pass(t1t, mid, n, cA, !tA, ot));
        const long long tot = (long long)n * ncol;
        CTM_LAUNCH(ctx, transpose_scale_kernel, dim3((int)std::min<long long>((tot + 255) / 256, 2048)), dim3(256), 0,
                           (const double*)ot, ncol, n, out.p, (long long)kc, d_scale);
        return CTM_OK;
    }
    DT t1;
    CTM_TRY(alloc_dt(ctx, {mid, ncol}, &t1));
    XM r; r.re = rows; r.im = ctx->cplx ? rows + (size_t)k * n : nullptr; r.ld = n; r.t = true; r.c = conj_rows;
    CTM_TRY(xgemm(ctx, mid, ncol, n, xm(cB, tB ? mid : n, tB), r, t1.p, t1.q, ncol));
    return xgemm(ctx, n, ncol, mid, xm(cA, tA ? n : mid, tA), xm(t1, ncol, false), out.p, out.q, kc, d_scale);
}

// ---------------------------------------------------------------------------------------------
// enlarged corners: table (same specs as oracle/ctm_oracle.py _CORNER)
// ---------------------------------------------------------------------------------------------
struct CornerSpec { int t1_axis, t1_leg, t2_axis, t2_leg; const char* closed; const char* open; };
const CornerSpec kCorner[4] = {
    /* LU */ {1, 1, 2, 2, "ab,bUVx,ayLM,sULdr,sVMDR->ydDxrR", "ab,bUVx,ayLM,sULdr,tVMDR->ydDxrRst"},
    /* RU */ {1, 4, 1, 1, "ab,brRx,yuUa,suldr,sULDR->ylLxdD", "ab,brRx,yuUa,suldr,tULDR->ylLxdDst"},
    /* RD */ {0, 3, 1, 4, "ab,dDxb,yrRa,suldr,sULDR->yuUxlL", "ab,dDxb,yrRa,suldr,tULDR->yuUxlLst"},
    /* LD */ {2, 2, 0, 3, "ab,xalL,dDby,suldr,sULDR->xuUyrR", "ab,xalL,dDby,suldr,tULDR->xuUyrRst"},
};

// T tensor of rel. direction with D^2 axis `axis` split into (D,D): dims of the 4-index view
std::vector<long long> t_view(int axis, long long chi, long long D) {
    std::vector<long long> d;
    for (int a = 0; a < 3; ++a) { if (a == axis) { d.push_back(D); d.push_back(D); } else d.push_back(chi); }
    return d;
}

int corner_impl(ctm_ctx* ctx, int corner, int open, const DT& C, const DT& T1, const DT& T2, const DT& a, int chi, const int* ad,
                DT* res) {
    if (corner < 0 || corner > 3) { ctx->set_error("c2x2: bad corner"); return CTM_ERR_BADARG; }
    const CornerSpec& sp = kCorner[corner];
    DT tC = C.view({chi, chi});
    DT tT1 = T1.view(t_view(sp.t1_axis, chi, ad[sp.t1_leg]));
    DT tT2 = T2.view(t_view(sp.t2_axis, chi, ad[sp.t2_leg]));
    DT tA = a.view({ad[0], ad[1], ad[2], ad[3], ad[4]});
    if (open == 2) {      // open corner with the physical legs LEADING: [s][t][n0][n1] (contiguous n0 x n1 slices per (s,t))
        std::string e(sp.open);
        const size_t ar = e.find("->");
        const std::string o = e.substr(ar + 2);
        e = e.substr(0, ar + 2) + o.substr(o.size() - 2) + o.substr(0, o.size() - 2);
        return dev_network(ctx, e, {tC, tT1, tT2, tA, tA.conj()}, res);
    }
    return dev_network(ctx, open ? sp.open : sp.closed, {tC, tT1, tT2, tA, tA.conj()}, res);
}

// the four input tensors (C, T1, T2, a) of one enlarged corner, marshalled
struct CornerIn { DT C, T1, T2, a; };
int corner_in(IO& io, const double* const* t, int chi, const int* ad, int corner, CornerIn* ci) {
    const CornerSpec& sp = kCorner[corner];
    const long long X = chi, D1 = (long long)ad[sp.t1_leg] * ad[sp.t1_leg], D2 = (long long)ad[sp.t2_leg] * ad[sp.t2_leg];
    CTM_TRY(io.in(t[0], {X, X}, &ci->C));
    CTM_TRY(io.in(t[1], {X, X, D1}, &ci->T1));       // only the element count matters here: corner_impl re-views
    CTM_TRY(io.in(t[2], {X, X, D2}, &ci->T2));
    return io.in(t[3], {ad[0], ad[1], ad[2], ad[3], ad[4]}, &ci->a);
}

// output extents of a corner: (n0, n1)
void corner_dims(int corner, int chi, const int* ad, long long* n0, long long* n1) {
    // LU: (chi*Dd^2, chi*Dr^2)  RU: (chi*Dl^2, chi*Dd^2)  RD: (chi*Du^2, chi*Dl^2)  LD: (chi*Du^2, chi*Dr^2)
    static const int leg0[4] = {3, 2, 1, 1}, leg1[4] = {4, 3, 2, 4};
    *n0 = (long long)chi * ad[leg0[corner]] * ad[leg0[corner]];
    *n1 = (long long)chi * ad[leg1[corner]] * ad[leg1[corner]];
}

// halves table (oracle _HALVES): corner ids and N/T ops for R and Rt; tensors16 are in the reference's order
struct HalfSpec { int cA, cB; int tA, tB; };
const HalfSpec kHalves[4][2] = {
    /* UP    */ {{CTM_RU, CTM_RD, 0, 0}, {CTM_LU, CTM_LD, 1, 0}},
    /* LEFT  */ {{CTM_LU, CTM_RU, 0, 0}, {CTM_LD, CTM_RD, 0, 1}},
    /* DOWN  */ {{CTM_LD, CTM_LU, 1, 0}, {CTM_RD, CTM_RU, 1, 1}},
    /* RIGHT */ {{CTM_RD, CTM_LD, 0, 1}, {CTM_RU, CTM_LU, 1, 1}},
};

// ---------------------------------------------------------------------------------------------
// absorb table (oracle _ABSORB); operand order nC1 <- (Pt1,C1,T1), nC2 <- (C2,T2,P2), nT <- (T,Pt2,A,A*,P1)
// ---------------------------------------------------------------------------------------------
struct AbsorbSpec {
    const char *nC1, *nC2, *nT;
    int t_axis, t_leg, pt2_leg, p1_leg;    // T split axis / site leg; site legs splitting Pt2 and P1
    int t1_axis, t2_axis;                   // position of the D^2 axis in T1 and T2 (3-index tensors)
    int fuse0;                              // first of the two adjacent output axes of nT that are fused
};
const AbsorbSpec kAbsorb[4] = {
    /* UP    */ {"abk,ac,cbd->kd", "ca,cdb,abk->dk", "abcd,aije,mbifk,mcjgl,dklh->efgh", 1, 1, 2, 4, 1, 2, 1},
    /* LEFT  */ {"abk,ac,cbd->kd", "ac,bcd,abk->kd", "abcd,bghm,iecgk,ifdhl,aefj->jmkl", 2, 2, 3, 1, 1, 0, 2},
    /* DOWN  */ {"abk,ca,dcb->dk", "ca,dbc,abk->dk", "abcd,dklh,mfiak,mgjbl,cije->fgeh", 0, 3, 4, 2, 2, 1, 0},
    /* RIGHT */ {"abk,ac,bdc->kd", "ca,dbc,abk->dk", "abcd,aefj,iekgb,iflhc,dghm->jklm", 1, 4, 1, 3, 0, 1, 1},
};

}  // namespace

// =================================================================================================
extern "C" {

int ctm_einsum(ctm_ctx* ctx, const char* expr, int ntensors, const double* const* tensors, const int* ndims, const long long* dims,
               const int* conj, double* out) {
    return ctm_entry(ctx, "ctm_einsum", [&]() -> int {
    if (!expr || ntensors < 2 || ntensors > 16) { ctx->set_error("einsum: 2..16 operands"); return CTM_ERR_BADARG; }
    const std::string e(expr);
    const size_t arrow = e.find("->");
    if (arrow == std::string::npos) { ctx->set_error("einsum: explicit output indices required"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    IO io(ctx);
    std::vector<DT> ops(ntensors);
    std::vector<std::string> ins;
    { const std::string lhs = e.substr(0, arrow); size_t s0 = 0; while (true) { size_t c = lhs.find(',', s0); ins.push_back(lhs.substr(s0, c == std::string::npos ? c : c - s0)); if (c == std::string::npos) break; s0 = c + 1; } }
    if ((int)ins.size() != ntensors) { ctx->set_error("einsum: operand count does not match the expression"); return CTM_ERR_BADARG; }
    std::map<char, long long> ext;
    const long long* dp = dims;
    for (int i = 0; i < ntensors; ++i) {
        if (ndims[i] != (int)ins[i].size() || ndims[i] > CTM_MAXD) { ctx->set_error("einsum: rank of operand " + std::to_string(i)); return CTM_ERR_SHAPE; }
        std::vector<long long> d(dp, dp + ndims[i]);
        for (int a = 0; a < ndims[i]; ++a) {
            auto it = ext.find(ins[i][a]);
            if (it != ext.end() && it->second != d[a]) { ctx->set_error(std::string("einsum: extent mismatch on index ") + ins[i][a]); return CTM_ERR_SHAPE; }
            ext[ins[i][a]] = d[a];
        }
        dp += ndims[i];
        // identical device pointers share one marshalled copy so that the fused (a, conj a) detection still sees one tensor
        int same = -1;
        for (int j = 0; j < i; ++j) if (tensors[j] == tensors[i] && ops[j].numel() == [&] { long long n = 1; for (auto v : d) n *= v; return n; }()) { same = j; break; }
        if (same >= 0) { ops[i] = ops[same]; ops[i].dims = d; ops[i].cj = false; }
        else CTM_TRY(io.in(tensors[i], d, &ops[i]));
        if (conj && conj[i]) ops[i] = ops[i].conj();
    }
    const std::string o = e.substr(arrow + 2);
    size_t on = 1;
    for (char ch : o) { auto it = ext.find(ch); if (it == ext.end()) { ctx->set_error("einsum: unknown output index"); return CTM_ERR_BADARG; } on *= (size_t)it->second; }
    DT res;
    CTM_TRY(io.out(out, on, &res));
    ArenaScope work(ctx);
    CTM_TRY(dev_network(ctx, e, ops, &res));
    return io.finish();
    });
}

int ctm_c2x2(ctm_ctx* ctx, int corner, int open, const double* C, const double* T1, const double* T2, const double* a,
             int chi, const int* adims, double* out) {
    return ctm_entry(ctx, "ctm_c2x2", [&]() -> int {
    if (corner < 0 || corner > 3) { ctx->set_error("c2x2: bad corner"); return CTM_ERR_BADARG; }
    PhaseTimer pt(ctx, CTM_T_CORNERS);
    ArenaScope scope(ctx);
    IO io(ctx);
    const double* t4[4] = {C, T1, T2, a};
    CornerIn ci;
    CTM_TRY(corner_in(io, t4, chi, adims, corner, &ci));
    long long n0, n1; corner_dims(corner, chi, adims, &n0, &n1);
    const long long pp = open ? (long long)adims[0] * adims[0] : 1;
    DT res;
    CTM_TRY(io.out(out, (size_t)(n0 * n1 * pp), &res));
    CTM_TRY(corner_impl(ctx, corner, open, ci.C, ci.T1, ci.T2, ci.a, chi, adims, &res));
    return io.finish();
    });
}

int ctm_halves(ctm_ctx* ctx, int dir, const double* const* t, int chi, const int* adims4x5, double* R, double* Rt) {
    return ctm_entry(ctx, "ctm_halves", [&]() -> int {
    if (dir < 0 || dir > 3) { ctx->set_error("halves: bad direction"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    IO io(ctx);
    double* outs[2] = {R, Rt};
    for (int h = 0; h < 2; ++h) {
        const HalfSpec& hs = kHalves[dir][h];
        const int ia = 2 * h, ib = 2 * h + 1;       // corner slots in tensors16: (A of R, B of R, A of Rt, B of Rt)
        long long a0, a1, b0, b1;
        corner_dims(hs.cA, chi, adims4x5 + 5 * ia, &a0, &a1);
        corner_dims(hs.cB, chi, adims4x5 + 5 * ib, &b0, &b1);
        const long long M = hs.tA ? a1 : a0, Ka = hs.tA ? a0 : a1;
        const long long N = hs.tB ? b0 : b1, Kb = hs.tB ? b1 : b0;
        if (Ka != Kb) { ctx->set_error("halves: contracted dims differ"); return CTM_ERR_SHAPE; }
        DT res;
        CTM_TRY(io.out(outs[h], (size_t)(M * N), &res));
        ArenaScope inner(ctx);
        DT cA, cB;
        CTM_TRY(alloc_dt(ctx, {a0, a1}, &cA));
        CTM_TRY(alloc_dt(ctx, {b0, b1}, &cB));
        {
            PhaseTimer pt(ctx, CTM_T_CORNERS);
            { ArenaScope s2(ctx); CornerIn ci; CTM_TRY(corner_in(io, t + 4 * ia, chi, adims4x5 + 5 * ia, hs.cA, &ci));
              CTM_TRY(corner_impl(ctx, hs.cA, 0, ci.C, ci.T1, ci.T2, ci.a, chi, adims4x5 + 5 * ia, &cA)); }
            { ArenaScope s2(ctx); CornerIn ci; CTM_TRY(corner_in(io, t + 4 * ib, chi, adims4x5 + 5 * ib, hs.cB, &ci));
              CTM_TRY(corner_impl(ctx, hs.cB, 0, ci.C, ci.T1, ci.T2, ci.a, chi, adims4x5 + 5 * ib, &cB)); }
        }
        PhaseTimer pt(ctx, CTM_T_HALVES);
        CTM_TRY(xgemm(ctx, (int)M, (int)N, (int)Ka, xm(cA, a1, hs.tA != 0), xm(cB, b1, hs.tB != 0), res.p, res.q, N));
    }
    return io.finish();
    });
}

int ctm_truncated_svd(ctm_ctx* ctx, const double* M, int n, int chi, const ctm_trunc_cfg* cfg_, double* U, double* S, double* V) {
    return ctm_entry(ctx, "ctm_truncated_svd", [&]() -> int {
    return ctm_truncated_svd_ws(ctx, M, n, chi, cfg_, U, S, V, nullptr);
    });
}

int ctm_truncated_svd_ws(ctm_ctx* ctx, const double* M, int n, int chi, const ctm_trunc_cfg* cfg_, double* U, double* S, double* V,
                         double* basis) {
    return ctm_entry(ctx, "ctm_truncated_svd_ws", [&]() -> int {
    const ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (chi < 1 || n < 1) { ctx->set_error("truncated_svd: bad dims"); return CTM_ERR_BADARG; }
    PhaseTimer pt(ctx, CTM_T_SVD);
    ArenaScope scope(ctx);
    IO io(ctx);
    const int k = (chi < n) ? chi + 1 : n, kc = std::min(chi, n), cz = ctx->cplx ? 2 : 1;
    DT tM, tU, tV;
    CTM_TRY(io.in(M, {n, n}, &tM));
    CTM_TRY(io.out(U, (size_t)n * kc, &tU));
    CTM_TRY(io.out(V, (size_t)n * kc, &tV));
    double *Ut, *Vt, *dS;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n * cz, (void**)&Ut));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n * cz, (void**)&Vt));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&dS));
    TruncOut to;
    CTM_TRY(svd_rows(ctx, tM, n, chi, cfg, Ut, Vt, dS, &to, basis));
    std::vector<double> Sh(kc);
    for (int i = 0; i < kc; ++i) Sh[i] = (i <= to.keep_last) ? to.S[i] : 0.0;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, Sh.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    CTM_TRY(rows_to_cols(ctx, Ut, k, n, kc, to.keep_last, tU));
    CTM_TRY(rows_to_cols(ctx, Vt, k, n, kc, to.keep_last, tV));
    CTM_TRY(io.finish());
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
    });
}

namespace {
int truncated_eigh_impl(ctm_ctx* ctx, const double* A, int n, int chi, const ctm_trunc_cfg* cfg_, double* D, double* U, double* warm);
int eigh_trunc_planar(ctm_ctx* ctx, const DT& A, int n, int chi, const ctm_trunc_cfg& cfg, double* dD, const DT& U, double* warm);
}

int ctm_truncated_eigh(ctm_ctx* ctx, const double* A, int n, int chi, const ctm_trunc_cfg* cfg_, double* D, double* U) {
    return ctm_entry(ctx, "ctm_truncated_eigh", [&]() -> int {
    return truncated_eigh_impl(ctx, A, n, chi, cfg_, D, U, nullptr);
    });
}

int ctm_truncated_eigh_ws(ctm_ctx* ctx, const double* A, int n, int chi, const ctm_trunc_cfg* cfg_, double* D, double* U, double* basis) {
    return ctm_entry(ctx, "ctm_truncated_eigh_ws", [&]() -> int {
    return truncated_eigh_impl(ctx, A, n, chi, cfg_, D, U, basis);
    });
}

namespace {
// Complex eigenvectors are defined up to a phase; the solver's phases drift with rounding, and a C4v run whose projector phases
// drift never sees the same enlarged corner twice (its environment legs are re-gauged every sweep).  Canonical choice per row
// u^H (planar k x n): the component of largest modulus (first one on ties at the 2^-40 level, like fix_svd_signs) real and positive.
__global__ __launch_bounds__(256) void canon_phase_rows_kernel(double* __restrict__ re, double* __restrict__ im, int n) {
    __shared__ double best[256];
    __shared__ int where[256];
    double* xr = re + (size_t)blockIdx.x * n;
    double* xi = im + (size_t)blockIdx.x * n;
    double b = -1.0; int w = 0;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const double m = floor((xr[c] * xr[c] + xi[c] * xi[c]) * 1099511627776.0);
        if (m > b) { b = m; w = c; }
    }
    best[threadIdx.x] = b; where[threadIdx.x] = w;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const double ob = best[threadIdx.x + off]; const int ow = where[threadIdx.x + off];
            if (ob > best[threadIdx.x] || (ob == best[threadIdx.x] && ow < where[threadIdx.x])) { best[threadIdx.x] = ob; where[threadIdx.x] = ow; }
        }
        __syncthreads();
    }
    const int c0 = where[0];
    const double pr = xr[c0], pi = xi[c0], m = sqrt(pr * pr + pi * pi);
    if (!(m > 0.0)) return;
    const double cr = pr / m, ci = -pi / m;        // multiply the row by conj(phase)
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const double a = xr[c], d = xi[c];
        xr[c] = a * cr - d * ci; xi[c] = a * ci + d * cr;
    }
}

// truncated_eig_sym (custom_eig.py:7-65) on a planar (real or complex Hermitian) matrix: dD (device, min(chi,n), signed, zeros beyond
// the last complete multiplet), U (n x min(chi,n), planar in complex contexts, columns beyond the kept multiplets zeroed)
int eigh_trunc_planar(ctm_ctx* ctx, const DT& A, int n, int chi, const ctm_trunc_cfg& cfg, double* dD, const DT& U, double* warm) {
    PhaseTimer pt(ctx, CTM_T_EIG);
    ArenaScope scope(ctx);
    const int k = (chi < n) ? chi + 1 : n, kc = std::min(chi, n), cz = A.q ? 2 : 1;
    double *Ut, *dk;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n * cz, (void**)&Ut));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&dk));
    if (A.q) {
        CTM_TRY(jacobi_eigh_top_c(ctx, A.p, A.q, n, k, dk, Ut, warm));
        CTM_LAUNCH(ctx, canon_phase_rows_kernel, dim3(k), dim3(256), 0, Ut, Ut + (size_t)k * n, n);
    } else CTM_TRY(jacobi_eigh_top(ctx, A.p, n, k, dk, Ut, warm));
    std::vector<double> Dh(k);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(Dh.data(), dk, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    int keep_last = kc - 1;
    if (cfg.keep_multiplets && chi < n) keep_last = std::min(kc - 1, multiplet_chi(Dh, chi, cfg.eps_multiplet, cfg.multiplet_abstol));
    std::vector<double> Do(kc);
    for (int i = 0; i < kc; ++i) Do[i] = (i <= keep_last) ? Dh[i] : 0.0;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(dD, Do.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    CTM_TRY(rows_to_cols(ctx, Ut, k, n, kc, keep_last, U));       // columns u_i from the rows u_i^H
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
}

int truncated_eigh_impl(ctm_ctx* ctx, const double* A, int n, int chi, const ctm_trunc_cfg* cfg_, double* D, double* U, double* warm) {
    ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (!cfg_) { cfg.eps_multiplet = 1.0e-12; }
    if (chi < 1 || n < 1) { ctx->set_error("truncated_eigh: bad dims"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    IO io(ctx);
    const int kc = std::min(chi, n);
    DT tA, tU;
    CTM_TRY(io.in(A, {n, n}, &tA));
    CTM_TRY(io.out(U, (size_t)n * kc, &tU));
    CTM_TRY(eigh_trunc_planar(ctx, tA, n, chi, cfg, D, tU, warm));
    CTM_TRY(io.finish());
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
}
}  // namespace

// svd_symeig (linalg/svd_symeig.py:12-34): V[:, j] = U[:, j] sign(D[j]), S[j] = |D[j]|  (sign(0) = 0 as torch.sign)
__global__ void symeig_to_svd_kernel(const double* __restrict__ U, const double* __restrict__ D, int n, int kc, double* __restrict__ V,
                                     double* __restrict__ S) {
    const size_t tot = (size_t)n * kc;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(q % kc);
        const double d = D[j];
        V[q] = U[q] * (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0));
        if (q < (size_t)kc) S[q] = fabs(D[q]);
    }
}

int ctm_svd_symeig(ctm_ctx* ctx, const double* A, int n, int chi, const ctm_trunc_cfg* cfg_, double* U, double* S, double* V) {
    return ctm_entry(ctx, "ctm_svd_symeig", [&]() -> int {
    ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (!cfg_) { cfg.eps_multiplet = 1.0e-12; cfg.keep_multiplets = 0; }
    if (chi < 1 || n < 1) { ctx->set_error("svd_symeig: bad dims"); return CTM_ERR_BADARG; }
    if (ctx->cplx) { ctx->set_error("svd_symeig: real symmetric matrices only (the reference uses torch.symeig)"); return CTM_ERR_UNSUPPORTED; }
    ArenaScope scope(ctx);
    const int kc = std::min(chi, n);
    double* D;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * kc, (void**)&D));
    CTM_TRY(truncated_eigh_impl(ctx, A, n, chi, &cfg, D, U, nullptr));
    const size_t tot = (size_t)n * kc;
    CTM_LAUNCH(ctx, symeig_to_svd_kernel, dim3((int)std::min<size_t>((tot + 255) / 256, 2048)), dim3(256), 0,
                       (const double*)U, (const double*)D, n, kc, V, S);
    CTM_HIP_CHECK(ctx, hipGetLastError());
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
    });
}

int ctm_svdvals(ctm_ctx* ctx, const double* M, int n, double* S) {
    return ctm_entry(ctx, "ctm_svdvals", [&]() -> int {
    ArenaScope scope(ctx);
    IO io(ctx);
    DT tM;
    CTM_TRY(io.in(M, {n, n}, &tM));
    return jacobi_svdvals(ctx, tM.p, tM.q, n, S);
    });
}

namespace {
// shared tail of the projector constructions: scale vector from the kept spectrum
int upload_scale(ctm_ctx* ctx, const TruncOut& to, int kc, double reltol, double* dScale, double* S_out, int* ncol) {
    std::vector<double> Sh, sc;
    proj_scale(to, kc, reltol, &Sh, &sc);
    *ncol = 0;
    for (int i = 0; i < kc; ++i) if (sc[i] != 0.0) *ncol = i + 1;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(dScale, sc.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    if (S_out) CTM_HIP_CHECK(ctx, hipMemcpyAsync(S_out, Sh.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));     // host vectors go out of scope
    return CTM_OK;
}
}  // namespace

int ctm_projectors(ctm_ctx* ctx, const double* R, const double* Rt, int n, int chi, const ctm_trunc_cfg* cfg_, double* P,
                   double* Pt, double* S_out) {
    return ctm_entry(ctx, "ctm_projectors", [&]() -> int {
    return ctm_projectors_rect(ctx, R, Rt, n, n, chi, cfg_, P, Pt, S_out);
    });
}

// R, Rt: a x b (a = the bond that is truncated, b = the far side of the half; the reference only asserts R.shape == Rt.shape,
// ctm_projectors.py:209): M = R^T Rt is b x b, P = R conj(U) S^-1/2 and Pt = Rt V S^-1/2 are a x min(chi, b)
int ctm_projectors_rect(ctm_ctx* ctx, const double* R, const double* Rt, int a, int b, int chi, const ctm_trunc_cfg* cfg_, double* P,
                        double* Pt, double* S_out) {
    return ctm_entry(ctx, "ctm_projectors_rect", [&]() -> int {
    const ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (chi < 1 || a < 1 || b < 1) { ctx->set_error("projectors: bad dims"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    IO io(ctx);
    const int n = b;
    const int k = (chi < n) ? chi + 1 : n, kc = std::min(chi, n), cz = ctx->cplx ? 2 : 1;
    DT tR, tRt, tM, tP, tPt;
    CTM_TRY(io.in(R, {a, n}, &tR));
    CTM_TRY(io.in(Rt, {a, n}, &tRt));
    CTM_TRY(io.out(P, (size_t)a * kc, &tP));
    CTM_TRY(io.out(Pt, (size_t)a * kc, &tPt));
    CTM_TRY(alloc_dt(ctx, {n, n}, &tM));
    double *Ut, *Vt, *dS, *dScale;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n * cz, (void**)&Ut));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n * cz, (void**)&Vt));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&dS));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * kc, (void**)&dScale));
    {   // M = R^T Rt  (ctm_projectors.py:263, plain transpose)
        PhaseTimer pt(ctx, CTM_T_HALVES);
        CTM_TRY(xgemm(ctx, n, n, a, xm(tR, n, true), xm(tRt, n, false), tM.p, tM.q, n));
    }
    TruncOut to;
    { PhaseTimer pt(ctx, CTM_T_SVD); CTM_TRY(svd_rows(ctx, tM, n, chi, cfg, Ut, Vt, dS, &to)); }
    PhaseTimer pt(ctx, CTM_T_PROJ);
    int ncol;
    CTM_TRY(upload_scale(ctx, to, kc, cfg.svd_reltol, dScale, S_out, &ncol));
    // P = R conj(U) diag(S_sqrt),  Pt = Rt V diag(S_sqrt)   (:283); with row factors Ut = U^H, Vt = V^H:
    // conj(U) = Ut^T (plain transpose), V = Vt^H -> NT GEMMs + fused column scale; zero-scale columns are not computed
    const size_t kn = (size_t)k * n;
    XM u{Ut, ctx->cplx ? Ut + kn : nullptr, n, true, false}, v{Vt, ctx->cplx ? Vt + kn : nullptr, n, true, true};
    CTM_TRY(fill_f64(ctx, tP.p, (size_t)a * kc * (tP.q ? 2 : 1), 0.0));
    CTM_TRY(fill_f64(ctx, tPt.p, (size_t)a * kc * (tPt.q ? 2 : 1), 0.0));
    if (ncol > 0) {
        CTM_TRY(xgemm(ctx, a, ncol, n, xm(tR, n, false), u, tP.p, tP.q, kc, dScale));
        CTM_TRY(xgemm(ctx, a, ncol, n, xm(tRt, n, false), v, tPt.p, tPt.q, kc, dScale));
    }
    CTM_TRY(io.finish());
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
    });
}

int ctm_projectors_4x4(ctm_ctx* ctx, int dir, const double* const* t, int chi, const int* adims4x5, const ctm_trunc_cfg* cfg_,
                       double* P, double* Pt, double* S_out) {
    return ctm_entry(ctx, "ctm_projectors_4x4", [&]() -> int {
    return ctm_projectors_4x4_ws(ctx, dir, t, chi, adims4x5, cfg_, P, Pt, S_out, nullptr);
    });
}

int ctm_projectors_4x4_ws(ctm_ctx* ctx, int dir, const double* const* t, int chi, const int* adims4x5, const ctm_trunc_cfg* cfg_,
                          double* P, double* Pt, double* S_out, double* basis) {
    return ctm_entry(ctx, "ctm_projectors_4x4_ws", [&]() -> int {
    return ctm_projectors_4x4_cc(ctx, dir, t, chi, adims4x5, cfg_, P, Pt, S_out, basis, nullptr, nullptr);
    });
}

int ctm_projectors_4x4_cc(ctm_ctx* ctx, int dir, const double* const* t, int chi, const int* adims4x5, const ctm_trunc_cfg* cfg_,
                          double* P, double* Pt, double* S_out, double* basis, double* const* corner_buf, const int* corner_valid) {
    return ctm_entry(ctx, "ctm_projectors_4x4_cc", [&]() -> int {
    const ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (dir < 0 || dir > 3) { ctx->set_error("projectors_4x4: bad direction"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    IO io(ctx);
    // the four enlarged corners of the move (reference order: A,B of R then A,B of Rt)
    DT c[4]; long long d0[4], d1[4]; int cid[4]; bool tr[4];
    for (int h = 0; h < 2; ++h) {
        const HalfSpec& hs = kHalves[dir][h];
        cid[2 * h] = hs.cA; cid[2 * h + 1] = hs.cB; tr[2 * h] = hs.tA != 0; tr[2 * h + 1] = hs.tB != 0;
    }
    for (int i = 0; i < 4; ++i) corner_dims(cid[i], chi, adims4x5 + 5 * i, &d0[i], &d1[i]);
    // R = opA(c0) opB(c1): (n x mid0)(mid0 x n) ; Rt = opC(c2) opD(c3): (n x mid1)(mid1 x n).  The truncated bond n must be
    // the same on both halves (bond dimensions may differ between the lattice directions, not along one cut).
    auto rows_of = [&](int i) { return tr[i] ? d1[i] : d0[i]; };
    auto cols_of = [&](int i) { return tr[i] ? d0[i] : d1[i]; };
    const long long n = rows_of(0), mid0 = cols_of(0), mid1 = cols_of(2);
    if (rows_of(1) != mid0 || cols_of(1) != n || rows_of(2) != n || rows_of(3) != mid1 || cols_of(3) != n) {
        ctx->set_error("projectors_4x4: the four enlarged corners do not chain to square halves (bond dimensions differ along one cut)");
        return CTM_ERR_UNSUPPORTED;
    }
    const int k = (chi < n) ? chi + 1 : (int)n, kc = std::min(chi, (int)n), cz = ctx->cplx ? 2 : 1;
    DT tP, tPt;
    CTM_TRY(io.out(P, (size_t)n * kc, &tP));
    CTM_TRY(io.out(Pt, (size_t)n * kc, &tPt));
    {
        PhaseTimer pt(ctx, CTM_T_CORNERS);
        for (int i = 0; i < 4; ++i) {
            // caller-owned corner buffers (opaque, engine-internal layout): a valid one is used as it is, an invalid one is
            // (re)computed in place for the caller to keep -- an enlarged corner only changes when one of ITS three
            // environment tensors does, i.e. two of the four corner types survive every directional move
            double* ext = corner_buf ? corner_buf[i] : nullptr;
            if (ext) {
                if (((uintptr_t)ext & 15) != 0) { ctx->set_error("projectors_4x4: corner buffers must be 16-byte aligned"); return CTM_ERR_BADARG; }
                c[i].p = ext; c[i].q = ctx->cplx ? ext + d0[i] * d1[i] : nullptr; c[i].dims = {d0[i], d1[i]}; c[i].cj = false;
                if (corner_valid && corner_valid[i]) { ctx->corner_cache_hits += 1; continue; }
            } else
                CTM_TRY(alloc_dt(ctx, {d0[i], d1[i]}, &c[i]));
            ArenaScope s2(ctx);
            CornerIn ci;
            CTM_TRY(corner_in(io, t + 4 * i, chi, adims4x5 + 5 * i, cid[i], &ci));
            CTM_TRY(corner_impl(ctx, cid[i], 0, ci.C, ci.T1, ci.T2, ci.a, chi, adims4x5 + 5 * i, &c[i]));
        }
    }
    double *Ut, *Vt, *dS, *dScale;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n * cz, (void**)&Ut));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n * cz, (void**)&Vt));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&dS));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * kc, (void**)&dScale));
    MatOp op; op.n = (int)n;
    for (int i = 0; i < 4; ++i) { op.c[i] = c[i].p; op.ci[i] = c[i].q; op.t[i] = tr[i]; }
    op.mid[0] = (int)mid0; op.mid[1] = (int)mid1;
    op.warm = basis;
    op.warm_hdr = basis ? basis + (size_t)cz * k * n : nullptr;      // header row behind the basis (see ctm_hip.h)
    bool have_mid = false;
    if (!ctx->cplx && ctx->proj_from_krylov) {
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&op.out_uR));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&op.out_vRt));
        op.have_mid = &have_mid;
    }
    TruncOut to;
    { PhaseTimer pt(ctx, CTM_T_SVD); CTM_TRY(svd_rows_op(ctx, op, chi, cfg, Ut, Vt, dS, &to)); }
    PhaseTimer pt(ctx, CTM_T_PROJ);
    int ncol;
    CTM_TRY(upload_scale(ctx, to, kc, cfg.svd_reltol, dScale, S_out, &ncol));
    if (have_mid) {
        // the block Krylov solver left u_i^T R^T and v_i^T Rt^T (rows, sign-fixed with their triplets): P[:, j] = (R u_j) s_j^-1/2
        // and Pt[:, j] = (Rt v_j) s_j^-1/2 are their scaled transposes -- no corner passes
        CTM_TRY(fill_f64(ctx, tP.p, (size_t)n * kc, 0.0));
        CTM_TRY(fill_f64(ctx, tPt.p, (size_t)n * kc, 0.0));
        if (ncol > 0) {
            const long long tot = (long long)n * ncol;
            const int blocks = (int)std::min<long long>((tot + 255) / 256, 2048);
            CTM_LAUNCH(ctx, transpose_scale_kernel, dim3(blocks), dim3(256), 0, (const double*)op.out_uR, ncol, (int)n, tP.p, (long long)kc, (const double*)dScale);
            CTM_LAUNCH(ctx, transpose_scale_kernel, dim3(blocks), dim3(256), 0, (const double*)op.out_vRt, ncol, (int)n, tPt.p, (long long)kc, (const double*)dScale);
        }
        CTM_TRY(io.finish());
        CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return CTM_OK;
    }
    // P = R conj(U) S^-1/2 = opA(cA) opB(cB) Ut^T ... ; Pt = Rt V S^-1/2 = opC(cC) opD(cD) Vt^H ...
    CTM_TRY(corner_chain_times_rowsT(ctx, (int)n, (int)mid0, k, kc, ncol, c[0], tr[0], c[1], tr[1], Ut, false, dScale, tP));
    CTM_TRY(corner_chain_times_rowsT(ctx, (int)n, (int)mid1, k, kc, ncol, c[2], tr[2], c[3], tr[3], Vt, true, dScale, tPt));
    CTM_TRY(io.finish());
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
    });
}

int ctm_absorb(ctm_ctx* ctx, int dir, const double* const* t, int chi, const int* ad, int normalize, double* nC1,
               double* nC2, double* nT) {
    return ctm_entry(ctx, "ctm_absorb", [&]() -> int {
    return ctm_absorb_x(ctx, dir, t, chi, chi, ad, normalize, nC1, nC2, nT);
    });
}

int ctm_absorb_x(ctm_ctx* ctx, int dir, const double* const* t, int chi_in, int chi_out, const int* ad, int normalize, double* nC1,
                 double* nC2, double* nT) {
    return ctm_entry(ctx, "ctm_absorb_x", [&]() -> int {
    if (dir < 0 || dir > 3) { ctx->set_error("absorb: bad direction"); return CTM_ERR_BADARG; }
    const AbsorbSpec& sp = kAbsorb[dir];
    // X: environment dimension of the incoming tensors; Y: dimension of the truncated (new) bond = columns of the projectors.
    // D^2 extents of the T tensors follow the site legs they attach to (uniform D assumed per leg pair)
    const long long X = chi_in, Y = chi_out, Dt = ad[sp.t_leg], Dpt2 = ad[sp.pt2_leg], Dp1 = ad[sp.p1_leg];
    // T1 carries the D^2 leg shared with Pt1, T2 the one shared with P2
    const long long Dt1 = Dp1 /* neighbour projector leg == this site's leg on that side */, Dt2 = Dpt2;
    auto d3 = [&](int axis, long long D2) { std::vector<long long> d; for (int a = 0; a < 3; ++a) d.push_back(a == axis ? D2 : X); return d; };
    // nT has one D^2 leg: the site leg opposite to the absorbed T (UP:d, LEFT:r, DOWN:u, RIGHT:l)
    static const int out_leg[4] = {3, 4, 1, 2};
    const long long D2out = (long long)ad[out_leg[dir]] * ad[out_leg[dir]];
    ArenaScope scope(ctx);
    IO io(ctx);
    DT r1, r2, r3;
    {   // algorithmic bytes: four projector blocks, three T tensors, two corners and the site read once, the three results written once
        const double el = ctx->cplx ? 16.0 : 8.0;
        ctx->absorb_bytes += el * ((double)X * Y * (2.0 * Dt1 * Dt1 + Dpt2 * Dpt2 + Dp1 * Dp1) + (double)X * X * (Dt1 * Dt1 + Dt * Dt + Dt2 * Dt2 + 2.0)
                                   + (double)ad[0] * ad[1] * ad[2] * ad[3] * ad[4] + 2.0 * X * Y + (double)Y * Y * D2out);
        ctx->absorb_calls += 1;
    }
    {
        PhaseTimer pt(ctx, CTM_T_ABSORB);
        DT tC1, tT1, tT, tT2, tC2, tA, tP2, tPt2, tP1, tPt1;
        CTM_TRY(io.in(t[0], {X, X}, &tC1));
        CTM_TRY(io.in(t[1], d3(sp.t1_axis, Dt1 * Dt1), &tT1));
        CTM_TRY(io.in(t[2], t_view(sp.t_axis, chi_in, Dt), &tT));
        CTM_TRY(io.in(t[3], d3(sp.t2_axis, Dt2 * Dt2), &tT2));
        CTM_TRY(io.in(t[4], {X, X}, &tC2));
        CTM_TRY(io.in(t[5], {ad[0], ad[1], ad[2], ad[3], ad[4]}, &tA));
        CTM_TRY(io.in(t[6], {X, Dt2 * Dt2, Y}, &tP2));
        CTM_TRY(io.in(t[7], {X, Dpt2, Dpt2, Y}, &tPt2));
        CTM_TRY(io.in(t[8], {X, Dp1, Dp1, Y}, &tP1));
        CTM_TRY(io.in(t[9], {X, Dt1 * Dt1, Y}, &tPt1));
        CTM_TRY(io.out(nC1, (size_t)(X * Y), &r1));          // one new (Y) and one old (X) leg; which is first follows the spec
        CTM_TRY(io.out(nC2, (size_t)(X * Y), &r2));
        CTM_TRY(io.out(nT, (size_t)(Y * Y * D2out), &r3));
        ArenaScope work(ctx);
        CTM_TRY(dev_seq_einsum(ctx, sp.nC1, {tPt1, tC1, tT1}, &r1));
        CTM_TRY(dev_seq_einsum(ctx, sp.nC2, {tC2, tT2, tP2}, &r2));
        CTM_TRY(dev_network(ctx, sp.nT, {tT, tPt2, tA, tA.conj(), tP1}, &r3));
        (void)sp.fuse0;   // the fused output axes are adjacent: the 4-index result IS the 3-index tensor in memory
    }
    if (normalize) {
        PhaseTimer pt(ctx, CTM_T_NORM);
        const int kind = (normalize == 2) ? 2 : 1;
        CTM_TRY(normalize_dt(ctx, r1.view({X * Y}), kind));
        CTM_TRY(normalize_dt(ctx, r2.view({X * Y}), kind));
        CTM_TRY(normalize_dt(ctx, r3.view({Y * Y * D2out}), kind));
    }
    return io.finish();
    });
}

// ---- C4v -------------------------------------------------------------------------------------------
int ctm_c2x2_c4v(ctm_ctx* ctx, int open, const double* a, const double* C, const double* T, int chi, int p, int D, double* out) {
    return ctm_entry(ctx, "ctm_c2x2_c4v", [&]() -> int {
    PhaseTimer pt(ctx, CTM_T_CORNERS);
    ArenaScope scope(ctx);
    IO io(ctx);
    DT tC, tT, tA, res;
    CTM_TRY(io.in(C, {chi, chi}, &tC));
    CTM_TRY(io.in(T, {chi, chi, D, D}, &tT));
    CTM_TRY(io.in(a, {p, D, D, D, D}, &tA));
    const size_t n = (size_t)chi * D * D;
    CTM_TRY(io.out(out, n * n * (open ? (size_t)p * p : 1), &res));
    CTM_TRY(dev_network(ctx, open ? "xy,cyuU,xelL,suldr,tULDR->edDcrRst" : "xy,cyuU,xelL,suldr,sULDR->edDcrR",
                        {tC, tT, tT, tA, tA.conj()}, &res));
    return io.finish();
    });
}

int ctm_move_c4v(ctm_ctx* ctx, const double* a, const double* C, const double* T, int chi, int p, int D,
                 const ctm_trunc_cfg* cfg_, double* C_out, double* T_out, double* D_out) {
    return ctm_entry(ctx, "ctm_move_c4v", [&]() -> int {
    return ctm_move_c4v_ws(ctx, a, C, T, chi, p, D, cfg_, C_out, T_out, D_out, nullptr);
    });
}

int ctm_move_c4v_ws(ctm_ctx* ctx, const double* a, const double* C, const double* T, int chi, int p, int D,
                    const ctm_trunc_cfg* cfg_, double* C_out, double* T_out, double* D_out, double* basis) {
    return ctm_entry(ctx, "ctm_move_c4v_ws", [&]() -> int {
    return ctm_move_c4v_x(ctx, a, C, T, chi, p, D, cfg_, 1, C_out, T_out, D_out, basis);
    });
}

int ctm_move_c4v_x(ctm_ctx* ctx, const double* a, const double* C, const double* T, int chi, int p, int D,
                   const ctm_trunc_cfg* cfg_, int normalize, double* C_out, double* T_out, double* D_out, double* basis) {
    return ctm_entry(ctx, "ctm_move_c4v_x", [&]() -> int {
    ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (!cfg_) cfg.eps_multiplet = 1.0e-12;          // custom_eig.py default used by ctmrg_c4v.py:49-52
    const int n = chi * D * D;
    ArenaScope scope(ctx);
    IO io(ctx);
    DT tA, tC, tT, c2, tP, rC, rT;
    CTM_TRY(io.in(a, {p, D, D, D, D}, &tA));
    CTM_TRY(io.in(C, {chi, chi}, &tC));
    CTM_TRY(io.in(T, {chi, chi, D, D}, &tT));
    CTM_TRY(io.out(C_out, (size_t)chi * chi, &rC));
    CTM_TRY(io.out(T_out, (size_t)chi * chi * D * D, &rT));
    CTM_TRY(alloc_dt(ctx, {n, n}, &c2));
    CTM_TRY(alloc_dt(ctx, {n, chi}, &tP));
    double* Dv;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * chi, (void**)&Dv));
    {   // 1) enlarged corner (ctm_components_c4v.py:52-130)
        PhaseTimer pt(ctx, CTM_T_CORNERS);
        CTM_TRY(dev_network(ctx, "xy,cyuU,xelL,suldr,sULDR->edDcrR", {tC, tT, tT, tA, tA.conj()}, &c2));
    }
    // 2) projector: truncated (Hermitian) eigendecomposition, D real with sign (ctmrg_c4v.py:49-52, 360-374)
    CTM_TRY(eigh_trunc_planar(ctx, c2, n, chi, cfg, Dv, tP, basis));
    PhaseTimer pt(ctx, CTM_T_ABSORB);
    CTM_TRY(diag_to_matrix(ctx, Dv, rC.p, chi));                                       // C = diag(D) (:374), imaginary part zero
    if (rC.q) CTM_TRY(fill_f64(ctx, rC.q, (size_t)chi * chi, 0.0));
    if (D_out) CTM_HIP_CHECK(ctx, hipMemcpyAsync(D_out, Dv, sizeof(double) * chi, hipMemcpyDeviceToDevice, ctx->stream));
    // 3) nT = P . T . a . a* . P*, symmetrised with its conjugate transpose in the environment legs (:383-446)
    DT tPv = tP.view({chi, D, D, chi});
    DT res = rT.view({chi, chi, D, D});
    CTM_TRY(dev_network(ctx, "xuUi,xelL,suldr,sULDR,edDj->ijrR", {tPv, tT, tA, tA.conj(), tPv.conj()}, &res));
    if (rT.q) CTM_TRY(add_conj_transposed01_c128(ctx, rT.p, rT.q, chi, D * D));
    else CTM_TRY(add_transposed01(ctx, rT.p, chi, D * D));
    // C /= |C[0,0]| ; T /= max|T| ('inf') or T /= |T|_2   (_move_normalize_c, :182-197)
    CTM_TRY(div_by_device_scalar(ctx, rC.p, (size_t)chi * chi, Dv, 1));
    CTM_TRY(normalize_dt(ctx, rT.view({(long long)chi * chi * D * D}), normalize == 2 ? 2 : 1));
    return io.finish();
    });
}

// ---- RDMs ------------------------------------------------------------------------------------------
int ctm_rdm2x2(ctm_ctx* ctx, const double* const* t, int chi, const int* ad4, double* out) {
    return ctm_entry(ctx, "ctm_rdm2x2", [&]() -> int {
    return ctm_rdm2x2_part(ctx, t, chi, ad4, 0, -1, out);
    });
}

int ctm_rdm2x2_part(ctm_ctx* ctx, const double* const* t, int chi, const int* ad4, int lo0, int lo1, double* out) {
    return ctm_entry(ctx, "ctm_rdm2x2_part", [&]() -> int {
    // rdm.py:1390-1588.  The reference holds the four open corners (n^2 p^2 each) and the two open halves (n^2 p^4 each) at
    // once; here the physical legs are unrolled: the lower half is kept as p^4 slices L[(s2 t2 s3 t3)] (n x n), the upper
    // half is produced one slice U[(s0 t0 s1 t1)] at a time and reduced against all of L immediately, so the peak is
    // n^2 (p^4 + 2 p^2 + 1) elements instead of n^2 (2 p^4 + 4 p^2) -- same flops, every product a plain n x n x n GEMM.
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    IO io(ctx);
    static const int cid[4] = {CTM_LU, CTM_RU, CTM_RD, CTM_LD};
    long long n0[4], n1[4], pp[4];
    for (int i = 0; i < 4; ++i) { corner_dims(cid[i], chi, ad4 + 5 * i, &n0[i], &n1[i]); pp[i] = ad4[5 * i]; }
    const long long A = n0[0], ku = n1[0], B = n1[1], kl = n1[3];
    if (n0[1] != ku || n0[3] != A || n0[2] != B || n1[2] != kl) { ctx->set_error("rdm2x2: corner dimensions do not chain"); return CTM_ERR_SHAPE; }
    const long long AB = A * B;
    const long long Pu = pp[0] * pp[0] * pp[1] * pp[1], Pl_all = pp[3] * pp[3] * pp[2] * pp[2];
    if (AB >= (1LL << 31)) { ctx->set_error("rdm2x2: n^2 exceeds the GEMM K range"); return CTM_ERR_UNSUPPORTED; }
    // partial evaluation: only the lower-half slices cl in [lo0, lo1) (cl = (s3 t3) * p2^2 + (s2 t2)) are built and the raw block
    // R[(s0 t0 s1 t1), lo0:lo1] is returned -- the unit of work that is sharded over a group of GPUs or looped over on one
    // GPU when n^2 p^4 elements do not fit (peak n^2 (2 p^2 + 1 + (lo1 - lo0)) elements)
    const bool partial = lo1 >= 0;
    if (partial && (lo0 < 0 || lo1 <= lo0 || lo1 > Pl_all)) { ctx->set_error("rdm2x2_part: bad slice range"); return CTM_ERR_BADARG; }
    const long long L0 = partial ? lo0 : 0, Pl = partial ? (lo1 - lo0) : Pl_all;
    DT r, R, Lall;
    CTM_TRY(io.out(out, (size_t)(Pu * Pl), &r));
    if (partial) R = r; else CTM_TRY(alloc_dt(ctx, {Pu, Pl}, &R));
    CTM_TRY(alloc_dt(ctx, {Pl, AB}, &Lall));
    auto slice = [&](const DT& c, long long idx, long long rows, long long cols) {     // (s,t) slice of a [p][p][rows][cols] corner
        DT v = c.view({rows, cols});
        v.p = c.p + idx * rows * cols;
        if (c.q) v.q = c.q + idx * rows * cols;
        return v;
    };
    auto build = [&](int i, DT* c) -> int {
        CTM_TRY(alloc_dt(ctx, {pp[i], pp[i], n0[i], n1[i]}, c));
        ArenaScope s2(ctx);
        CornerIn ci;
        CTM_TRY(corner_in(io, t + 4 * i, chi, ad4 + 5 * i, cid[i], &ci));
        return corner_impl(ctx, cid[i], 2, ci.C, ci.T1, ci.T2, ci.a, chi, ad4 + 5 * i, c);
    };
    {   // lower half: L[(s2 t2 s3 t3)](a,b) = sum_k LD[s2,t2](a,k) RD[s3,t3](b,k)                  (rdm.py:1527-1528)
        ArenaScope s1(ctx);
        DT c3, c2;
        CTM_TRY(build(3, &c3));
        CTM_TRY(build(2, &c2));
        for (long long i3 = 0; i3 < pp[3] * pp[3]; ++i3)
            for (long long i2 = 0; i2 < pp[2] * pp[2]; ++i2) {
                const DT x = slice(c3, i3, A, kl), y = slice(c2, i2, B, kl);
                const long long cl = i3 * pp[2] * pp[2] + i2 - L0;
                if (cl < 0 || cl >= Pl) continue;
                CTM_TRY(xgemm(ctx, (int)A, (int)B, (int)kl, xm(x, kl, false), xm(y, kl, true), Lall.p + cl * AB,
                              Lall.q ? Lall.q + cl * AB : nullptr, B));
            }
    }
    {   // upper half, one slice at a time: U(a,b) = sum_k LU[s0,t0](a,k) RU[s1,t1](k,b); R[(s0 t0 s1 t1), :] = <U, L[:]>   (:1459-1460, 1581)
        ArenaScope s1(ctx);
        DT c0, c1, U;
        CTM_TRY(build(0, &c0));
        CTM_TRY(build(1, &c1));
        CTM_TRY(alloc_dt(ctx, {A, B}, &U));
        for (long long i0 = 0; i0 < pp[0] * pp[0]; ++i0)
            for (long long i1 = 0; i1 < pp[1] * pp[1]; ++i1) {
                const DT x = slice(c0, i0, A, ku), y = slice(c1, i1, ku, B);
                CTM_TRY(xgemm(ctx, (int)A, (int)B, (int)ku, xm(x, ku, false), xm(y, B, false), U.p, U.q, B));
                const long long cu = i0 * pp[1] * pp[1] + i1;
                XM u{U.p, U.q, AB, false, false}, l{Lall.p, Lall.q, AB, true, false};
                CTM_TRY(xgemm(ctx, 1, (int)Pl, (int)AB, u, l, R.p + cu * Pl, R.q ? R.q + cu * Pl : nullptr, Pl));
            }
    }
    if (partial) return io.finish();
    // R[s0 t0 s1 t1 ; s2 t2 s3 t3] -> rdm[s0 s1 s2 s3 ; t0 t1 t2 t3]                                               (:1581-1588)
    long long dims[8] = {pp[0], pp[0], pp[1], pp[1], pp[3], pp[3], pp[2], pp[2]};
    int perm[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    CTM_TRY(permute_f64(ctx, R.p, r.p, 8, dims, perm));
    if (R.q) CTM_TRY(permute_f64(ctx, R.q, r.q, 8, dims, perm));
    return io.finish();
    });
}

int ctm_rdm1x1(ctm_ctx* ctx, const double* const* t, int chi, const int* ad, double* out) {
    return ctm_entry(ctx, "ctm_rdm1x1", [&]() -> int {
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    IO io(ctx);
    const long long X = chi;
    DT C1, C2, C3, C4, T1, T2, T3, T4, A, r;
    CTM_TRY(io.in(t[0], {X, X}, &C1)); CTM_TRY(io.in(t[1], {X, X}, &C2)); CTM_TRY(io.in(t[2], {X, X}, &C3)); CTM_TRY(io.in(t[3], {X, X}, &C4));
    CTM_TRY(io.in(t[4], {X, ad[1], ad[1], X}, &T1)); CTM_TRY(io.in(t[5], {X, ad[4], ad[4], X}, &T2));
    CTM_TRY(io.in(t[6], {ad[3], ad[3], X, X}, &T3)); CTM_TRY(io.in(t[7], {X, X, ad[2], ad[2]}, &T4));
    CTM_TRY(io.in(t[8], {ad[0], ad[1], ad[2], ad[3], ad[4]}, &A));
    CTM_TRY(io.out(out, (size_t)ad[0] * ad[0], &r));
    // left column, then site ket/bra, then right column (every pairwise step is a plain GEMM)
    CTM_TRY(dev_seq_einsum(ctx, "ab,bUVc,aiLM,ih,XYhg,sULXR,tVMYQ,ce,eRQf,fg->st", {C1, T1, T4, C4, T3, A, A.conj(), C2, T2, C3}, &r));
    return io.finish();
    });
}

int ctm_rdm2x1(ctm_ctx* ctx, const double* const* t, int chi, const int* ad2, double* out) {
    return ctm_entry(ctx, "ctm_rdm2x1", [&]() -> int {
    // tensors12: C1,T1a,T4,C4,T3a,a0 (site 0), C2,T2,C3,T1b,T3b,a1 (site 1 = coord+(1,0))
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    IO io(ctx);
    const long long X = chi; const int* a0 = ad2; const int* a1 = ad2 + 5;
    DT C1, T1a, T4, C4, T3a, A0, C2, T2, C3, T1b, T3b, A1;
    CTM_TRY(io.in(t[0], {X, X}, &C1)); CTM_TRY(io.in(t[1], {X, a0[1], a0[1], X}, &T1a)); CTM_TRY(io.in(t[2], {X, X, a0[2], a0[2]}, &T4));
    CTM_TRY(io.in(t[3], {X, X}, &C4)); CTM_TRY(io.in(t[4], {a0[3], a0[3], X, X}, &T3a));
    CTM_TRY(io.in(t[5], {a0[0], a0[1], a0[2], a0[3], a0[4]}, &A0));
    CTM_TRY(io.in(t[6], {X, X}, &C2)); CTM_TRY(io.in(t[7], {X, a1[4], a1[4], X}, &T2)); CTM_TRY(io.in(t[8], {X, X}, &C3));
    CTM_TRY(io.in(t[9], {X, a1[1], a1[1], X}, &T1b)); CTM_TRY(io.in(t[10], {a1[3], a1[3], X, X}, &T3b));
    CTM_TRY(io.in(t[11], {a1[0], a1[1], a1[2], a1[3], a1[4]}, &A1));
    DT left, right, r;
    CTM_TRY(io.out(out, (size_t)a0[0] * a0[0] * a1[0] * a1[0], &r));
    CTM_TRY(dev_seq_einsum(ctx, "ab,bUVc,aiLM,ih,XYhg,sULXR,tVMYQ->cRQgst", {C1, T1a, T4, C4, T3a, A0, A0.conj()}, &left));
    CTM_TRY(dev_seq_einsum(ctx, "ce,eRQf,fg,jUVc,XYhg,sULXR,tVMYQ->jLMhst", {C2, T2, C3, T1b, T3b, A1, A1.conj()}, &right));
    CTM_TRY(dev_einsum2(ctx, "cRQgst", left, "cRQguv", right, "sutv", &r));
    return io.finish();
    });
}

int ctm_rdm1x2(ctm_ctx* ctx, const double* const* t, int chi, const int* ad2, double* out) {
    return ctm_entry(ctx, "ctm_rdm1x2", [&]() -> int {
    // tensors12: C1,T1,C2,T4a,T2a,a0 (site 0), C4,T3,C3,T4b,T2b,a1 (site 1 = coord+(0,1))
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    IO io(ctx);
    const long long X = chi; const int* a0 = ad2; const int* a1 = ad2 + 5;
    DT C1, T1, C2, T4a, T2a, A0, C4, T3, C3, T4b, T2b, A1;
    CTM_TRY(io.in(t[0], {X, X}, &C1)); CTM_TRY(io.in(t[1], {X, a0[1], a0[1], X}, &T1)); CTM_TRY(io.in(t[2], {X, X}, &C2));
    CTM_TRY(io.in(t[3], {X, X, a0[2], a0[2]}, &T4a)); CTM_TRY(io.in(t[4], {X, a0[4], a0[4], X}, &T2a));
    CTM_TRY(io.in(t[5], {a0[0], a0[1], a0[2], a0[3], a0[4]}, &A0));
    CTM_TRY(io.in(t[6], {X, X}, &C4)); CTM_TRY(io.in(t[7], {a1[3], a1[3], X, X}, &T3)); CTM_TRY(io.in(t[8], {X, X}, &C3));
    CTM_TRY(io.in(t[9], {X, X, a1[2], a1[2]}, &T4b)); CTM_TRY(io.in(t[10], {X, a1[4], a1[4], X}, &T2b));
    CTM_TRY(io.in(t[11], {a1[0], a1[1], a1[2], a1[3], a1[4]}, &A1));
    DT up, lo, r;
    CTM_TRY(io.out(out, (size_t)a0[0] * a0[0] * a1[0] * a1[0], &r));
    CTM_TRY(dev_seq_einsum(ctx, "ab,bUVc,ce,aiLM,eRQf,sULXR,tVMYQ->iXYfst", {C1, T1, C2, T4a, T2a, A0, A0.conj()}, &up));
    CTM_TRY(dev_seq_einsum(ctx, "jh,XYhg,fg,ijLM,eRQf,sULXR,tVMYQ->iUVest", {C4, T3, C3, T4b, T2b, A1, A1.conj()}, &lo));
    CTM_TRY(dev_einsum2(ctx, "iXYfst", up, "iXYfuv", lo, "sutv", &r));
    return io.finish();
    });
}

int ctm_rdm_c4v(ctm_ctx* ctx, int which, const double* a, const double* C, const double* T, int chi, int p, int D, double* out) {
    return ctm_entry(ctx, "ctm_rdm_c4v", [&]() -> int {
    if (which < 0 || which > 3) { ctx->set_error("rdm_c4v: bad selector"); return CTM_ERR_BADARG; }
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    IO io(ctx);
    const long long n = (long long)chi * D * D, X = chi, D2 = (long long)D * D, P = p;
    DT tA, tC, tT, c, r;
    CTM_TRY(io.in(a, {P, D, D, D, D}, &tA));
    CTM_TRY(io.in(C, {X, X}, &tC));
    CTM_TRY(io.in(T, {X, X, D, D}, &tT));
    CTM_TRY(io.out(out, (size_t)(which == 3 ? P * P * P * P * P * P * P * P : P * P * P * P), &r));
    CTM_TRY(alloc_dt(ctx, {n, n, P, P}, &c));
    {   // open enlarged corner (rdm_c4v.py:13-93)
        ArenaScope s2(ctx);
        CTM_TRY(dev_network(ctx, "xy,cyuU,xelL,suldr,tULDR->edDcrRst", {tC, tT, tT, tA, tA.conj()}, &c));
    }
    auto perm = [&](const DT& src, const DT& dst, int nd, const long long* dims, const int* pm) -> int {
        CTM_TRY(permute_f64(ctx, src.p, dst.p, nd, dims, pm));
        if (src.q) CTM_TRY(permute_f64(ctx, src.q, dst.q, nd, dims, pm));
        return CTM_OK;
    };
    if (which == 0) {            // rdm2x1_sl (rdm_c4v.py:530-665)
        DT c6 = c.view({X, D2, X, D2, P, P}), tT3 = tT.view({X, X, D2});
        DT c2x1, left;
        CTM_TRY(dev_einsum2(ctx, "ab", tC, "bcd", tT3, "acd", &c2x1));
        CTM_TRY(dev_einsum2(ctx, "acd", c2x1, "adefst", c6, "cefst", &left));
        CTM_TRY(dev_einsum2(ctx, "cefst", left, "ecfuv", left, "sutv", &r));
        return io.finish();
    }
    if (which == 1 || which == 2) {
        DT cc;
        CTM_TRY(alloc_dt(ctx, {n, n}, &cc));
        CTM_TRY(trace_partial(ctx, c.p, cc.p, n * n, p));                     // C2x2c = einsum('abii->ab')
        if (c.q) CTM_TRY(trace_partial(ctx, c.q, cc.q, n * n, p));
        DT c3 = c.view({n, n, P * P});
        DT r3;
        if (which == 1) {        // _rdm2x2_NN_lowmem (rdm_c4v.py:1204-1284)
            DT r1, r2;
            CTM_TRY(dev_einsum2(ctx, "ab", cc, "bcs", c3, "acs", &r1));
            CTM_TRY(dev_einsum2(ctx, "da", cc, "acs", r1, "dcs", &r2));
            CTM_TRY(dev_einsum2(ctx, "cdt", c3, "dcs", r2, "ts", &r3));        // [(k0 b0), (k1 b1)]
        } else {                 // _rdm2x2_NNN_lowmem (rdm_c4v.py:1373-1443)
            DT h;
            CTM_TRY(dev_einsum2(ctx, "ab", cc, "bcs", c3, "acs", &h));
            CTM_TRY(dev_einsum2(ctx, "acs", h, "cat", h, "st", &r3));
        }
        long long dims[4] = {p, p, p, p}; int pm[4] = {0, 2, 1, 3};
        CTM_TRY(perm(r3, r, 4, dims, pm));
        return io.finish();
    }
    // rdm2x2 (rdm_c4v.py:1446-1545)
    DT c4 = c.view({n, n, P, P});
    DT up, r8;
    CTM_TRY(dev_einsum2(ctx, "akst", c4, "kbuv", c4, "abstuv", &up));
    CTM_TRY(dev_einsum2(ctx, "abstuv", up, "bawxyz", up, "stuvwxyz", &r8));
    long long dims[8]; for (int i = 0; i < 8; ++i) dims[i] = p;
    int pm[8] = {0, 2, 6, 4, 1, 3, 7, 5};
    CTM_TRY(perm(r8, r, 8, dims, pm));
    return io.finish();
    });
}

int ctm_init_piece(ctm_ctx* ctx, int kind, const double* a, const int* ad, double* out) {
    return ctm_entry(ctx, "ctm_init_piece", [&]() -> int {
    // env.py:367-536: 'mijef,mijab->eafb' etc.  A <- conj(A) pairs; implemented as one GEMM per piece:
    // permute the site so that the traced legs (+ physical) lead, then out[(kept ket),(kept bra)] = X^T X,
    // finally interleave ket/bra legs.
    static const int kept[8][3] = {{3, 4, -1}, {2, 3, -1}, {1, 2, -1}, {1, 4, -1}, {2, 3, 4}, {1, 3, 4}, {1, 2, 4}, {1, 2, 3}};
    if (kind < 0 || kind > 7) { ctx->set_error("init_piece: kind"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    const int nk = kind < 4 ? 2 : 3;
    int perm[5], np_ = 0; bool is_kept[5] = {false, false, false, false, false};
    for (int i = 0; i < nk; ++i) is_kept[kept[kind][i]] = true;
    long long Ktr = 1, Mk = 1, dims[5];
    for (int i = 0; i < 5; ++i) dims[i] = ad[i];
    for (int i = 0; i < 5; ++i) if (!is_kept[i]) { perm[np_++] = i; Ktr *= ad[i]; }
    for (int i = 0; i < nk; ++i) { perm[np_++] = kept[kind][i]; Mk *= ad[kept[kind][i]]; }
    IO io(ctx);
    DT A, X, G, O;
    CTM_TRY(io.in(a, {dims[0], dims[1], dims[2], dims[3], dims[4]}, &A));
    CTM_TRY(alloc_dt(ctx, {Ktr, Mk}, &X));
    CTM_TRY(alloc_dt(ctx, {Mk, Mk}, &G));
    CTM_TRY(io.out(out, (size_t)(Mk * Mk), &O));
    CTM_TRY(permute_f64(ctx, A.p, X.p, 5, dims, perm));
    if (A.q) CTM_TRY(permute_f64(ctx, A.q, X.q, 5, dims, perm));
    // G[(kept ket),(kept bra)] = sum_traced X[.,ket] conj(X[.,bra])
    CTM_TRY(xgemm(ctx, (int)Mk, (int)Mk, (int)Ktr, xm(X, Mk, true), xm(X, Mk, false, true), G.p, G.q, Mk));
    // G[(e f [g]), (a b [c])] -> out[e a f b [g c]]
    long long gd[6]; int gp[6];
    for (int i = 0; i < nk; ++i) { gd[i] = ad[kept[kind][i]]; gd[nk + i] = ad[kept[kind][i]]; gp[2 * i] = i; gp[2 * i + 1] = nk + i; }
    CTM_TRY(permute_f64(ctx, G.p, O.p, 2 * nk, gd, gp));
    if (G.q) CTM_TRY(permute_f64(ctx, G.q, O.q, 2 * nk, gd, gp));
    CTM_TRY(normalize_dt(ctx, O.view({Mk * Mk})));
    return io.finish();
    });
}

}  // extern "C"
