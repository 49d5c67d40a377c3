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
__global__ void fix_signs_rows_kernel(double* Ut, double* Vt, int k, int n) {
    const int r = blockIdx.x;
    if (r >= k) return;
    double* u = Ut + (size_t)r * n;
    double* v = Vt + (size_t)r * n;
    long long best = -1; int bi = 0;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const long long a = (long long)(fabs(u[c]) * 1099511627776.0);   // 2^40
        if (a > best) { best = a; bi = c; }                                 // strided scan keeps the first max per thread
    }
    __shared__ long long sb[256]; __shared__ int si[256];
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
    const double sg = (ph < 0.0) ? -1.0 : 1.0;
    __syncthreads();
    if (sg < 0.0) {
        for (int c = threadIdx.x; c < n; c += blockDim.x) { u[c] = -u[c]; v[c] = -v[c]; }
    }
}

// out (n x chi, row-major) = transpose of rows[0:chi] (k x n) with columns > keep zeroed
__global__ void rows_to_cols_kernel(const double* rows, int n, int chi, int keep_last, double* out) {
    const size_t tot = (size_t)n * chi;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const size_t i = q / chi, j = q - i * chi;
        out[q] = ((int)j <= keep_last) ? rows[j * n + i] : 0.0;
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

struct TruncOut { std::vector<double> S; int keep_last; int k; };

// leading triplets of M (n x n) as ROW factors Ut, Vt (k x n), S on host, multiplet-aware keep index
int svd_rows_op(ctm_ctx* ctx, const MatOp& op, int chi, const ctm_trunc_cfg& cfg, double* Ut, double* Vt, double* dS, TruncOut* to) {
    const int n = op.n;
    const int k = (chi < n) ? chi + 1 : n;
    CTM_TRY(jacobi_svd_top_op(ctx, op, k, dS, Ut, Vt));
    to->S.resize(k); to->k = k;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(to->S.data(), dS, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (cfg.fix_signs) hipLaunchKernelGGL(fix_signs_rows_kernel, dim3(std::min(k, chi)), dim3(256), 0, ctx->stream, Ut, Vt, std::min(k, chi), n);
    const int kc = std::min(chi, n);
    to->keep_last = kc - 1;
    if (cfg.keep_multiplets && chi < n) to->keep_last = std::min(kc - 1, multiplet_chi(to->S, chi, cfg.eps_multiplet, cfg.multiplet_abstol));
    return CTM_OK;
}

int svd_rows(ctm_ctx* ctx, const double* M, int n, int chi, const ctm_trunc_cfg& cfg, double* Ut, double* Vt, double* dS,
             TruncOut* to) {
    MatOp op; op.n = n; op.M = M;
    return svd_rows_op(ctx, op, chi, cfg, Ut, Vt, dS, to);
}

// S_sqrt = rsqrt(S) where S/S[0] > reltol (ctm_projectors.py:266-270), zero beyond the kept multiplets
void proj_scale(const TruncOut& to, int kc, double reltol, std::vector<double>* Sh, std::vector<double>* sc) {
    Sh->assign(kc, 0.0); sc->assign(kc, 0.0);
    for (int i = 0; i < kc; ++i) (*Sh)[i] = (i <= to.keep_last) ? to.S[i] : 0.0;
    int nz = 0;
    for (int i = 0; i < kc; ++i) if ((*Sh)[0] > 0.0 && (*Sh)[i] / (*Sh)[0] > reltol) { (*sc)[nz] = 1.0 / std::sqrt((*Sh)[i]); ++nz; }
}

// out (n x kc) = opA(cA) * ( opB(cB) * rows^T ) * diag(scale)   with rows = kc x n row factors
int corner_chain_times_rowsT(ctm_ctx* ctx, int n, int kc, const double* cA, bool tA, const double* cB, bool tB, const double* rows,
                             const double* d_scale, double* out) {
    ArenaScope scope(ctx);
    double* t1;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * kc, (void**)&t1));
    GemmDesc g; g.M = n; g.N = kc; g.K = n;
    g.A = cB; if (tB) { g.sam = 1; g.sak = n; } else { g.sam = n; g.sak = 1; }
    g.B = rows; g.sbk = 1; g.sbn = n; g.C = t1; g.ldc = kc;
    CTM_TRY(gemm_f64(ctx, g));
    GemmDesc h; h.M = n; h.N = kc; h.K = n;
    h.A = cA; if (tA) { h.sam = 1; h.sak = n; } else { h.sam = n; h.sak = 1; }
    h.B = t1; h.sbk = kc; h.sbn = 1; h.C = out; h.ldc = kc; h.colscale = d_scale;
    return gemm_f64(ctx, h);
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

int corner_impl(ctm_ctx* ctx, int corner, int open, const double* C, const double* T1, const double* T2, const double* a,
                int chi, const int* ad, double* out) {
    if (corner < 0 || corner > 3) { ctx->set_error("c2x2: bad corner"); return CTM_ERR_BADARG; }
    const CornerSpec& sp = kCorner[corner];
    DT tC(C, {chi, chi});
    DT tT1(T1, t_view(sp.t1_axis, chi, ad[sp.t1_leg]));
    DT tT2(T2, t_view(sp.t2_axis, chi, ad[sp.t2_leg]));
    DT tA(a, {ad[0], ad[1], ad[2], ad[3], ad[4]});
    DT res; res.p = out;
    return dev_network(ctx, open ? sp.open : sp.closed, {tC, tT1, tT2, tA, tA}, &res);
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

int ctm_c2x2(ctm_ctx* ctx, int corner, int open, const double* C, const double* T1, const double* T2, const double* a,
             int chi, const int* adims, double* out) {
    PhaseTimer pt(ctx, CTM_T_CORNERS);
    ArenaScope scope(ctx);
    return corner_impl(ctx, corner, open, C, T1, T2, a, chi, adims, out);
}

int ctm_halves(ctm_ctx* ctx, int dir, const double* const* t, int chi, const int* adims4x5, double* R, double* Rt) {
    if (dir < 0 || dir > 3) { ctx->set_error("halves: bad direction"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    double* outs[2] = {R, Rt};
    for (int h = 0; h < 2; ++h) {
        const HalfSpec& hs = kHalves[dir][h];
        const int ia = 2 * h, ib = 2 * h + 1;       // corner slots in tensors16: (A of R, B of R, A of Rt, B of Rt)
        long long a0, a1, b0, b1;
        corner_dims(hs.cA, chi, adims4x5 + 5 * ia, &a0, &a1);
        corner_dims(hs.cB, chi, adims4x5 + 5 * ib, &b0, &b1);
        ArenaScope inner(ctx);
        double *cA, *cB;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(a0 * a1), (void**)&cA));
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(b0 * b1), (void**)&cB));
        {
            PhaseTimer pt(ctx, CTM_T_CORNERS);
            { ArenaScope s2(ctx); CTM_TRY(corner_impl(ctx, hs.cA, 0, t[4 * ia], t[4 * ia + 1], t[4 * ia + 2], t[4 * ia + 3], chi, adims4x5 + 5 * ia, cA)); }
            { ArenaScope s2(ctx); CTM_TRY(corner_impl(ctx, hs.cB, 0, t[4 * ib], t[4 * ib + 1], t[4 * ib + 2], t[4 * ib + 3], chi, adims4x5 + 5 * ib, cB)); }
        }
        PhaseTimer pt(ctx, CTM_T_HALVES);
        const long long M = hs.tA ? a1 : a0, Ka = hs.tA ? a0 : a1;
        const long long N = hs.tB ? b0 : b1, Kb = hs.tB ? b1 : b0;
        if (Ka != Kb) { ctx->set_error("halves: contracted dims differ"); return CTM_ERR_SHAPE; }
        GemmDesc g;
        g.M = (int)M; g.N = (int)N; g.K = (int)Ka;
        g.A = cA; if (hs.tA) { g.sam = 1; g.sak = a1; } else { g.sam = a1; g.sak = 1; }
        g.B = cB; if (hs.tB) { g.sbk = 1; g.sbn = b1; } else { g.sbk = b1; g.sbn = 1; }
        g.C = outs[h]; g.ldc = N;
        CTM_TRY(gemm_f64(ctx, g));
    }
    return CTM_OK;
}

int ctm_truncated_svd(ctm_ctx* ctx, const double* M, int n, int chi, const ctm_trunc_cfg* cfg_, double* U, double* S, double* V) {
    const ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (chi < 1 || n < 1) { ctx->set_error("truncated_svd: bad dims"); return CTM_ERR_BADARG; }
    PhaseTimer pt(ctx, CTM_T_SVD);
    ArenaScope scope(ctx);
    const int k = (chi < n) ? chi + 1 : n, kc = std::min(chi, n);
    double *Ut, *Vt, *dS;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Ut));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Vt));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&dS));
    TruncOut to;
    CTM_TRY(svd_rows(ctx, M, n, chi, cfg, Ut, Vt, dS, &to));
    std::vector<double> Sh(kc);
    for (int i = 0; i < kc; ++i) Sh[i] = (i <= to.keep_last) ? to.S[i] : 0.0;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(S, Sh.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(rows_to_cols_kernel, dim3(1024), dim3(256), 0, ctx->stream, Ut, n, kc, to.keep_last, U);
    hipLaunchKernelGGL(rows_to_cols_kernel, dim3(1024), dim3(256), 0, ctx->stream, Vt, n, kc, to.keep_last, V);
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
}

int ctm_truncated_eigh(ctm_ctx* ctx, const double* A, int n, int chi, const ctm_trunc_cfg* cfg_, double* D, double* U) {
    ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (!cfg_) { cfg.eps_multiplet = 1.0e-12; }
    if (chi < 1 || n < 1) { ctx->set_error("truncated_eigh: bad dims"); return CTM_ERR_BADARG; }
    PhaseTimer pt(ctx, CTM_T_EIG);
    ArenaScope scope(ctx);
    const int k = (chi < n) ? chi + 1 : n, kc = std::min(chi, n);
    double *Ut, *dD;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Ut));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&dD));
    CTM_TRY(jacobi_eigh_top(ctx, A, n, k, dD, Ut));
    std::vector<double> Dh(k);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(Dh.data(), dD, sizeof(double) * k, hipMemcpyDeviceToHost, ctx->stream));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    int keep_last = kc - 1;
    if (cfg.keep_multiplets && chi < n) keep_last = std::min(kc - 1, multiplet_chi(Dh, chi, cfg.eps_multiplet, cfg.multiplet_abstol));
    std::vector<double> Do(kc);
    for (int i = 0; i < kc; ++i) Do[i] = (i <= keep_last) ? Dh[i] : 0.0;
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(D, Do.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(rows_to_cols_kernel, dim3(1024), dim3(256), 0, ctx->stream, Ut, n, kc, keep_last, U);
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
}

int ctm_svdvals(ctm_ctx* ctx, const double* M, int n, double* S) {
    ArenaScope scope(ctx);
    return jacobi_svdvals(ctx, M, n, S);
}

int ctm_projectors(ctm_ctx* ctx, const double* R, const double* Rt, int n, int chi, const ctm_trunc_cfg* cfg_, double* P,
                   double* Pt, double* S_out) {
    const ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (chi < 1 || n < 1) { ctx->set_error("projectors: bad dims"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    const int k = (chi < n) ? chi + 1 : n, kc = std::min(chi, n);
    double *M, *Ut, *Vt, *dS, *dScale;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * n, (void**)&M));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Ut));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Vt));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&dS));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * kc, (void**)&dScale));
    {   // M = R^T Rt  (ctm_projectors.py:263, plain transpose)
        PhaseTimer pt(ctx, CTM_T_HALVES);
        GemmDesc g; g.M = n; g.N = n; g.K = n; g.A = R; g.sam = 1; g.sak = n; g.B = Rt; g.sbk = n; g.sbn = 1; g.C = M; g.ldc = n;
        CTM_TRY(gemm_f64(ctx, g));
    }
    TruncOut to;
    { PhaseTimer pt(ctx, CTM_T_SVD); CTM_TRY(svd_rows(ctx, M, n, chi, cfg, Ut, Vt, dS, &to)); }
    PhaseTimer pt(ctx, CTM_T_PROJ);
    // S_sqrt = rsqrt(S) where S/S[0] > reltol (ctm_projectors.py:266-270), zero beyond the kept multiplets
    std::vector<double> Sh(kc), sc(kc, 0.0);
    for (int i = 0; i < kc; ++i) Sh[i] = (i <= to.keep_last) ? to.S[i] : 0.0;
    int nz = 0;
    for (int i = 0; i < kc; ++i) if (Sh[0] > 0.0 && Sh[i] / Sh[0] > cfg.svd_reltol) { sc[nz] = 1.0 / std::sqrt(Sh[i]); ++nz; }
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(dScale, sc.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    if (S_out) CTM_HIP_CHECK(ctx, hipMemcpyAsync(S_out, Sh.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    // P = R conj(U) diag(S_sqrt),  Pt = Rt V diag(S_sqrt)   (:283); U,V as row factors -> NT GEMM + fused column scale
    GemmDesc g; g.M = n; g.N = kc; g.K = n; g.A = R; g.sam = n; g.sak = 1; g.B = Ut; g.sbk = 1; g.sbn = n; g.C = P; g.ldc = kc; g.colscale = dScale;
    CTM_TRY(gemm_f64(ctx, g));
    g.A = Rt; g.B = Vt; g.C = Pt;
    CTM_TRY(gemm_f64(ctx, g));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
}

int ctm_projectors_4x4(ctm_ctx* ctx, int dir, const double* const* t, int chi, const int* adims4x5, const ctm_trunc_cfg* cfg_,
                       double* P, double* Pt, double* S_out) {
    const ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (dir < 0 || dir > 3) { ctx->set_error("projectors_4x4: bad direction"); return CTM_ERR_BADARG; }
    ArenaScope scope(ctx);
    // the four enlarged corners of the move (reference order: A,B of R then A,B of Rt)
    double* c[4]; long long d0[4], d1[4]; int cid[4]; bool tr[4];
    for (int h = 0; h < 2; ++h) {
        const HalfSpec& hs = kHalves[dir][h];
        cid[2 * h] = hs.cA; cid[2 * h + 1] = hs.cB; tr[2 * h] = hs.tA != 0; tr[2 * h + 1] = hs.tB != 0;
    }
    {
        PhaseTimer pt(ctx, CTM_T_CORNERS);
        for (int i = 0; i < 4; ++i) {
            corner_dims(cid[i], chi, adims4x5 + 5 * i, &d0[i], &d1[i]);
            CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(d0[i] * d1[i]), (void**)&c[i]));
            ArenaScope s2(ctx);
            CTM_TRY(corner_impl(ctx, cid[i], 0, t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3], chi, adims4x5 + 5 * i, c[i]));
        }
    }
    const long long n = d0[0];
    for (int i = 0; i < 4; ++i) if (d0[i] != n || d1[i] != n) { ctx->set_error("projectors_4x4: non-uniform bond dimensions are not supported on the fused path"); return CTM_ERR_UNSUPPORTED; }
    const int k = (chi < n) ? chi + 1 : (int)n, kc = std::min(chi, (int)n);
    double *Ut, *Vt, *dS, *dScale;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Ut));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)k * n, (void**)&Vt));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * k, (void**)&dS));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * kc, (void**)&dScale));
    MatOp op; op.n = (int)n;
    for (int i = 0; i < 4; ++i) { op.c[i] = c[i]; op.t[i] = tr[i]; }
    TruncOut to;
    { PhaseTimer pt(ctx, CTM_T_SVD); CTM_TRY(svd_rows_op(ctx, op, chi, cfg, Ut, Vt, dS, &to)); }
    PhaseTimer pt(ctx, CTM_T_PROJ);
    std::vector<double> Sh, sc;
    proj_scale(to, kc, cfg.svd_reltol, &Sh, &sc);
    CTM_HIP_CHECK(ctx, hipMemcpyAsync(dScale, sc.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    if (S_out) CTM_HIP_CHECK(ctx, hipMemcpyAsync(S_out, Sh.data(), sizeof(double) * kc, hipMemcpyHostToDevice, ctx->stream));
    // P = R conj(U) S^-1/2 = opA(cA) opB(cB) U ... ; Pt = Rt V S^-1/2 = opC(cC) opD(cD) V ...
    CTM_TRY(corner_chain_times_rowsT(ctx, (int)n, kc, c[0], tr[0], c[1], tr[1], Ut, dScale, P));
    CTM_TRY(corner_chain_times_rowsT(ctx, (int)n, kc, c[2], tr[2], c[3], tr[3], Vt, dScale, Pt));
    CTM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CTM_OK;
}

int ctm_absorb(ctm_ctx* ctx, int dir, const double* const* t, int chi, const int* ad, int normalize, double* nC1,
               double* nC2, double* nT) {
    if (dir < 0 || dir > 3) { ctx->set_error("absorb: bad direction"); return CTM_ERR_BADARG; }
    const AbsorbSpec& sp = kAbsorb[dir];
    const double *C1 = t[0], *T1 = t[1], *T = t[2], *T2 = t[3], *C2 = t[4], *A = t[5], *P2 = t[6], *Pt2 = t[7], *P1 = t[8], *Pt1 = t[9];
    // D^2 extents of the T tensors follow the site legs they attach to (uniform D assumed per leg pair)
    const long long Dt = ad[sp.t_leg], Dpt2 = ad[sp.pt2_leg], Dp1 = ad[sp.p1_leg];
    // T1 carries the D^2 leg shared with Pt1, T2 the one shared with P2
    const long long Dt1 = Dp1 /* neighbour projector leg == this site's leg on that side */, Dt2 = Dpt2;
    auto t3 = [&](const double* p, int axis, long long D2) {
        std::vector<long long> d; for (int a = 0; a < 3; ++a) d.push_back(a == axis ? D2 : (long long)chi); return DT(p, d);
    };
    {
        PhaseTimer pt(ctx, CTM_T_ABSORB);
        ArenaScope scope(ctx);
        DT tC1(C1, {chi, chi}), tC2(C2, {chi, chi});
        DT tT1 = t3(T1, sp.t1_axis, Dt1 * Dt1), tT2 = t3(T2, sp.t2_axis, Dt2 * Dt2);
        DT tPt1(Pt1, {chi, Dt1 * Dt1, chi}), tP2(P2, {chi, Dt2 * Dt2, chi});
        DT r1; r1.p = nC1; CTM_TRY(dev_seq_einsum(ctx, sp.nC1, {tPt1, tC1, tT1}, &r1));
        DT r2; r2.p = nC2; CTM_TRY(dev_seq_einsum(ctx, sp.nC2, {tC2, tT2, tP2}, &r2));
        DT tT(T, t_view(sp.t_axis, chi, Dt));
        DT tPt2(Pt2, {chi, Dpt2, Dpt2, chi}), tP1(P1, {chi, Dp1, Dp1, chi});
        DT tA(A, {ad[0], ad[1], ad[2], ad[3], ad[4]});
        DT r3; r3.p = nT; CTM_TRY(dev_network(ctx, sp.nT, {tT, tPt2, tA, tA, tP1}, &r3));
        (void)sp.fuse0;   // the fused output axes are adjacent: the 4-index result IS the 3-index tensor in memory
    }
    if (normalize) {
        PhaseTimer pt(ctx, CTM_T_NORM);
        long long D2out = 0;
        {   // nT has one D^2 leg: the site leg opposite to the absorbed T (UP:d, LEFT:r, DOWN:u, RIGHT:l)
            static const int out_leg[4] = {3, 4, 1, 2};
            D2out = (long long)ad[out_leg[dir]] * ad[out_leg[dir]];
        }
        CTM_TRY(ctm_normalize_inf(ctx, nC1, (long long)chi * chi));
        CTM_TRY(ctm_normalize_inf(ctx, nC2, (long long)chi * chi));
        CTM_TRY(ctm_normalize_inf(ctx, nT, (long long)chi * chi * D2out));
    }
    return CTM_OK;
}

// ---- C4v -------------------------------------------------------------------------------------------
int ctm_c2x2_c4v(ctm_ctx* ctx, int open, const double* a, const double* C, const double* T, int chi, int p, int D, double* out) {
    PhaseTimer pt(ctx, CTM_T_CORNERS);
    ArenaScope scope(ctx);
    DT tC(C, {chi, chi}), tT(T, {chi, chi, D, D}), tA(a, {p, D, D, D, D});
    DT res; res.p = out;
    return dev_network(ctx, open ? "xy,cyuU,xelL,suldr,tULDR->edDcrRst" : "xy,cyuU,xelL,suldr,sULDR->edDcrR",
                       {tC, tT, tT, tA, tA}, &res);
}

int ctm_move_c4v(ctm_ctx* ctx, const double* a, const double* C, const double* T, int chi, int p, int D,
                 const ctm_trunc_cfg* cfg_, double* C_out, double* T_out, double* D_out) {
    ctm_trunc_cfg cfg = cfg_ ? *cfg_ : kDefaultCfg;
    if (!cfg_) cfg.eps_multiplet = 1.0e-12;          // custom_eig.py default used by ctmrg_c4v.py:49-52
    const int n = chi * D * D;
    ArenaScope scope(ctx);
    double *C2X2, *Dv, *P;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * n, (void**)&C2X2));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * chi, (void**)&Dv));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)n * chi, (void**)&P));
    CTM_TRY(ctm_c2x2_c4v(ctx, 0, a, C, T, chi, p, D, C2X2));
    CTM_TRY(ctm_truncated_eigh(ctx, C2X2, n, chi, &cfg, Dv, P));
    PhaseTimer pt(ctx, CTM_T_ABSORB);
    CTM_TRY(diag_to_matrix(ctx, Dv, C_out, chi));                                     // ctmrg_c4v.py:374
    if (D_out) CTM_HIP_CHECK(ctx, hipMemcpyAsync(D_out, Dv, sizeof(double) * chi, hipMemcpyDeviceToDevice, ctx->stream));
    DT tP(P, {chi, D, D, chi}), tT(T, {chi, chi, D, D}), tA(a, {p, D, D, D, D});
    DT res; res.p = T_out;
    CTM_TRY(dev_network(ctx, "xuUi,xelL,suldr,sULDR,edDj->ijrR", {tP, tT, tA, tA, tP}, &res));   // :383-443
    CTM_TRY(add_transposed01(ctx, T_out, chi, D * D));                                 // :446
    // C /= |C[0,0]| ; T /= max|T|   (:182-197)
    CTM_TRY(div_by_device_scalar(ctx, C_out, (size_t)chi * chi, Dv, 1));
    CTM_TRY(ctm_normalize_inf(ctx, T_out, (long long)chi * chi * D * D));
    return CTM_OK;
}

// ---- RDMs ------------------------------------------------------------------------------------------
int ctm_rdm2x2(ctm_ctx* ctx, const double* const* t, int chi, const int* ad4, double* out) {
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    static const int cid[4] = {CTM_LU, CTM_RU, CTM_RD, CTM_LD};
    DT c[4];
    for (int i = 0; i < 4; ++i) {
        long long n0, n1; corner_dims(cid[i], chi, ad4 + 5 * i, &n0, &n1);
        const long long p = ad4[5 * i];
        double* buf; CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(n0 * n1 * p * p), (void**)&buf));
        { ArenaScope s2(ctx); CTM_TRY(corner_impl(ctx, cid[i], 1, t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3], chi, ad4 + 5 * i, buf)); }
        c[i] = DT(buf, {n0, n1, p, p});
    }
    DT up, lo, r; r.p = out;
    CTM_TRY(dev_einsum2(ctx, "akst", c[0], "kbuv", c[1], "abstuv", &up));          // rdm.py:1459-1460
    CTM_TRY(dev_einsum2(ctx, "akst", c[3], "bkuv", c[2], "abstuv", &lo));          // :1527-1528
    // rdm[s0 s1 s2 s3 ; t0 t1 t2 t3]                                                  // :1581-1588
    return dev_einsum2(ctx, "abstuv", up, "abwxyz", lo, "suwytvxz", &r);
}

int ctm_rdm1x1(ctm_ctx* ctx, const double* const* t, int chi, const int* ad, double* out) {
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    const long long X = chi;
    DT C1(t[0], {X, X}), C2(t[1], {X, X}), C3(t[2], {X, X}), C4(t[3], {X, X});
    DT T1(t[4], {X, ad[1], ad[1], X}), T2(t[5], {X, ad[4], ad[4], X}), T3(t[6], {ad[3], ad[3], X, X}), T4(t[7], {X, X, ad[2], ad[2]});
    DT A(t[8], {ad[0], ad[1], ad[2], ad[3], ad[4]});
    DT r; r.p = out;
    // left column, then site ket/bra, then right column (every pairwise step is a plain GEMM)
    return dev_seq_einsum(ctx, "ab,bUVc,aiLM,ih,XYhg,sULXR,tVMYQ,ce,eRQf,fg->st", {C1, T1, T4, C4, T3, A, A, C2, T2, C3}, &r);
}

int ctm_rdm2x1(ctm_ctx* ctx, const double* const* t, int chi, const int* ad2, double* out) {
    // tensors12: C1,T1a,T4,C4,T3a,a0 (site 0), C2,T2,C3,T1b,T3b,a1 (site 1 = coord+(1,0))
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    const long long X = chi; const int* a0 = ad2; const int* a1 = ad2 + 5;
    DT C1(t[0], {X, X}), T1a(t[1], {X, a0[1], a0[1], X}), T4(t[2], {X, X, a0[2], a0[2]}), C4(t[3], {X, X}), T3a(t[4], {a0[3], a0[3], X, X});
    DT A0(t[5], {a0[0], a0[1], a0[2], a0[3], a0[4]});
    DT C2(t[6], {X, X}), T2(t[7], {X, a1[4], a1[4], X}), C3(t[8], {X, X}), T1b(t[9], {X, a1[1], a1[1], X}), T3b(t[10], {a1[3], a1[3], X, X});
    DT A1(t[11], {a1[0], a1[1], a1[2], a1[3], a1[4]});
    DT left, right, r; r.p = out;
    CTM_TRY(dev_seq_einsum(ctx, "ab,bUVc,aiLM,ih,XYhg,sULXR,tVMYQ->cRQgst", {C1, T1a, T4, C4, T3a, A0, A0}, &left));
    CTM_TRY(dev_seq_einsum(ctx, "ce,eRQf,fg,jUVc,XYhg,sULXR,tVMYQ->jLMhst", {C2, T2, C3, T1b, T3b, A1, A1}, &right));
    return dev_einsum2(ctx, "cRQgst", left, "cRQguv", right, "sutv", &r);
}

int ctm_rdm1x2(ctm_ctx* ctx, const double* const* t, int chi, const int* ad2, double* out) {
    // tensors12: C1,T1,C2,T4a,T2a,a0 (site 0), C4,T3,C3,T4b,T2b,a1 (site 1 = coord+(0,1))
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    const long long X = chi; const int* a0 = ad2; const int* a1 = ad2 + 5;
    DT C1(t[0], {X, X}), T1(t[1], {X, a0[1], a0[1], X}), C2(t[2], {X, X}), T4a(t[3], {X, X, a0[2], a0[2]}), T2a(t[4], {X, a0[4], a0[4], X});
    DT A0(t[5], {a0[0], a0[1], a0[2], a0[3], a0[4]});
    DT C4(t[6], {X, X}), T3(t[7], {a1[3], a1[3], X, X}), C3(t[8], {X, X}), T4b(t[9], {X, X, a1[2], a1[2]}), T2b(t[10], {X, a1[4], a1[4], X});
    DT A1(t[11], {a1[0], a1[1], a1[2], a1[3], a1[4]});
    DT up, lo, r; r.p = out;
    CTM_TRY(dev_seq_einsum(ctx, "ab,bUVc,ce,aiLM,eRQf,sULXR,tVMYQ->iXYfst", {C1, T1, C2, T4a, T2a, A0, A0}, &up));
    CTM_TRY(dev_seq_einsum(ctx, "jh,XYhg,fg,ijLM,eRQf,sULXR,tVMYQ->iUVest", {C4, T3, C3, T4b, T2b, A1, A1}, &lo));
    return dev_einsum2(ctx, "iXYfst", up, "iXYfuv", lo, "sutv", &r);
}

int ctm_rdm_c4v(ctm_ctx* ctx, int which, const double* a, const double* C, const double* T, int chi, int p, int D, double* out) {
    PhaseTimer pt(ctx, CTM_T_RDM);
    ArenaScope scope(ctx);
    const long long n = (long long)chi * D * D, X = chi, D2 = (long long)D * D;
    double* c; CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(n * n * p * p), (void**)&c));
    { ArenaScope s2(ctx); CTM_TRY(ctm_c2x2_c4v(ctx, 1, a, C, T, chi, p, D, c)); }
    DT r; r.p = out;
    if (which == 0) {            // rdm2x1_sl (rdm_c4v.py:530-665)
        DT c6(c, {X, D2, X, D2, (long long)p, (long long)p}), tC(C, {X, X}), tT(T, {X, X, D2});
        DT c2x1, left;
        CTM_TRY(dev_einsum2(ctx, "ab", tC, "bcd", tT, "acd", &c2x1));
        CTM_TRY(dev_einsum2(ctx, "acd", c2x1, "adefst", c6, "cefst", &left));
        return dev_einsum2(ctx, "cefst", left, "ecfuv", left, "sutv", &r);
    }
    if (which == 1 || which == 2) {
        double* cc; CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(n * n), (void**)&cc));
        CTM_TRY(trace_partial(ctx, c, cc, n * n, p));                         // C2x2c = einsum('abii->ab')
        DT c3(c, {n, n, (long long)p * p}), tcc(cc, {n, n});
        DT r3;
        if (which == 1) {        // _rdm2x2_NN_lowmem (rdm_c4v.py:1204-1284)
            DT r1, r2;
            CTM_TRY(dev_einsum2(ctx, "ab", tcc, "bcs", c3, "acs", &r1));
            CTM_TRY(dev_einsum2(ctx, "da", tcc, "acs", r1, "dcs", &r2));
            CTM_TRY(dev_einsum2(ctx, "cdt", c3, "dcs", r2, "ts", &r3));        // [(k0 b0), (k1 b1)]
        } else {                 // _rdm2x2_NNN_lowmem (rdm_c4v.py:1373-1443)
            DT h;
            CTM_TRY(dev_einsum2(ctx, "ab", tcc, "bcs", c3, "acs", &h));
            CTM_TRY(dev_einsum2(ctx, "acs", h, "cat", h, "st", &r3));
        }
        long long dims[4] = {p, p, p, p}; int perm[4] = {0, 2, 1, 3};
        return permute_f64(ctx, r3.p, out, 4, dims, perm);
    }
    if (which == 3) {            // rdm2x2 (rdm_c4v.py:1446-1545)
        DT c4(c, {n, n, (long long)p, (long long)p});
        DT up, r8;
        CTM_TRY(dev_einsum2(ctx, "akst", c4, "kbuv", c4, "abstuv", &up));
        CTM_TRY(dev_einsum2(ctx, "abstuv", up, "bawxyz", up, "stuvwxyz", &r8));
        long long dims[8]; for (int i = 0; i < 8; ++i) dims[i] = p;
        int perm[8] = {0, 2, 6, 4, 1, 3, 7, 5};
        return permute_f64(ctx, r8.p, out, 8, dims, perm);
    }
    ctx->set_error("rdm_c4v: bad selector"); return CTM_ERR_BADARG;
}

int ctm_init_piece(ctm_ctx* ctx, int kind, const double* a, const int* ad, double* out) {
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
    double *X, *G;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(Ktr * Mk), (void**)&X));
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(Mk * Mk), (void**)&G));
    CTM_TRY(permute_f64(ctx, a, X, 5, dims, perm));
    GemmDesc g; g.M = (int)Mk; g.N = (int)Mk; g.K = (int)Ktr; g.A = X; g.sam = 1; g.sak = Mk; g.B = X; g.sbk = Mk; g.sbn = 1; g.C = G; g.ldc = Mk;
    CTM_TRY(gemm_f64(ctx, g));
    // G[(e f [g]), (a b [c])] -> out[e a f b [g c]]
    long long gd[6]; int gp[6];
    for (int i = 0; i < nk; ++i) { gd[i] = ad[kept[kind][i]]; gd[nk + i] = ad[kept[kind][i]]; gp[2 * i] = i; gp[2 * i + 1] = nk + i; }
    CTM_TRY(permute_f64(ctx, G, out, 2 * nk, gd, gp));
    return ctm_normalize_inf(ctx, out, Mk * Mk);
}

}  // extern "C"
