// Fused two-layer site absorption (SURVEY 2.3 K3+K4+K5): for every pair (x,y) of spectator (chi) indices
//
//   out[x,y,(e1k,e2k),(e1b,e2b)] = sum_{c1k,c2k,c1b,c2b,s}  Z[x,y,c1k,c1b,c2k,c2b] a[s,c1k,c2k,e1k,e2k] conj(a)[s,c1b,c2b,e1b,e2b]
//
// i.e. the ket layer and the bra layer of the on-site tensor are absorbed into a (chi,chi,D^2,D^2) block tensor in
// ONE pass: the reference does it as two tensordots on non-adjacent legs with a 6-index permute before, between and
// after (ctm_components.py:406-414, ctmrg.py:405-425, ctmrg_c4v.py:383-443) -- three n^2-sized round trips through
// HBM that are pure layout work.  Here Z is gathered once (arbitrary strides), both layers run on the FP64 matrix
// cores out of LDS, and the result is scattered once in the layout the NEXT contraction wants (arbitrary strides):
//   step A (per phys s):  W[kb][e] = sum_kk Zs[kb][kk] As[kk][s][e]     kb=(c1b,c2b) kk=(c1k,c2k) e=(e1k,e2k)
//   step B (acc over s):  O[e][E] += sum_kb W[kb][e] As[kb][s][E]       E=(e1b,e2b)
// The site tensor (p D^4 <= 8192 doubles) lives in LDS for the whole kernel in ONE image [c][s][e] that serves both
// layers (real dtype: conj(a) = a).  One workgroup = 4 waves; each (x,y) costs 2 p (KAp/16)(NEp/16)(KAp/4) MFMAs.
#include "contract.h"
#include <mutex>

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

struct Layer2Params {
    const double* Z; long long zs_x, zs_y, zs_c1k, zs_c1b, zs_c2k, zs_c2b;
    int nx, ny;
    const double* A;                 // [c1][c2][s][e1][e2] contiguous
    int D1, D2, E1, E2, p;
    double* out; long long os_x, os_y, os_e1k, os_e1b, os_e2k, os_e2b;
    int KA, KAp, NE, NEp;
    int ldz, lda, ldw;               // LDS row strides (doubles)
    int dbg;                         // development: 1 skip gather, 2 skip MFMA, 4 skip scatter
    // complex128 (planar): imaginary planes, the physical index handled by this launch, accumulate into out
    const double* Zi = nullptr; const double* Ai = nullptr; double* out_i = nullptr;
    int s_sel = 0, accumulate = 0;
};

// KT = KAp/16 = NEp/16 (square case: contracted and open leg pairs have the same padded size).
// Register-resident variant: KT waves per workgroup, wave w owns the open-ket block e in [16 w, 16 w + 16).  Step A produces the
// W tiles (all KT contracted-bra blocks x this e block) in registers in the MFMA C/D layout, which IS the A-operand layout
// of step B (lane l, register r of tile j holds W[kb = 16 j + 4 r + (l >> 4)][e = l & 15] = row e, k-slice r of block j), so W
// never travels through LDS and the only workgroup barriers are the two around the staging of Z.
// FULL: no padding anywhere (D1 D2 = 16 KT = E1 E2, e.g. D = 4, 8): every output element exists, the stores carry no predicate.
// What one (x,y) pair costs besides its 2 p KT^3 * 4 matrix instructions runs on the same four waves, one per SIMD, with the matrix pipe
// idle: at D = 8 the gather, the scatter and the loop bookkeeping were a third of the kernel (ablation with CTM_LAYER2_DBG, DESIGN.md
// section 5 "Round 5") -- hence no 64-bit divisions in the loop (the pair index advances as (x, y) with carry), no per-element
// branches in the gather (entries a padded shape does not have go to a padding slot of the LDS image) and, in the FULL
// instantiation, none in the scatter.
template <int KT, bool FULL>
__global__ __launch_bounds__(64 * KT) void layer2_reg_kernel(Layer2Params p) {
    constexpr int KAp = 16 * KT, NEp = 16 * KT, NTH = 64 * KT;
    constexpr int NZ = (KAp * KAp + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Zs = smem;                               // KAp x ldz
    double* As = Zs + (size_t)KAp * p.ldz;           // KAp x lda   (row kk: [s][e], e padded to NEp)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int tot = KAp * (p.ldz + p.lda);
    for (int q = tid; q < tot; q += NTH) smem[q] = 0.0;
    __syncthreads();
    {
        const int na = p.KA * p.p * p.NE;
        for (int q = tid; q < na; q += NTH) {
            const int e = q % p.NE, s = (q / p.NE) % p.p, kk = q / (p.NE * p.p);
            As[kk * p.lda + s * NEp + e] = p.A[q];
        }
    }
    long long zoff[NZ]; int zdst[NZ];
    const int nz = p.KA * p.KA;
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        const int e = tid + NTH * j;
        if (e < nz) {
            const int c2b = e % p.D2; int r = e / p.D2;
            const int c2k = r % p.D2; r /= p.D2;
            const int c1b = r % p.D1; const int c1k = r / p.D1;
            zoff[j] = c1k * p.zs_c1k + c1b * p.zs_c1b + c2k * p.zs_c2k + c2b * p.zs_c2b;
            zdst[j] = (c1b * p.D2 + c2b) * p.ldz + (c1k * p.D2 + c2k);
        } else { zoff[j] = 0; zdst[j] = KAp; }          // (row 0, column KAp: the padding column of the image, never an operand)
    }
    long long ooff[KT][4]; bool ook[KT][4];
#pragma unroll
    for (int n = 0; n < KT; ++n) {
        const int E = n * 16 + lr;
        const long long ob = (long long)(E / p.E2) * p.os_e1b + (long long)(E % p.E2) * p.os_e2b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = wid * 16 + lk + 4 * r;
            ook[n][r] = E < p.NE && e < p.NE && !(p.dbg & 4);
            ooff[n][r] = (long long)(e / p.E2) * p.os_e1k + (long long)(e % p.E2) * p.os_e2k + ob;
        }
    }
    const long long npair = (long long)p.nx * p.ny;
    double zreg[NZ];
    // pair q = x ny + y; q advances by the grid size: (x, y) += (gx, gy) with carry
    const int gx = (int)(gridDim.x / (unsigned)p.ny), gy = (int)(gridDim.x % (unsigned)p.ny);
    int x = (int)(blockIdx.x / (unsigned)p.ny), y = (int)(blockIdx.x % (unsigned)p.ny);
    if ((long long)blockIdx.x < npair) {
        const double* z = p.Z + x * p.zs_x + y * p.zs_y;
#pragma unroll
        for (int j = 0; j < NZ; ++j) zreg[j] = z[zoff[j]];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NZ; ++j) Zs[zdst[j]] = zreg[j];

    for (long long q = blockIdx.x; q < npair; q += gridDim.x) {
        __syncthreads();                               // Zs of this pair is complete
        const long long qn = q + gridDim.x;
        int xn = x + gx, yn = y + gy;
        if (yn >= p.ny) { yn -= p.ny; xn += 1; }
        if (qn < npair && !(p.dbg & 1)) {
            const double* z = p.Z + xn * p.zs_x + yn * p.zs_y;
#pragma unroll
            for (int j = 0; j < NZ; ++j) zreg[j] = z[zoff[j]];
        }
        d4 acc[KT];
#pragma unroll
        for (int n = 0; n < KT; ++n) acc[n] = (d4){0., 0., 0., 0.};
        for (int s = 0; s < ((p.dbg & 2) ? 0 : p.p); ++s) {
            const double* Asl = As + s * NEp;
            // The B operands are fetched AHEAD of the matrix instructions that consume them, into registers of their own (left to itself
            // the compiler reuses one register pair for all of them: ds_read, s_waitcnt lgkmcnt(0), two or four MFMAs, ds_read ...).
            // ---- step A: W[kb][e] = sum_kk Zs[kb][kk] As[kk][s][e]   (e block of this wave, every kb block)
            double bA[KAp / 4];
#pragma unroll
            for (int kq = 0; kq < KAp / 4; ++kq) bA[kq] = Asl[(lk + 4 * kq) * p.lda + wid * 16 + lr];
            double bB[2][KT];                                  // step B operands of two consecutive (j, r): double buffer
#pragma unroll
            for (int n = 0; n < KT; ++n) bB[0][n] = Asl[lk * p.lda + lr + n * 16];
            __builtin_amdgcn_sched_barrier(0);
            d4 w[KT];
#pragma unroll
            for (int j = 0; j < KT; ++j) w[j] = (d4){0., 0., 0., 0.};
#pragma unroll
            for (int k0 = 0; k0 < KAp; k0 += 4) {
                const double b = bA[k0 / 4];
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    const double a = Zs[(j * 16 + lr) * p.ldz + lk + k0];
                    w[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w[j], 0, 0, 0);
                }
            }
            // ---- step B: O[e][E] += sum_kb W[kb][e] As[kb][s][E]   (W straight from the accumulator registers)
#pragma unroll
            for (int jr = 0; jr < 4 * KT; ++jr) {
                const int j = jr >> 2, r = jr & 3;
                if (jr + 1 < 4 * KT) {
                    const double* brow = Asl + (((jr + 1) >> 2) * 16 + 4 * ((jr + 1) & 3) + lk) * p.lda + lr;
#pragma unroll
                    for (int n = 0; n < KT; ++n) bB[(jr + 1) & 1][n] = brow[n * 16];
                }
                __builtin_amdgcn_sched_barrier(0);
                const double a = w[j][r];
#pragma unroll
                for (int n = 0; n < KT; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bB[jr & 1][n], acc[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double* o = p.out + x * p.os_x + y * p.os_y;
#pragma unroll
        for (int n = 0; n < KT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (FULL || ook[n][r]) o[ooff[n][r]] = acc[n][r];
        __syncthreads();                               // every wave finished reading Zs
        if (qn < npair) {
#pragma unroll
            for (int j = 0; j < NZ; ++j) Zs[zdst[j]] = zreg[j];
        }
        x = xn; y = yn;
    }
}

// complex128 variant of the register-resident kernel (planar data, one physical index s per launch so that the two planes of
// Z and of the site slice a[.,s,.] fit in LDS together; launches s > 0 accumulate into the output):
//   W = Z a_s            Wr = Zr Ar - Zi Ai          Wi = Zr Ai + Zi Ar
//   O += W^T conj(a_s)   Or += Wr Ar + Wi Ai         Oi += Wi Ar - Wr Ai
// four real FP64 MFMAs per complex tile step, operands negated on the fly where a product is subtracted.
template <int KT>
__global__ __launch_bounds__(64 * KT) void layer2_c_kernel(Layer2Params p) {
    constexpr int KAp = 16 * KT, NEp = 16 * KT, NTH = 64 * KT;
    constexpr int NZ = (KAp * KAp + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ldz = KAp + 1, lda = NEp + 1;
    double* Zr = smem;
    double* Zi = Zr + KAp * ldz;
    double* Ar = Zi + KAp * ldz;
    double* Ai = Ar + KAp * lda;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int tot = 2 * KAp * (ldz + lda);
    for (int q = tid; q < tot; q += NTH) smem[q] = 0.0;
    __syncthreads();
    {
        const int na = p.KA * p.NE;
        for (int q = tid; q < na; q += NTH) {
            const int e = q % p.NE, kk = q / p.NE;
            const size_t src = ((size_t)kk * p.p + p.s_sel) * p.NE + e;
            Ar[kk * lda + e] = p.A[src]; Ai[kk * lda + e] = p.Ai[src];
        }
    }
    long long zoff[NZ]; int zdst[NZ];
    const int nz = p.KA * p.KA;
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        const int e = tid + NTH * j;
        if (e < nz) {
            const int c2b = e % p.D2; int r = e / p.D2;
            const int c2k = r % p.D2; r /= p.D2;
            const int c1b = r % p.D1; const int c1k = r / p.D1;
            zoff[j] = c1k * p.zs_c1k + c1b * p.zs_c1b + c2k * p.zs_c2k + c2b * p.zs_c2b;
            zdst[j] = (c1b * p.D2 + c2b) * ldz + (c1k * p.D2 + c2k);
        } else { zoff[j] = 0; zdst[j] = KAp; }          // (padding column of row 0: never an operand)
    }
    long long ooff[KT][4]; bool ook[KT][4];
#pragma unroll
    for (int n = 0; n < KT; ++n) {
        const int E = n * 16 + lr;
        const long long ob = (long long)(E / p.E2) * p.os_e1b + (long long)(E % p.E2) * p.os_e2b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = wid * 16 + lk + 4 * r;
            ook[n][r] = E < p.NE && e < p.NE;
            ooff[n][r] = (long long)(e / p.E2) * p.os_e1k + (long long)(e % p.E2) * p.os_e2k + ob;
        }
    }
    const long long npair = (long long)p.nx * p.ny;
    double zr[NZ], zi[NZ];
    // pair q = x ny + y advances by the grid size as (x, y) += (gx, gy) with carry (no 64-bit divisions in the loop, see layer2_reg_kernel)
    const int gx = (int)(gridDim.x / (unsigned)p.ny), gy = (int)(gridDim.x % (unsigned)p.ny);
    int x = (int)(blockIdx.x / (unsigned)p.ny), y = (int)(blockIdx.x % (unsigned)p.ny);
    if ((long long)blockIdx.x < npair) {
        const long long zo = x * p.zs_x + y * p.zs_y;
#pragma unroll
        for (int j = 0; j < NZ; ++j) { zr[j] = p.Z[zo + zoff[j]]; zi[j] = p.Zi[zo + zoff[j]]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NZ; ++j) { Zr[zdst[j]] = zr[j]; Zi[zdst[j]] = zi[j]; }

    for (long long q = blockIdx.x; q < npair; q += gridDim.x) {
        __syncthreads();
        const long long qn = q + gridDim.x;
        int xn = x + gx, yn = y + gy;
        if (yn >= p.ny) { yn -= p.ny; xn += 1; }
        if (qn < npair) {
            const long long zo = xn * p.zs_x + yn * p.zs_y;
#pragma unroll
            for (int j = 0; j < NZ; ++j) { zr[j] = p.Z[zo + zoff[j]]; zi[j] = p.Zi[zo + zoff[j]]; }
        }
        d4 wr[KT], wi[KT], accr[KT], acci[KT];
#pragma unroll
        for (int n = 0; n < KT; ++n) { wr[n] = (d4){0., 0., 0., 0.}; wi[n] = wr[n]; accr[n] = wr[n]; acci[n] = wr[n]; }
#pragma unroll
        for (int k0 = 0; k0 < KAp; k0 += 4) {
            const double br = Ar[(lk + k0) * lda + wid * 16 + lr], bi = Ai[(lk + k0) * lda + wid * 16 + lr];
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const double ar = Zr[(j * 16 + lr) * ldz + lk + k0], ai = Zi[(j * 16 + lr) * ldz + lk + k0];
                wr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, wr[j], 0, 0, 0);
                wr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, wr[j], 0, 0, 0);
                wi[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, wi[j], 0, 0, 0);
                wi[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, wi[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < KT; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double a_r = wr[j][r], a_i = wi[j][r];
                const int row = (j * 16 + 4 * r + lk) * lda + lr;
#pragma unroll
                for (int n = 0; n < KT; ++n) {
                    const double Br = Ar[row + n * 16], Bi = Ai[row + n * 16];
                    accr[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_r, Br, accr[n], 0, 0, 0);
                    accr[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_i, Bi, accr[n], 0, 0, 0);
                    acci[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_i, Br, acci[n], 0, 0, 0);
                    acci[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_r, Bi, acci[n], 0, 0, 0);
                }
            }
        }
        const long long oo = x * p.os_x + y * p.os_y;
#pragma unroll
        for (int n = 0; n < KT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ook[n][r]) {
                    const long long o = oo + ooff[n][r];
                    if (p.accumulate) { p.out[o] += accr[n][r]; p.out_i[o] += acci[n][r]; }
                    else { p.out[o] = accr[n][r]; p.out_i[o] = acci[n][r]; }
                }
        __syncthreads();
        if (qn < npair) {
#pragma unroll
            for (int j = 0; j < NZ; ++j) { Zr[zdst[j]] = zr[j]; Zi[zdst[j]] = zi[j]; }
        }
        x = xn; y = yn;
    }
}

template <int KT>
int launch_layer2_c(ctm_ctx* ctx, const Layer2Params& p) {
    const size_t lds_bytes = sizeof(double) * 2 * (size_t)(16 * KT) * (2 * 16 * KT + 2);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] { (void)hipFuncSetAttribute((const void*)layer2_c_kernel<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    const long long npair = (long long)p.nx * p.ny;
    const int per_cu = std::max(1, std::min(8, (int)((150 * 1024) / lds_bytes)));
    const int grid = (int)std::min<long long>(npair, 256LL * per_cu * 2);
    CTM_LAUNCH(ctx, layer2_c_kernel<KT>, dim3(grid), dim3(64 * KT), lds_bytes, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ctx->set_error(std::string("layer2 launch: ") + hipGetErrorString(e)); return CTM_ERR_HIP; }
    return CTM_OK;
}

template <int KT, bool FULL>
int launch_layer2_reg_t(ctm_ctx* ctx, const Layer2Params& p) {
    const size_t lds_bytes = sizeof(double) * (size_t)p.KAp * (p.ldz + p.lda);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] { (void)hipFuncSetAttribute((const void*)layer2_reg_kernel<KT, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    const long long npair = (long long)p.nx * p.ny;
    const int per_cu = std::max(1, std::min(8, (int)((150 * 1024) / lds_bytes)));
    const int grid = (int)std::min<long long>(npair, 256LL * per_cu * 2);
    CTM_LAUNCH(ctx, (layer2_reg_kernel<KT, FULL>), dim3(grid), dim3(64 * KT), lds_bytes, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ctx->set_error(std::string("layer2 launch: ") + hipGetErrorString(e)); return CTM_ERR_HIP; }
    return CTM_OK;
}

template <int KT>
int launch_layer2_reg(ctm_ctx* ctx, const Layer2Params& p) {
    const bool full = p.KA == p.KAp && p.NE == p.NEp && p.dbg == 0;
    return full ? launch_layer2_reg_t<KT, true>(ctx, p) : launch_layer2_reg_t<KT, false>(ctx, p);
}

long long stride_of(const std::string& idx, const DT& t, char ch) {
    long long s = 1;
    for (int a = (int)idx.size() - 1; a >= 0; --a) { if (idx[a] == ch) return s; s *= t.dims[a]; }
    return -1;
}

}  // namespace

// Is the pair of site operands (a with indices ia, conj(a) with ib) absorbable by the fused kernel into a tensor Z
// with indices iz?  Fills the role letters.
bool layer2_roles(const std::string& iz, const DT& Z, const std::string& ia, const std::string& ib, const DT& A,
                  char ck[2], char cb[2], char ek[2], char eb[2], char sp[2]) {
    if (ia.size() != 5 || ib.size() != 5 || A.dims.size() != 5 || ia[0] != ib[0]) return false;
    int nc = 0, ne = 0;
    for (int pos = 1; pos < 5; ++pos) {
        const bool inz_k = iz.find(ia[pos]) != std::string::npos, inz_b = iz.find(ib[pos]) != std::string::npos;
        if (inz_k != inz_b) return false;
        if (inz_k) { if (nc == 2) return false; ck[nc] = ia[pos]; cb[nc] = ib[pos]; ++nc; }
        else { if (ne == 2) return false; ek[ne] = ia[pos]; eb[ne] = ib[pos]; ++ne; }
    }
    if (nc != 2 || ne != 2) return false;
    int ns = 0;
    for (char ch : iz) {
        if (ch == ck[0] || ch == ck[1] || ch == cb[0] || ch == cb[1]) continue;
        if (ns == 2) return false;
        sp[ns++] = ch;
    }
    if (ns != 2) return false;
    for (int i = 1; i < 5; ++i) if (A.dims[i] > 8) return false;
    {   // the fused kernel handles equal padded sizes of the contracted and the open leg pair
        auto dimA = [&](char ch) { return (int)A.dims[ia.find(ch)]; };
        const int KA = dimA(ck[0]) * dimA(ck[1]), NE = dimA(ek[0]) * dimA(ek[1]);
        if ((KA + 15) / 16 != (NE + 15) / 16) return false;
    }
    (void)Z;
    return true;
}

// out[io] = sum Z[iz] a[ia] a[ib]  (both site operands are the same real tensor A); io must contain exactly the two
// spectator letters and the four open letters in any order.
int dev_layer2(ctm_ctx* ctx, const std::string& iz, const DT& Z, const std::string& ia, const std::string& ib, const DT& A,
               const std::string& io, DT* out) {
    char ck[2], cb[2], ek[2], eb[2], sp[2];
    if (!layer2_roles(iz, Z, ia, ib, A, ck, cb, ek, eb, sp)) { ctx->set_error("layer2: pattern mismatch"); return CTM_ERR_BADARG; }
    // choose "leg 2" = the contracted leg whose bra index is the fastest-varying in Z (coalesced gather)
    if (stride_of(iz, Z, cb[0]) < stride_of(iz, Z, cb[1])) { std::swap(ck[0], ck[1]); std::swap(cb[0], cb[1]); }
    // the open legs keep the site-tensor order: e1 before e2
    auto dimA = [&](char ch) { return (int)A.dims[ia.find(ch)]; };
    Layer2Params p;
    p.D1 = dimA(ck[0]); p.D2 = dimA(ck[1]); p.E1 = dimA(ek[0]); p.E2 = dimA(ek[1]); p.p = (int)A.dims[0];
    p.KA = p.D1 * p.D2; p.NE = p.E1 * p.E2;
    p.KAp = (p.KA + 15) / 16 * 16; p.NEp = (p.NE + 15) / 16 * 16;
    if (p.KAp > 64 || p.NEp > 64 || p.KAp != p.NEp) { ctx->set_error("layer2: shape not supported by the fused kernel"); return CTM_ERR_UNSUPPORTED; }
    p.ldz = p.KAp + 1; p.lda = p.p * p.NEp + 1; p.ldw = p.NEp + 1;
    const size_t lds_bytes = sizeof(double) * (size_t)p.KAp * (p.ldz + p.lda + p.ldw);
    if (lds_bytes > 150 * 1024) { ctx->set_error("layer2: LDS budget"); return CTM_ERR_UNSUPPORTED; }
    ArenaScope scope(ctx);
    // site tensor image [c1][c2][s][e1][e2]
    const bool cx = (A.q != nullptr);
    if (cx && !(Z.q && out->q)) { ctx->set_error("layer2: complex operands need two planes"); return CTM_ERR_BADARG; }
    double* Ap;
    CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)A.numel() * (cx ? 2 : 1), (void**)&Ap));
    {
        long long dims[5]; int perm[5];
        for (int i = 0; i < 5; ++i) dims[i] = A.dims[i];
        perm[0] = (int)ia.find(ck[0]); perm[1] = (int)ia.find(ck[1]); perm[2] = 0; perm[3] = (int)ia.find(ek[0]); perm[4] = (int)ia.find(ek[1]);
        CTM_TRY(permute_f64(ctx, A.p, Ap, 5, dims, perm));
        if (cx) CTM_TRY(permute_f64(ctx, A.q, Ap + A.numel(), 5, dims, perm));
    }
    p.A = Ap;
    p.Z = Z.p;
    if (cx) { p.Ai = Ap + A.numel(); p.Zi = Z.q; }
    p.zs_x = stride_of(iz, Z, sp[0]); p.zs_y = stride_of(iz, Z, sp[1]);
    p.nx = (int)Z.dims[iz.find(sp[0])]; p.ny = (int)Z.dims[iz.find(sp[1])];
    p.zs_c1k = stride_of(iz, Z, ck[0]); p.zs_c1b = stride_of(iz, Z, cb[0]);
    p.zs_c2k = stride_of(iz, Z, ck[1]); p.zs_c2b = stride_of(iz, Z, cb[1]);
    // output tensor
    DT O;
    O.dims.resize(io.size());
    for (size_t a = 0; a < io.size(); ++a) {
        const char ch = io[a];
        if (ch == sp[0]) O.dims[a] = p.nx; else if (ch == sp[1]) O.dims[a] = p.ny;
        else if (ch == ek[0] || ch == eb[0]) O.dims[a] = p.E1; else if (ch == ek[1] || ch == eb[1]) O.dims[a] = p.E2;
        else { ctx->set_error("layer2: unexpected output index"); return CTM_ERR_BADARG; }
    }
    if (io.size() != 6) { ctx->set_error("layer2: output rank"); return CTM_ERR_BADARG; }
    O.p = out->p;
    if (!O.p) {
        // allocate OUTSIDE this function's scope: caller-visible result must survive -> use the caller's arena position
        ctx->set_error("layer2: output buffer must be provided"); return CTM_ERR_BADARG;
    }
    p.out = O.p; p.out_i = out->q;
    p.os_x = stride_of(io, O, sp[0]); p.os_y = stride_of(io, O, sp[1]);
    p.os_e1k = stride_of(io, O, ek[0]); p.os_e1b = stride_of(io, O, eb[0]);
    p.os_e2k = stride_of(io, O, ek[1]); p.os_e2b = stride_of(io, O, eb[1]);
    { static const int dbg_env = getenv("CTM_LAYER2_DBG") ? atoi(getenv("CTM_LAYER2_DBG")) : 0; p.dbg = dbg_env; }     // development (timing ablations; results are wrong): 1 skip gather, 2 skip MFMA, 4 skip scatter
    const long long npair = (long long)p.nx * p.ny;
    const double fl = 2.0 * npair * p.p * ((double)p.KA * p.KA * p.NE + (double)p.KA * p.NE * p.NE) * (cx ? 4.0 : 1.0);
    const int ev = timing_begin(ctx);
    int st = CTM_OK;
    if (cx) {
        for (int s_ = 0; s_ < p.p && st == CTM_OK; ++s_) {
            p.s_sel = s_; p.accumulate = s_ > 0;
            switch (p.KAp / 16) {
                case 1: st = launch_layer2_c<1>(ctx, p); break;
                case 2: st = launch_layer2_c<2>(ctx, p); break;
                case 3: st = launch_layer2_c<3>(ctx, p); break;
                default: st = launch_layer2_c<4>(ctx, p); break;
            }
        }
    } else
    switch (p.KAp / 16) {
        case 1: st = launch_layer2_reg<1>(ctx, p); break;
        case 2: st = launch_layer2_reg<2>(ctx, p); break;
        case 3: st = launch_layer2_reg<3>(ctx, p); break;
        default: st = launch_layer2_reg<4>(ctx, p); break;
    }
    timing_end(ctx, ev, 2, fl);
    CTM_TRY(st);
    ctx->layer2_flops += fl;
    ctx->layer2_calls += 1;
    out->dims = O.dims; out->cj = false;
    return CTM_OK;
}
