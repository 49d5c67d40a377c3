// FP64 GEMM for gfx950 on the matrix cores: v_mfma_f64_16x16x4_f64, LDS-staged, double-buffered.
//
// This is the kernel every heavy contraction of the CTM move lowers to (SURVEY 2.3 K2,K6,K7,K11,
// K12,K13 and the Gram/apply steps of the block-Jacobi SVD).  Design (CDNA4):
//   * block = 256 threads = 4 wave64 in a 2x2 grid; each wave owns TM x TN MFMA tiles of 16x16
//     (accumulators stay in registers for the whole K loop: 4 f64 per lane per tile);
//   * A and B tiles are staged global -> VGPR -> LDS in a k-major image As[k][m], Bs[k][n] so that
//     the MFMA operand fetch (lane l: row/col l&15, k = l>>4) is one ds_read_b64 per operand;
//   * two LDS buffers, one barrier per K tile: the global loads of tile t+1 are issued before
//     the MFMA work of tile t and written to the other buffer after it;
//   * operands are addressed with arbitrary (row, k) strides + an optional two-segment row
//     map, so transposed operands, tensor slices and the (i,j) row-panel pairs of the Jacobi
//     sweep need no gather copy; batching via blockIdx.z;
//   * the blockIdx -> tile map walks tiles XCD-major so that the 8 XCD-private L2s each see a
//     compact band of B columns.
// FP64 MFMA peak on MI355X is 78.6 TFLOP/s (32 flop/clk/SIMD): one MFMA (2048 flop) issues every
// 64 cycles, so LDS and global bandwidth needs are modest; the kernel is issue-bound on MFMA when
// K is long.  C/D fragment layout of the f64 MFMA: col = lane&15, row = (lane>>4) + 4*reg.
#include "ctm_common.h"
#include <map>
#include <atomic>
#include <mutex>
#include <mutex>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 16;
constexpr int PAD = 2;

struct GemmParams {
    int M, N, K;
    const double* A; long long sam, sak;
    const double* B; long long sbk, sbn;
    double* C; long long ldc;
    double alpha, beta;
    long long strideA, strideB, strideC;
    const GemmOff* offs;
    int splitA, splitB, splitC, splitB_dim;
    const double* colscale;
    const int* skip_flags;
    const int* skip_all;                           // device word: the whole launch does nothing when *skip_all == 0
    int tilesM, tilesN;
    int ksplit, klen; long long split_stride;      // split-K: blockIdx.z = K slice, partial C written at z*split_stride
};

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmParams p) {
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int LDA = BM + PAD, LDB = BN + PAD;
    constexpr int EA = BM * BK / 256, EB = BN * BK / 256;
    __shared__ double smem[2 * BK * LDA + 2 * BK * LDB];
    double* As = smem;
    double* Bs = smem + 2 * BK * LDA;

    if (p.skip_flags && p.skip_flags[blockIdx.z] == 0) return;   // uniform per workgroup
    if (p.skip_all && *p.skip_all == 0) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // XCD-aware tile walk: consecutive block ids land on consecutive XCDs (b % 8); give each XCD
    // a contiguous range of tiles (bijective for any tile count).
    const int ntiles = p.tilesM * p.tilesN;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = bid % p.tilesM, tn = bid / p.tilesM;
    const int m0 = tm * BM, n0 = tn * BN;

    long long a0, a1, b0, b1, c0, c1;
    if (p.offs) {
        const GemmOff o = p.offs[blockIdx.z];
        a0 = o.a0; a1 = o.a1; b0 = o.b0; b1 = o.b1; c0 = o.c0; c1 = o.c1;
        if (o.klen > 0) p.K = o.klen;
    } else if (p.ksplit > 1) {
        const long long k0 = (long long)blockIdx.z * p.klen;
        a0 = a1 = k0 * p.sak;
        b0 = b1 = k0 * p.sbk;
        c0 = c1 = (long long)blockIdx.z * p.split_stride;
        p.K = (int)((p.K - k0) < p.klen ? (p.K - k0) : p.klen);
    } else {
        a0 = a1 = (long long)blockIdx.z * p.strideA;
        b0 = b1 = (long long)blockIdx.z * p.strideB;
        c0 = c1 = (long long)blockIdx.z * p.strideC;
    }

    const bool a_kfast = p.sak <= p.sam;   // lanes run along the faster global dimension
    const bool b_nfast = p.sbn <= p.sbk;

    // per-thread staging coordinates (tile-relative).  256 % BK == 0 and 256 % BM == 0, so one of
    // the two coordinates is the same for every staged element i and the other advances by a constant.
    const int a_k0 = a_kfast ? (tid % BK) : (tid / BM), a_m0 = a_kfast ? (tid / BK) : (tid % BM);
    const int a_dk = a_kfast ? 0 : (256 / BM), a_dm = a_kfast ? (256 / BK) : 0;
    const int b_k0 = b_nfast ? (tid / BN) : (tid % BK), b_n0 = b_nfast ? (tid % BN) : (tid / BK);
    const int b_dk = b_nfast ? (256 / BN) : 0, b_dn = b_nfast ? 0 : (256 / BK);

    d4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (d4){0., 0., 0., 0.};

    double ra[EA], rb[EB];
    const int nk = (p.K + BK - 1) / BK;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < EA; ++i) {
            const int k = k0 + a_k0 + i * a_dk, m = m0 + a_m0 + i * a_dm;
            double v = 0.0;
            if (m < p.M && k < p.K) {
                const long long off = (m < p.splitA ? a0 + (long long)m * p.sam : a1 + (long long)(m - p.splitA) * p.sam);
                v = p.A[off + (long long)k * p.sak];
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < EB; ++i) {
            const int k = k0 + b_k0 + i * b_dk, n = n0 + b_n0 + i * b_dn;
            double v = 0.0;
            if (n < p.N && k < p.K) {
                long long off;
                if (p.splitB_dim == 1)
                    off = (k < p.splitB ? b0 + (long long)k * p.sbk : b1 + (long long)(k - p.splitB) * p.sbk) + (long long)n * p.sbn;
                else if (p.splitB_dim == 2)
                    off = (n < p.splitB ? b0 + (long long)n * p.sbn : b1 + (long long)(n - p.splitB) * p.sbn) + (long long)k * p.sbk;
                else
                    off = b0 + (long long)k * p.sbk + (long long)n * p.sbn;
                v = p.B[off];
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        double* as = As + buf * BK * LDA;
        double* bs = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < EA; ++i) as[(a_k0 + i * a_dk) * LDA + a_m0 + i * a_dm] = ra[i];
#pragma unroll
        for (int i = 0; i < EB; ++i) bs[(b_k0 + i * b_dk) * LDB + b_n0 + i * b_dn] = rb[i];
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int lr = lane & 15, lk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const double* as = As + buf * BK * LDA + wm * 16 * TM + lr;
        const double* bs = Bs + buf * BK * LDB + wn * 16 * TN + lr;
#pragma unroll
        for (int k4 = 0; k4 < BK / 4; ++k4) {
            double af[TM], bf[TN];
            const int kr = k4 * 4 + lk;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = as[kr * LDA + i * 16];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = bs[kr * LDB + j * 16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: row = (lane>>4) + 4*r, col = lane&15 inside each 16x16 tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 16 * TM + i * 16 + lk + 4 * r;
            if (m >= p.M) continue;
            const long long crow = (m < p.splitC ? c0 + (long long)m * p.ldc : c1 + (long long)(m - p.splitC) * p.ldc);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * 16 * TN + j * 16 + lr;
                if (n >= p.N) continue;
                double v = p.alpha * acc[i][j][r];
                if (p.colscale) v *= p.colscale[n];
                if (p.beta != 0.0) v += p.beta * p.C[crow + n];
                p.C[crow + n] = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Fast path: plain (un-segmented, un-batched-by-offsets) GEMM whose M, N are multiples of 128 and K of 16, with
// 16-byte aligned operands.  Same 128x128x16 double-buffered structure, but every global access is a 16-byte
// vector load along the operand's contiguous dimension, addresses are one pointer bump per K tile, and there is
// no bounds / segment logic in the loop.  AK: A is k-contiguous (row-major M x K); BNF: B is n-contiguous (K x N).
// ---------------------------------------------------------------------------------------------------------
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool AK, bool BNF>
__global__ __launch_bounds__(256, 2) void gemm_f64_fast_kernel(GemmParams p) {
    constexpr int BM = 128, BN = 128, LDA = BM + PAD, LDB = BN + PAD;
    __shared__ __attribute__((aligned(16))) double smem[2 * BK * LDA + 2 * BK * LDB];
    double* As = smem;
    double* Bs = smem + 2 * BK * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int ntiles = p.tilesM * p.tilesN;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = bid % p.tilesM, tn = bid / p.tilesM;
    const long long zb = blockIdx.z;
    const double* Ab = p.A + zb * p.strideA;
    const double* Bb = p.B + zb * p.strideB;
    double* Cb = p.C + zb * p.strideC;

    // staging: 4 x d2 per operand per thread
    const double* ap[4]; const double* bp[4];
    int a_lds[4], b_lds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = tid + 256 * i;
        if (AK) { const int row = u >> 3, kp = u & 7; ap[i] = Ab + (long long)(tm * BM + row) * p.sam + 2 * kp; a_lds[i] = (2 * kp) * LDA + row; }
        else    { const int kk = u >> 6, mp = u & 63; ap[i] = Ab + (long long)kk * p.sak + tm * BM + 2 * mp; a_lds[i] = kk * LDA + 2 * mp; }
        if (BNF) { const int kk = u >> 6, np = u & 63; bp[i] = Bb + (long long)kk * p.sbk + tn * BN + 2 * np; b_lds[i] = kk * LDB + 2 * np; }
        else     { const int col = u >> 3, kp = u & 7; bp[i] = Bb + (long long)(tn * BN + col) * p.sbn + 2 * kp; b_lds[i] = (2 * kp) * LDB + col; }
    }
    const long long a_step = AK ? BK : (long long)BK * p.sak;
    const long long b_step = BNF ? (long long)BK * p.sbk : BK;

    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0., 0., 0., 0.};
    d2 ra[4], rb[4];
    const int nk = p.K / BK;
    auto load_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[i] = *(const d2*)ap[i]; ap[i] += a_step; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { rb[i] = *(const d2*)bp[i]; bp[i] += b_step; }
    };
    auto store_tile = [&](int buf) {
        double* as = As + buf * BK * LDA;
        double* bs = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (AK) { as[a_lds[i]] = ra[i][0]; as[a_lds[i] + LDA] = ra[i][1]; }
            else *(d2*)(as + a_lds[i]) = ra[i];
            if (BNF) *(d2*)(bs + b_lds[i]) = rb[i];
            else { bs[b_lds[i]] = rb[i][0]; bs[b_lds[i] + LDB] = rb[i][1]; }
        }
    };
    load_tile();
    store_tile(0);
    __syncthreads();
    const int lr = lane & 15, lk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile();
        const double* as = As + buf * BK * LDA + wm * 64 + lr;
        const double* bs = Bs + buf * BK * LDB + wn * 64 + lr;
#pragma unroll
        for (int k4 = 0; k4 < BK / 4; ++k4) {
            double af[4], bf[4];
            const int kr = k4 * 4 + lk;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = as[kr * LDA + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = bs[kr * LDB + j * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = tm * BM + wm * 64 + i * 16 + lk + 4 * r;
            double* crow = Cb + (long long)m * p.ldc + tn * BN + wn * 64 + lr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double v = p.alpha * acc[i][j][r];
                if (p.colscale) v *= p.colscale[tn * BN + wn * 64 + j * 16 + lr];
                if (p.beta != 0.0) v += p.beta * crow[j * 16];
                crow[j * 16] = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// Strip kernel: C (M x N, M <= 64) = A (M x K, k-contiguous rows) * B, with B a large K x N operand that is read exactly
// once -- a block of <= 64 vectors times an n x n enlarged corner in the block power / block Krylov iterations.  At
// M <= 32 this is HBM-bound (2.15 GB of corner per pass at n = 16384), so the kernel is built to stream: no LDS, no
// barriers; every wave loads its operands straight from global memory in MFMA operand layout with 16-byte loads and
// keeps one 16-deep K block in flight behind the one it is multiplying.
//   * a wave owns all TM*16 rows x 32 columns; a 16-deep K block is contracted by 4 MFMAs per tile whose k index is
//     k = 2*(lane>>4) + t for t = 0, 1 and 8 + 2*(lane>>4) + t - 2 for t = 2, 3 (t = instruction): a lane's A operands are two
//     16-byte loads, each covering with the 4 lanes of its row 64 contiguous bytes;
//   * BNF (B n-contiguous, K x N row major): a lane loads the double2 B[k][c0 + 2*(lane&15) .. +1] of its four rows k(t);
//     element 0 feeds column tile 0, element 1 column tile 1 (columns interleaved), so 16 lanes cover 256 contiguous
//     bytes of a B row, the 4 waves of a workgroup 1 KB, and the partial results are stored as double2;
//   * !BNF (B k-contiguous, stored N x K): same pattern as A, column tile j = columns c0 + 16 j + (lane&15);
//   * K is split over blockIdx.y (ks slices) into partial products that the fixed-order reduce kernel sums (deterministic).
// ---------------------------------------------------------------------------------------------------------
struct StripParams {
    int M, N, K;
    const double* A; long long sam;
    const double* B; long long ldb;          // BNF: row stride of K x N ; !BNF: row stride of N x K
    double* P;                               // partials [ks][M][N]
    int klen;
    // row-block kernel only: the K-slice partials are summed inside the launch by the last workgroup to finish a column tile
    // (fixed slice order: deterministic), which writes C = alpha * sum [* colscale] + beta * C.  ks == 1: written directly.
    double* C = nullptr; long long ldc = 0; double alpha = 1.0, beta = 0.0; const double* colscale = nullptr;
    unsigned* cnt = nullptr; int ks = 1;
};

typedef double d4v __attribute__((ext_vector_type(4)));

template <int TM, bool BNF>
__global__ __launch_bounds__(256) void gemm_strip_kernel(StripParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int c0 = (blockIdx.x * 4 + wid) * 32;
    if (c0 >= p.N) return;
    const long long kbeg = (long long)blockIdx.y * p.klen;
    const int nblk = (int)((std::min<long long>(p.K, kbeg + p.klen) - kbeg) / 16);
    if (nblk <= 0) return;

    // k index of MFMA instruction t in a 16-deep block: k = 2*lk + t (t = 0, 1), 8 + 2*lk + (t - 2) (t = 2, 3): each 16-byte
    // load of a k-contiguous operand then covers 64 contiguous bytes per row across the 4 lanes of that row
    const double* ap[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) ap[i] = p.A + (long long)std::min(i * 16 + lr, p.M - 1) * p.sam + kbeg + 2 * lk;
    const double* bp[2];
    if (BNF) { bp[0] = p.B + (kbeg + 2 * lk) * p.ldb + c0 + 2 * lr; bp[1] = nullptr; }
    else     { bp[0] = p.B + (long long)(c0 + lr) * p.ldb + kbeg + 2 * lk; bp[1] = bp[0] + 16 * p.ldb; }
    const long long bstep = BNF ? 16 * p.ldb : 16;

    d4 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) { acc[i][0] = (d4){0., 0., 0., 0.}; acc[i][1] = (d4){0., 0., 0., 0.}; }
    // two register sets: the loads of block t+1 are in flight while block t is multiplied (a third set measured no faster
    // at <= 32 rows and slower at 64 rows, where it costs a wave of occupancy)
    d2 a0[TM][2], a1[TM][2];
    d2 bn0[4], bn1[4];         // BNF: rows of instruction t = 0..3
    d2 bt0[2][2], bt1[2][2];   // !BNF: [column tile j][low / high half of the block]

    auto load = [&](d2 (*a)[2], d2* bn, d2 (*bt)[2]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { a[i][0] = *(const d2*)ap[i]; a[i][1] = *(const d2*)(ap[i] + 8); ap[i] += 16; }
        if (BNF) {
            bn[0] = __builtin_nontemporal_load((const d2*)bp[0]);
            bn[1] = __builtin_nontemporal_load((const d2*)(bp[0] + p.ldb));
            bn[2] = __builtin_nontemporal_load((const d2*)(bp[0] + 8 * p.ldb));
            bn[3] = __builtin_nontemporal_load((const d2*)(bp[0] + 9 * p.ldb));
            bp[0] += bstep;
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bt[j][0] = __builtin_nontemporal_load((const d2*)bp[j]);
                bt[j][1] = __builtin_nontemporal_load((const d2*)(bp[j] + 8));
                bp[j] += bstep;
            }
        }
    };
    auto compute = [&](const d2 (*a)[2], const d2* bn, const d2 (*bt)[2]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const double av = a[i][t >> 1][t & 1];
                acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, BNF ? bn[t][0] : bt[0][t >> 1][t & 1], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, BNF ? bn[t][1] : bt[1][t >> 1][t & 1], acc[i][1], 0, 0, 0);
            }
    };
    load(a0, bn0, bt0);
    int kt = 0;
    for (; kt + 2 <= nblk; kt += 2) {
        load(a1, bn1, bt1);
        compute(a0, bn0, bt0);
        if (kt + 2 < nblk) load(a0, bn0, bt0);
        compute(a1, bn1, bt1);
    }
    if (kt < nblk) compute(a0, bn0, bt0);

    double* P = p.P + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = i * 16 + lk + 4 * r;
            if (m >= p.M) continue;
            double* row = P + (long long)m * p.N + c0;
            if (BNF) *(d2*)(row + 2 * lr) = (d2){acc[i][0][r], acc[i][1][r]};
            else { row[lr] = acc[i][0][r]; row[16 + lr] = acc[i][1][r]; }
        }
}

// ---------------------------------------------------------------------------------------------------------
// Row-block kernel: the same product (<= 64 k-contiguous rows times a big operand read once, K split into partials) for the
// MFMA-bound block sizes (33..64 rows: 16 flop per byte of the big operand).  There the LDS-free strip kernel is limited by
// what it re-reads through L1/L2: each of its waves loads the whole A block for its 32 columns (A traffic = 2x the B stream)
// and holds one K block in flight.  Here the workgroup shares one A tile: tile = all rows x 128 columns x 16 k, A and B
// staged global -> VGPR -> LDS with 16-byte loads into a k-major image (one ds_read_b64 per MFMA operand), double buffered,
// one barrier per K tile; wave w owns the rows x 32 columns block w.  50 KB of LDS and ~120 VGPRs: three workgroups per CU
// keep three K tiles of the stream in flight per CU.
// ---------------------------------------------------------------------------------------------------------
template <int TMW, bool BNF, bool DEEP>
__global__ __launch_bounds__(256, 3) void gemm_rows_kernel(StripParams p) {
    constexpr int BM = 16 * TMW, BN = 128, LDA = BM + PAD, LDB = BN + PAD;
    constexpr int NA = (BM * BK / 2 + 255) / 256;                 // d2 loads of A per thread per K tile (1 or 2)
    __shared__ __attribute__((aligned(16))) double smem[2 * BK * LDA + 2 * BK * LDB];
    double* As = smem;
    double* Bs = smem + 2 * BK * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int n0 = blockIdx.x * BN;
    const long long kbeg = (long long)blockIdx.y * p.klen;
    const int nk = (int)((std::min<long long>(p.K, kbeg + p.klen) - kbeg) / BK);
    if (nk <= 0) return;

    const double* ap[NA]; int a_lds[NA]; bool a_on[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int u = tid + 256 * i, row = u >> 3, kp = u & 7;
        a_on[i] = row < BM;
        ap[i] = p.A + (long long)std::min(row, p.M - 1) * p.sam + kbeg + 2 * kp;
        a_lds[i] = (2 * kp) * LDA + row;
    }
    const double* bp[4]; int b_lds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = tid + 256 * i;
        if (BNF) { const int kk = u >> 6, np = u & 63; bp[i] = p.B + (kbeg + kk) * p.ldb + n0 + 2 * np; b_lds[i] = kk * LDB + 2 * np; }
        else     { const int col = u >> 3, kp = u & 7; bp[i] = p.B + (long long)(n0 + col) * p.ldb + kbeg + 2 * kp; b_lds[i] = (2 * kp) * LDB + col; }
    }
    const long long b_step = BNF ? (long long)BK * p.ldb : BK;

    d4 acc[TMW][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i) { acc[i][0] = (d4){0., 0., 0., 0.}; acc[i][1] = (d4){0., 0., 0., 0.}; }
    // Two register sets: while tile kt is multiplied out of LDS, tile kt + 1 waits in one set (loaded during the previous iteration, stored
    // to the other LDS buffer after the MFMAs) and tile kt + 2 is being loaded into the other -- a load has two compute phases to land.
    // With three workgroups per CU (big operands) one phase was enough; a mid-size operand (n = 4608: 36 column tiles x 7 slices) runs ONE
    // workgroup per CU and paid the HBM latency in every K tile (64 rows x n = 4608: 127 us alone for 2.7e9 flop).
    d2 ra0[NA], rb0[4], ra1[NA], rb1[4];
    auto load_tile = [&](d2* ra, d2* rb) {
#pragma unroll
        for (int i = 0; i < NA; ++i) { if (a_on[i]) ra[i] = *(const d2*)ap[i]; ap[i] += BK; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { rb[i] = __builtin_nontemporal_load((const d2*)bp[i]); bp[i] += b_step; }
    };
    auto store_tile = [&](int buf, const d2* ra, const d2* rb) {
        double* as = As + buf * BK * LDA;
        double* bs = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < NA; ++i) if (a_on[i]) { as[a_lds[i]] = ra[i][0]; as[a_lds[i] + LDA] = ra[i][1]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (BNF) *(d2*)(bs + b_lds[i]) = rb[i];
            else { bs[b_lds[i]] = rb[i][0]; bs[b_lds[i] + LDB] = rb[i][1]; }
        }
    };
    const int lr = lane & 15, lk = lane >> 4;
    auto compute = [&](int buf) {
        const double* as = As + buf * BK * LDA + lr;
        const double* bs = Bs + buf * BK * LDB + wn * 32 + lr;
#pragma unroll
        for (int k4 = 0; k4 < BK / 4; ++k4) {
            double af[TMW], bf[2];
            const int kr = k4 * 4 + lk;
#pragma unroll
            for (int i = 0; i < TMW; ++i) af[i] = as[kr * LDA + i * 16];
            bf[0] = bs[kr * LDB]; bf[1] = bs[kr * LDB + 16];
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[1], acc[i][1], 0, 0, 0);
            }
        }
    };
    load_tile(ra0, rb0);
    store_tile(0, ra0, rb0);
    if (DEEP) {
        if (nk > 1) load_tile(ra1, rb1);                        // tile 1
        __syncthreads();
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            // even iteration: tile kt in LDS buffer 0, tile kt + 1 in set 1, set 0 free
            if (kt + 2 < nk) load_tile(ra0, rb0);
            compute(0);
            store_tile(1, ra1, rb1);
            __syncthreads();
            // odd iteration: tile kt + 1 in LDS buffer 1, tile kt + 2 in set 0, set 1 free
            if (kt + 3 < nk) load_tile(ra1, rb1);
            compute(1);
            if (kt + 2 < nk) store_tile(0, ra0, rb0);
            __syncthreads();
        }
        if (kt < nk) { compute(0); __syncthreads(); }           // (nk odd: the last tile sits in buffer 0)
    } else {
        // one tile ahead (one register set): with three or four workgroups per CU the other workgroups cover the latency, and the smaller
        // register footprint keeps four workgroups of the <= 32-row instances resident (HBM-bound: 5.6 against 5.2 TB/s at n = 16384)
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_tile(ra0, rb0);
            compute(buf);
            if (kt + 1 < nk) store_tile(buf ^ 1, ra0, rb0);
            __syncthreads();
        }
    }
    if (p.ks == 1) {          // no K split: finished values
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = i * 16 + lk + 4 * r;
                if (m >= p.M) continue;
                double* row = p.C + (long long)m * p.ldc + n0 + wn * 32;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = 16 * j + lr;
                    double v = p.alpha * acc[i][j][r];
                    if (p.colscale) v *= p.colscale[n0 + wn * 32 + c];
                    if (p.beta != 0.0) v += p.beta * row[c];
                    row[c] = v;
                }
            }
        return;
    }
    double* P = p.P + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = i * 16 + lk + 4 * r;
            if (m >= p.M) continue;
            double* row = P + (long long)m * p.N + n0 + wn * 32;
            row[lr] = acc[i][0][r]; row[16 + lr] = acc[i][1][r];
        }
    if (p.ks <= 0) return;               // partials only: a separate reduce kernel follows
    // In-launch combine of the K-slice partials (cdna_hip_programming.md, split-K counter recipe): plain slab stores, every wave
    // drains its stores, one agent-scope release by lane 0, then the ticket; the workgroup that draws ks-1 acquires once and sums
    // the slabs of its column tile in slice order.  Correct for any placement of the slices over XCDs; replaces a reduce launch
    // that re-streamed ks * M * N partials through HBM behind a kernel boundary.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);              // the K loop is over: the tile buffers are free
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(&p.cnt[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(p.ks - 1));
        if (last) {
            __hip_atomic_store(&p.cnt[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    // reducer: thread = (column pair, row group); rows rg, rg + 4, ...; 16-byte loads, four rows of all slices in flight
    const int cp = tid & 63, rg = tid >> 6;
    const long long slab = (long long)p.M * p.N;
    const double* P0 = p.P + n0 + 2 * cp;
    double cs0 = 1.0, cs1 = 1.0;
    if (p.colscale) { cs0 = p.colscale[n0 + 2 * cp]; cs1 = p.colscale[n0 + 2 * cp + 1]; }
    for (int m0 = rg; m0 < p.M; m0 += 16) {
        d2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (d2){0.0, 0.0};
        for (int s = 0; s < p.ks; ++s) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = m0 + 4 * u;
                if (m < p.M) v[u] += *(const d2*)(P0 + s * slab + (long long)m * p.N);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + 4 * u;
            if (m >= p.M) continue;
            double* c = p.C + (long long)m * p.ldc + n0 + 2 * cp;
            d2 w = v[u] * p.alpha;
            w[0] *= cs0; w[1] *= cs1;
            if (p.beta != 0.0) { w[0] += p.beta * c[0]; w[1] += p.beta * c[1]; }
            c[0] = w[0]; c[1] = w[1];
        }
    }
}

template <bool BNF, bool DEEP>
void launch_rows2(int tm, dim3 grid, hipStream_t st, const StripParams& sp) {
    switch (tm) {
        case 1: hipLaunchKernelGGL((gemm_rows_kernel<1, BNF, DEEP>), grid, dim3(256), 0, st, sp); break;
        case 2: hipLaunchKernelGGL((gemm_rows_kernel<2, BNF, DEEP>), grid, dim3(256), 0, st, sp); break;
        case 3: hipLaunchKernelGGL((gemm_rows_kernel<3, BNF, DEEP>), grid, dim3(256), 0, st, sp); break;
        default: hipLaunchKernelGGL((gemm_rows_kernel<4, BNF, DEEP>), grid, dim3(256), 0, st, sp); break;
    }
}
// two tiles ahead when a CU holds at most two workgroups of the launch (mid-size operands), one tile ahead otherwise
template <bool BNF>
void launch_rows(int tm, dim3 grid, hipStream_t st, const StripParams& sp, bool deep) {
    if (deep) launch_rows2<BNF, true>(tm, grid, st, sp); else launch_rows2<BNF, false>(tm, grid, st, sp);
}

template <bool BNF>
void launch_strip(int tm, dim3 grid, hipStream_t st, const StripParams& sp) {
    switch (tm) {
        case 1: hipLaunchKernelGGL((gemm_strip_kernel<1, BNF>), grid, dim3(256), 0, st, sp); break;
        case 2: hipLaunchKernelGGL((gemm_strip_kernel<2, BNF>), grid, dim3(256), 0, st, sp); break;
        case 3: hipLaunchKernelGGL((gemm_strip_kernel<3, BNF>), grid, dim3(256), 0, st, sp); break;
        default: hipLaunchKernelGGL((gemm_strip_kernel<4, BNF>), grid, dim3(256), 0, st, sp); break;
    }
}

__global__ void splitk_reduce_kernel(const double* __restrict__ part, int nsplit, long long stride, double* __restrict__ C, int M, int N,
                                     long long ldc, double alpha, double beta, const double* __restrict__ colscale, int vec,
                                     const int* __restrict__ skip_all = nullptr) {
    if (skip_all && *skip_all == 0) return;
    // sum of the K-slice partials in a fixed order (deterministic).  Two adjacent elements per thread (16-byte loads when N is
    // even) and four slices in flight per step: the partials of a strip GEMM are tens of MB, this kernel is pure HBM streaming.
    const long long tot = (long long)M * N;
    if (vec && (N & 1) == 0 && (stride & 1) == 0) {
        const long long half = tot >> 1;
        for (long long h = (long long)blockIdx.x * blockDim.x + threadIdx.x; h < half; h += (long long)gridDim.x * blockDim.x) {
            const long long q = 2 * h;
            d2 v = (d2){0.0, 0.0};
            int s = 0;
            for (; s + 4 <= nsplit; s += 4) {
                const d2 a = *(const d2*)(part + (long long)s * stride + q), b = *(const d2*)(part + (long long)(s + 1) * stride + q);
                const d2 c = *(const d2*)(part + (long long)(s + 2) * stride + q), d = *(const d2*)(part + (long long)(s + 3) * stride + q);
                v = (((v + a) + b) + c) + d;
            }
            for (; s < nsplit; ++s) v += *(const d2*)(part + (long long)s * stride + q);
            const long long m = q / N, n = q - m * N;
            v *= alpha;
            if (colscale) { v[0] *= colscale[n]; v[1] *= colscale[n + 1]; }
            double* c = C + m * ldc + n;
            if (beta != 0.0) { v[0] += beta * c[0]; v[1] += beta * c[1]; }
            c[0] = v[0]; c[1] = v[1];
        }
        return;
    }
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (long long)gridDim.x * blockDim.x) {
        double v = 0.0;
        for (int s = 0; s < nsplit; ++s) v += part[s * stride + q];
        const long long m = q / N, n = q - m * N;
        v *= alpha;
        if (colscale) v *= colscale[n];
        if (beta != 0.0) v += beta * C[m * ldc + n];
        C[m * ldc + n] = v;
    }
}

}  // namespace

// operands of a plain GEMM the vectorised 128x128 kernel can take (apart from M, N being multiples of 128)
static bool fast_operands(const ctm_ctx* ctx, const GemmDesc& d) {
    const bool ak = d.sak == 1, amf = d.sam == 1, bnf = d.sbn == 1, bkf = d.sbk == 1;
    return ctx->gemm_fast && !d.offs && d.splitB_dim == 0 && d.splitA >= d.M && d.splitC >= d.M && !d.skip_flags && !d.skip_all && d.K % 16 == 0 &&
           (ak || amf) && (bnf || bkf) && (((uintptr_t)d.A | (uintptr_t)d.B) & 15) == 0 && ((ak ? d.sam : d.sak) % 2 == 0) &&
           ((bnf ? d.sbk : d.sbn) % 2 == 0) && (d.strideA % 2 == 0) && (d.strideB % 2 == 0);
}

int gemm_f64(ctm_ctx* ctx, const GemmDesc& d) {
    if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return CTM_OK;
    if (d.K <= 0) { ctx->set_error("gemm: K<=0"); return CTM_ERR_BADARG; }
    if (ctx->gemm_log)
        fprintf(stderr, "GEMM M=%d N=%d K=%d batch=%d sam=%lld sak=%lld sbk=%lld sbn=%lld offs=%d seg=%d\n", d.M, d.N, d.K, d.batch, d.sam, d.sak, d.sbk,
                d.sbn, d.offs ? 1 : 0, (d.splitA < d.M || d.splitC < d.M || d.splitB_dim) ? 1 : 0);
    // A block of chi+1 (or any 128 q + r, r <= 64) vectors times an n x n corner: the 128-row tiling would pay a whole extra
    // tile row for the r remainder rows (M = 257: 5.2 ms instead of 2.3 ms at n = 16384).  Split it into the multiple of
    // 128 (vectorised kernel) and an r-row strip (HBM-streaming split-K path); same for a remainder in N.
    if (ctx->gemm_split_rem && d.batch == 1 && d.K >= 1024 && fast_operands(ctx, d)) {
        const int rm = d.M % 128, rn = d.N % 128;
        if (d.M > 128 && rm > 0 && rm <= 64 && (rn == 0 || (d.N > 128 && rn <= 64)) && (long long)d.N * d.K >= (1ll << 22)) {   // (an N remainder is split off by the recursion)
            GemmDesc a = d, b = d;
            a.M = d.M - rm; a.splitA = a.splitC = a.M;
            b.M = rm; b.A = d.A + (long long)a.M * d.sam; b.C = d.C + (long long)a.M * d.ldc; b.splitA = b.splitC = rm;
            int st = gemm_f64(ctx, a);
            return st != CTM_OK ? st : gemm_f64(ctx, b);
        }
        if (d.N > 128 && rn > 0 && rn <= 64 && rm == 0 && (long long)d.M * d.K >= (1ll << 22)) {
            GemmDesc a = d, b = d;
            a.N = d.N - rn;
            b.N = rn; b.B = d.B + (long long)a.N * d.sbn; b.C = d.C + a.N; if (d.colscale) b.colscale = d.colscale + a.N;
            int st = gemm_f64(ctx, a);
            return st != CTM_OK ? st : gemm_f64(ctx, b);
        }
    }
    ArenaScope split_scope(ctx);      // split-K partials live only until the (stream-ordered) reduce kernel
    // a block of <= 64 k-contiguous rows times a big operand that is read once: the streaming strip kernel
    if (ctx->gemm_strip && d.M <= 64 && d.batch == 1 && !d.offs && d.splitB_dim == 0 && d.splitA >= d.M && d.splitC >= d.M && !d.skip_flags && !d.skip_all &&
        d.sak == 1 && (d.sbn == 1 || d.sbk == 1) && d.K % 16 == 0 && d.N % 32 == 0 && d.K >= 1024 && (long long)d.N * d.K >= (1ll << 22) &&
        (((uintptr_t)d.A | (uintptr_t)d.B) & 15) == 0 && d.sam % 2 == 0 && (d.sbn == 1 ? d.sbk : d.sbn) % 2 == 0) {
        const bool bnf = d.sbn == 1;
        const int gx = (d.N / 32 + 3) / 4;
        // MFMA-bound block sizes go to the LDS-tiled row-block kernel (same partial-sum interface, 128-column workgroups)
        const bool rows_kernel = d.M >= (bnf ? ctx->rows_kernel_min_m : ctx->rows_kernel_min_m_kc) && d.N % 128 == 0;
        // big operands (>= 1 GB): fewer K slices, fewer partials; smaller ones need the extra workgroups to fill the chip
        int target = ((long long)d.N * d.K >= (1ll << 27)) ? ctx->strip_target_wgs : 2 * ctx->strip_target_wgs;
        if (rows_kernel) target = ctx->rows_target_wgs;
        int ks = std::max(1, std::min(std::min((target + gx - 1) / gx, d.K / 256), 64));
        // mid-size operands (n = 4608: 36 column tiles) would run 18 slices of 256 k: no slice shorter than rows_min_klen, but at least
        // two slices (a single slice means N / 128 workgroups: 36 on 256 CUs).  The direct-write epilogue (ks == 1) is reached when the
        // column tiles alone fill the target (N >= 128 * rows_target_wgs) or by option; tests/test_gpu_gemm_rows.py drives it.
        // (<= 32 rows are HBM-bound: what a CU needs there is bytes in flight, i.e. more, shorter workgroups -- rows_min_klen_hbm)
        const int min_klen = d.M <= 32 ? ctx->rows_min_klen_hbm : ctx->rows_min_klen;
        if (rows_kernel && min_klen > 256 && ks > 2) {
            ks = std::max(2, std::min(ks, d.K / min_klen));
            // ... and no partial round of workgroups: 36 column tiles x 8 slices = 288 workgroups on 256 CUs leave 32 CUs with two
            // MFMA-bound workgroups and the others with one (the launch takes as long as 512); 7 slices = 252 fill one round.  Among the
            // slice counts allowed above, take the largest whose workgroup count is at most a multiple of the CU count that it nearly
            // fills (>= 85 % of the last round).
            if (ctx->rows_quantise && gx * ks > 128) {
                int best = ks;
                for (int c = ks; c >= 2; --c) {
                    const int w = gx * c, rounds = (w + 255) / 256;
                    if (w >= rounds * 256 * 0.85) { best = c; break; }
                }
                ks = best;
            }
        }
        int klen = (((d.K + ks - 1) / ks) + 15) / 16 * 16;
        ks = (d.K + klen - 1) / klen;
        // the row-block kernel sums its K slices inside the launch (last workgroup of a column tile); the strip kernel keeps the
        // separate fixed-order reduce kernel
        auto overlaps = [](const double* a, size_t na, const double* b, size_t nb) { return a < b + nb && b < a + na; };
        const size_t spanC = (size_t)(d.M - 1) * d.ldc + d.N, spanA = (size_t)(d.M - 1) * d.sam + d.K;
        const size_t spanB = (size_t)((bnf ? d.K : d.N) - 1) * (bnf ? d.sbk : d.sbn) + (bnf ? d.N : d.K);
        // (the combine writes C while other workgroups of the launch still read A and B: no aliasing)
        const bool fused = rows_kernel && ctx->rows_fused_reduce && d.N / 128 <= CTM_TILE_COUNTERS && (d.ldc % 2 == 0) &&
                           (((uintptr_t)d.C) & 15) == 0 && !overlaps(d.C, spanC, d.A, spanA) && !overlaps(d.C, spanC, d.B, spanB);
        if (fused && !ctx->tile_cnt) {
            // zeroed ON THE CONTEXT'S STREAM and waited for: a plain hipMemset runs on the null stream, which non-blocking streams (torch's)
            // are not ordered with -- the first combine could otherwise start on garbage counters, or be reset half-way
            if (hipMalloc((void**)&ctx->tile_cnt, sizeof(unsigned) * CTM_TILE_COUNTERS) != hipSuccess ||
                hipMemsetAsync(ctx->tile_cnt, 0, sizeof(unsigned) * CTM_TILE_COUNTERS, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->set_error("gemm: tile counters"); return CTM_ERR_NOMEM; }
        }
        double* part = nullptr;
        if (!(fused && ks == 1))
            if (arena_alloc(ctx, sizeof(double) * (size_t)ks * d.M * d.N, (void**)&part) != CTM_OK) return CTM_ERR_NOMEM;
        StripParams sp;
        sp.M = d.M; sp.N = d.N; sp.K = d.K; sp.A = d.A; sp.sam = d.sam; sp.B = d.B; sp.ldb = bnf ? d.sbk : d.sbn; sp.P = part; sp.klen = klen;
        if (fused) { sp.C = d.C; sp.ldc = d.ldc; sp.alpha = d.alpha; sp.beta = d.beta; sp.colscale = d.colscale; sp.cnt = ctx->tile_cnt; sp.ks = ks; }
        else sp.ks = 0;                  // partials only (ks == 1 included): the reduce kernel below finishes
        int e0 = timing_begin(ctx);
        if (g_ctm_sync_launch) ctm_note_launch(rows_kernel ? "gemm_rows_kernel" : "gemm_strip_kernel");
        if (rows_kernel) {
            const bool deep = ctx->rows_deep_prefetch && (long long)(d.N / 128) * ks <= 512;
            if (bnf) launch_rows<true>((d.M + 15) / 16, dim3(d.N / 128, ks), ctx->stream, sp, deep);
            else launch_rows<false>((d.M + 15) / 16, dim3(d.N / 128, ks), ctx->stream, sp, deep);
        }
        else if (bnf) launch_strip<true>((d.M + 15) / 16, dim3(gx, ks), ctx->stream, sp);
        else launch_strip<false>((d.M + 15) / 16, dim3(gx, ks), ctx->stream, sp);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && g_ctm_sync_launch) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->set_error(std::string("strip gemm launch: ") + hipGetErrorString(e)); return CTM_ERR_HIP; }
        const long long tot = (long long)d.M * d.N;
        if (!fused)
            CTM_LAUNCH(ctx, splitk_reduce_kernel, dim3((int)std::min<long long>((tot + 255) / 256, 2048)), dim3(256), 0,
                               (const double*)part, ks, tot, d.C, d.M, d.N, d.ldc, d.alpha, d.beta, d.colscale, ctx->splitk_reduce_vec, (const int*)nullptr);
        const double fl = 2.0 * d.M * d.N * (double)d.K;
        // class 3 (<= 32 rows: HBM-bound) reports algorithmic BYTES, class 4 (33..64 rows: MFMA-bound) flops
        if (d.M <= 32) timing_end(ctx, e0, 3, 8.0 * ((double)d.K * d.N + (double)d.M * d.K + (double)d.M * d.N));
        else timing_end(ctx, e0, 4, fl);
        ctx->gemm_flops += fl;
        ctx->gemm_calls += 1;
        return CTM_OK;
    }
    GemmParams p;
    p.M = d.M; p.N = d.N; p.K = d.K;
    p.A = d.A; p.sam = d.sam; p.sak = d.sak;
    p.B = d.B; p.sbk = d.sbk; p.sbn = d.sbn;
    p.C = d.C; p.ldc = d.ldc;
    p.alpha = d.alpha; p.beta = d.beta;
    p.strideA = d.strideA; p.strideB = d.strideB; p.strideC = d.strideC;
    p.offs = d.offs;
    p.splitA = d.splitA; p.splitB = d.splitB; p.splitC = d.splitC; p.splitB_dim = d.splitB_dim;
    p.colscale = d.colscale;
    p.skip_flags = d.skip_flags;
    p.skip_all = d.skip_all;
    p.ksplit = 1; p.klen = 0; p.split_stride = 0;
    const long long tiles128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch;
    // medium-skinny products (a few hundred rows or columns against a long K): the vectorised kernel with K split so that the
    // workgroup count is a multiple of the 256 CUs (384 tiles run as long as 512; 128 tiles leave half the chip idle).  The
    // K split is expressed as a batch over K slices (strides klen*sak, klen*sbk) writing partial products.
    int fast_ks = 0;
    if (d.batch == 1 && d.M % 128 == 0 && d.N % 128 == 0 && tiles128 >= 32 && tiles128 < 1024 && d.K >= 2048 && fast_operands(ctx, d)) {
        double best = 1e30;
        for (int ks : {1, 2, 3, 4, 6, 8}) {
            if (d.K % (16 * ks) != 0 || d.K / ks < 1024) continue;
            const double cost = (double)((tiles128 * ks + 255) / 256) / ks * (ks > 1 ? 1.05 : 1.0);   // 5 %: partials + reduce pass
            if (cost < best - 1e-9) { best = cost; fast_ks = ks; }
        }
    }
    const bool small = !fast_ks && (d.M <= 64 || d.N <= 64 || tiles128 < 200);
    const int BM = small ? 64 : 128, BN = small ? 64 : 128;
    p.tilesM = (d.M + BM - 1) / BM;
    p.tilesN = (d.N + BN - 1) / BN;
    dim3 grid((unsigned)(p.tilesM * p.tilesN), 1, (unsigned)d.batch);
    // split-K for skinny outputs with a long K (few tiles cannot fill 256 CUs): partial products into the arena,
    // summed in a fixed order by a second tiny kernel (deterministic, no atomics)
    double* part = nullptr;
    const int ntile = p.tilesM * p.tilesN;
    // very skinny outputs (one 64-row or 64-column strip, e.g. a 64-vector block times an n x n corner) stream the big operand
    // once and are HBM-latency bound with one workgroup per CU: aim for ~4 workgroups per CU there, ~2 otherwise
    const bool strip = (d.M <= 64 || d.N <= 64);
    const int max_tiles = strip ? ctx->splitk_max_tiles : 127, target = strip ? ctx->splitk_target_wgs : 512;
    if (fast_ks > 1) {
        if (arena_alloc(ctx, sizeof(double) * (size_t)fast_ks * d.M * d.N, (void**)&part) != CTM_OK) return CTM_ERR_NOMEM;
        p.K = d.K / fast_ks;
        p.strideA = (long long)p.K * d.sak; p.strideB = (long long)p.K * d.sbk; p.strideC = (long long)d.M * d.N;
        p.split_stride = p.strideC;
        p.C = part; p.ldc = d.N; p.alpha = 1.0; p.beta = 0.0; p.colscale = nullptr;
        grid.z = (unsigned)fast_ks;
    } else if (d.batch == 1 && !d.offs && d.splitB_dim == 0 && d.splitA >= d.M && d.splitC >= d.M && ntile <= max_tiles && d.K >= 1024) {
        // partial products cost ks*M*N doubles: tiny outputs (reductions over a huge K) may split much further
        const int ks_cap = ((long long)d.M * d.N <= 4096) ? 256 : 16;
        int ks = std::min(ks_cap, std::min(target / std::max(ntile, 1), d.K / 256));
        if (ks > 1) {
            p.klen = (((d.K + ks - 1) / ks) + 15) / 16 * 16;
            ks = (d.K + p.klen - 1) / p.klen;
        }
        if (ks > 1) {
            if (arena_alloc(ctx, sizeof(double) * (size_t)ks * d.M * d.N, (void**)&part) != CTM_OK) return CTM_ERR_NOMEM;
            p.ksplit = ks; p.split_stride = (long long)d.M * d.N;
            p.C = part; p.ldc = d.N; p.alpha = 1.0; p.beta = 0.0; p.colscale = nullptr;
            grid.z = (unsigned)ks;
        }
    }
    int e0 = -1, e1 = -1;
    // event pair only around launches that matter for a roofline (>= timing_min_flops): the 64-tile products of the latency-bound
    // stages are ~90 % of all launches and ~0 % of the flops, and an event pair per launch costs the concurrent units ~10 % of a sweep
    if (ctx->gemm_timing && 2.0 * d.M * d.N * (double)d.K * d.batch >= ctx->timing_min_flops) {
        if (ctx->ev_next + 2 > (int)ctx->ev_pool.size()) {
            if (ctx->ev_pool.size() >= 8192) gemm_timing_drain(ctx);
            else { for (int i = 0; i < 512; ++i) { hipEvent_t e; (void)hipEventCreate(&e); ctx->ev_pool.push_back(e); } }
        }
        e0 = ctx->ev_next++; e1 = ctx->ev_next++;
        (void)hipEventRecord(ctx->ev_pool[e0], ctx->stream);
    }
    const bool ak = d.sak == 1, amf = d.sam == 1, bnf = d.sbn == 1, bkf = d.sbk == 1;
    const bool fast = fast_ks > 0 || (!small && !part && d.M % 128 == 0 && d.N % 128 == 0 && fast_operands(ctx, d));
    (void)amf; (void)bkf;
    if (fast) {
        if (ak && bnf) CTM_LAUNCH(ctx, (gemm_f64_fast_kernel<true, true>), grid, dim3(256), 0, p);
        else if (ak && !bnf) CTM_LAUNCH(ctx, (gemm_f64_fast_kernel<true, false>), grid, dim3(256), 0, p);
        else if (!ak && bnf) CTM_LAUNCH(ctx, (gemm_f64_fast_kernel<false, true>), grid, dim3(256), 0, p);
        else CTM_LAUNCH(ctx, (gemm_f64_fast_kernel<false, false>), grid, dim3(256), 0, p);
    } else if (small)
        CTM_LAUNCH(ctx, (gemm_f64_kernel<2, 2>), grid, dim3(256), 0, p);
    else
        CTM_LAUNCH(ctx, (gemm_f64_kernel<4, 4>), grid, dim3(256), 0, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ctx->set_error(std::string("gemm launch: ") + hipGetErrorString(e)); return CTM_ERR_HIP; }
    if (part) {
        const long long tot = (long long)d.M * d.N;
        int blocks = (int)std::min<long long>((tot + 255) / 256, 2048);
        CTM_LAUNCH(ctx, splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (const double*)part, fast_ks > 1 ? fast_ks : p.ksplit, p.split_stride, d.C,
                           d.M, d.N, d.ldc, d.alpha, d.beta, d.colscale, ctx->splitk_reduce_vec, d.skip_all);
    }
    const double fl = 2.0 * d.M * d.N * (double)d.K * d.batch;
    if (e1 >= 0) {
        (void)hipEventRecord(ctx->ev_pool[e1], ctx->stream);
        ctx->ev_pending.push_back({e0, e1, small ? 1 : 0, fl});
    }
    ctx->gemm_flops += fl;
    ctx->gemm_calls += 1;
    return CTM_OK;
}


int timing_begin(ctm_ctx* ctx) {
    if (!ctx->gemm_timing) return -1;
    if (ctx->ev_next + 2 > (int)ctx->ev_pool.size()) {
        if (ctx->ev_pool.size() >= 8192) gemm_timing_drain(ctx);
        else { for (int i = 0; i < 512; ++i) { hipEvent_t e; (void)hipEventCreate(&e); ctx->ev_pool.push_back(e); } }
    }
    const int e0 = ctx->ev_next; ctx->ev_next += 2;
    (void)hipEventRecord(ctx->ev_pool[e0], ctx->stream);
    return e0;
}

void timing_end(ctm_ctx* ctx, int e0, int kind, double flops) {
    if (e0 < 0) return;
    (void)hipEventRecord(ctx->ev_pool[e0 + 1], ctx->stream);
    ctx->ev_pending.push_back({e0, e0 + 1, kind, flops});
}

// [-Ai; Ar]: the planar complex block times i (see xgemm)
__global__ void planar_times_i_kernel(const double* __restrict__ re, const double* __restrict__ im, double* __restrict__ out, size_t tot) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) { out[q] = -im[q]; out[tot + q] = re[q]; }
}

int xgemm(ctm_ctx* ctx, int M, int N, int K, const XM& A, const XM& B, double* Cre, double* Cim, long long ldc, const double* colscale) {
    GemmDesc g;
    g.M = M; g.N = N; g.K = K;
    if (A.t) { g.sam = 1; g.sak = A.ld; } else { g.sam = A.ld; g.sak = 1; }
    if (B.t) { g.sbk = 1; g.sbn = B.ld; } else { g.sbk = B.ld; g.sbn = 1; }
    g.ldc = ldc; g.colscale = colscale;
    if (!A.im && !B.im) {
        g.A = A.re; g.B = B.re; g.C = Cre;
        return gemm_f64(ctx, g);
    }
    if (!A.im || !B.im || !Cim) { ctx->set_error("xgemm: mixed real/complex operands"); return CTM_ERR_BADARG; }
    const double sA = A.c ? -1.0 : 1.0, sB = B.c ? -1.0 : 1.0;
    // A row block (<= 64 complex rows) whose planes lie one behind the other IS the 2M-row real matrix S = [Ar; Ai], and so is the output:
    //   [Cr; Ci] = S op(Br) + sB [-Ai; Ar] op(Bi)
    // -- two real products with twice the rows instead of four: every plane of the big operand is streamed once, not twice, and 64
    // complex rows run on the 128-row tile (0.75-0.79 of the MFMA peak) instead of the 64-row row-block kernel (0.70).  This is every
    // corner pass of the complex block iterations (matop_apply_planar: 8 per Krylov step, 9.7 GB of corner planes each at n = 24576).
    if (ctx->xgemm_stack_rows && !A.t && !A.c && M <= 64 && M % 16 == 0 && A.ld == K /* whole rows are copied: no padding behind the last one */ &&
        A.im == A.re + (size_t)M * A.ld && Cim == Cre + (size_t)M * ldc &&
        (long long)N * K >= (1ll << 22)) {
        ArenaScope scope(ctx);
        double* S2;
        CTM_TRY(arena_alloc(ctx, sizeof(double) * 2 * (size_t)M * A.ld, (void**)&S2));
        const size_t tot = (size_t)M * A.ld;
        CTM_LAUNCH(ctx, planar_times_i_kernel, dim3((int)std::min<size_t>((tot + 255) / 256, 2048)), dim3(256), 0, A.re, A.im, S2, tot);
        g.M = 2 * M;
        g.A = A.re; g.B = B.re; g.C = Cre; g.alpha = 1.0; g.beta = 0.0;   CTM_TRY(gemm_f64(ctx, g));
        g.A = S2;   g.B = B.im; g.C = Cre; g.alpha = sB;  g.beta = 1.0;   return gemm_f64(ctx, g);
    }
    g.A = A.re; g.B = B.re; g.C = Cre; g.alpha = 1.0; g.beta = 0.0;        CTM_TRY(gemm_f64(ctx, g));
    g.A = A.im; g.B = B.im; g.C = Cre; g.alpha = -sA * sB; g.beta = 1.0;   CTM_TRY(gemm_f64(ctx, g));
    g.A = A.re; g.B = B.im; g.C = Cim; g.alpha = sB; g.beta = 0.0;         CTM_TRY(gemm_f64(ctx, g));
    g.A = A.im; g.B = B.re; g.C = Cim; g.alpha = sA; g.beta = 1.0;         return gemm_f64(ctx, g);
}

// process-wide time base of the GEMM intervals: contexts on different streams (concurrent units) report [start, end] on
// one clock so that the host can take the union of overlapping launches
static hipEvent_t g_base_event = nullptr;
static std::mutex g_base_mutex;
void gemm_timing_base(ctm_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_base_mutex);
    if (g_base_event) return;
    if (hipEventCreate(&g_base_event) != hipSuccess) { g_base_event = nullptr; return; }
    (void)hipEventRecord(g_base_event, ctx->stream);
    (void)hipEventSynchronize(g_base_event);
}

void gemm_timing_drain(ctm_ctx* ctx) {
    if (ctx->ev_pending.empty()) { ctx->ev_next = 0; return; }
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& pe : ctx->ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev_pool[pe.e0], ctx->ev_pool[pe.e1]) == hipSuccess) {
            if (pe.kind >= CTM_KIND_PHASE0) { ctx->timers[pe.kind - CTM_KIND_PHASE0] += 1e-3 * ms; continue; }
            ctx->k_ms[pe.kind] += ms; ctx->k_flops[pe.kind] += pe.flops; ctx->k_calls[pe.kind] += 1;
            float t0 = 0.f;
            if (g_base_event && hipEventElapsedTime(&t0, g_base_event, ctx->ev_pool[pe.e0]) == hipSuccess && ctx->intervals.size() < (1u << 22)) {
                ctx->intervals.push_back((double)pe.kind); ctx->intervals.push_back((double)t0);
                ctx->intervals.push_back((double)t0 + ms); ctx->intervals.push_back(pe.flops);
            }
        }
    }
    ctx->ev_pending.clear();
    ctx->ev_next = 0;
}
