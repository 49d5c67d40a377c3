/* ctm_hip.h -- C-ABI of libctm_hip.so: the MI355X-native CTMRG + RDM engine.
 *
 * Drop-in boundary.  The reference (jurajHasik/peps-torch) has no FFI; the seam this ABI
 * replaces is its family of "raw tensor tuple in, raw tensor tuple out" closures (the ones it
 * wraps in torch.utils.checkpoint / offloads to a GPU), each cited below with the reference
 * file:line (relative to the reference repo root).  All pointers are DEVICE pointers to
 * C-contiguous (row-major, as torch) float64 buffers unless stated otherwise; nothing is
 * retained after a call returns.  In a CTM_C128 context every tensor pointer (declared double* below)
 * addresses interleaved (re,im) complex128 elements exactly as torch stores them, all sizes/dims count
 * complex elements, and singular values / eigenvalues / spectra / S outputs stay real (ctm_svd_symeig is float64 only:
 * CTM_ERR_UNSUPPORTED in a CTM_C128 context).  Every function returns an int status (CTM_OK == 0) and
 * records a message retrievable with ctm_last_error().  One ctm_ctx is used by one host
 * thread at a time; work is enqueued on the context's HIP stream and the call returns after
 * enqueueing unless it needs a host decision (truncation logic), in which case it syncs.
 *
 * Tensor conventions (reference ctm/generic/env.py:57-76, ipeps/ipeps.py:117-124):
 *   site a[p,u,l,d,r];  C(-1,-1)=(down,right) C(1,-1)=(left,down) C(1,1)=(up,left) C(-1,1)=(up,right)
 *   T(0,-1)=(chi,D^2,chi) T(-1,0)=(chi,chi,D^2) T(0,1)=(D^2,chi,chi) T(1,0)=(chi,D^2,chi)
 *   fused D^2 leg = (ket,bra), ket first.
 */
#ifndef CTM_HIP_H
#define CTM_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ctm_ctx ctm_ctx;

enum { CTM_OK = 0, CTM_ERR_BADARG = 1, CTM_ERR_SHAPE = 2, CTM_ERR_NOCONV = 3, CTM_ERR_HIP = 4,
       CTM_ERR_UNSUPPORTED = 5, CTM_ERR_NOMEM = 6,
       CTM_ERR_BUSY = 7 /* the context is inside a call of another thread: one call at a time per context */ };
enum { CTM_F64 = 0, CTM_C128 = 1 };
enum { CTM_LU = 0, CTM_RU = 1, CTM_RD = 2, CTM_LD = 3 };          /* enlarged corners          */
enum { CTM_UP = 0, CTM_LEFT = 1, CTM_DOWN = 2, CTM_RIGHT = 3 };   /* directional moves         */

/* truncation / projector options: reference config.py:370-409 (CTMARGS) */
typedef struct ctm_trunc_cfg {
    double svd_reltol;        /* projector_svd_reltol      (1e-8)  ctm_projectors.py:266            */
    double eps_multiplet;     /* projector_eps_multiplet   (1e-8)  custom_svd.py:70-95              */
    double multiplet_abstol;  /* projector_multiplet_abstol(1e-14)                                  */
    int keep_multiplets;      /* 1                                                                  */
    int fix_signs;            /* 1: apply fix_svd_signs (svd_gesdd.py:18-26)                        */
} ctm_trunc_cfg;

/* ---- context ------------------------------------------------------------------------------ */
/* Diagnostic: with CTM_ABORT_BACKTRACE=1 in the environment the first ctm_create installs a SIGABRT / SIGSEGV handler that writes the
 * native stack of the failing thread to fd 2 and then hands the signal to its previous owner.  Off by default (a library should not
 * take over a host application's signals); the test-suite turns it on. */
int ctm_create(ctm_ctx** out, void* hip_stream /* hipStream_t or NULL */, int dtype);
int ctm_destroy(ctm_ctx* ctx);
const char* ctm_last_error(ctm_ctx* ctx);
const char* ctm_version(void);
int ctm_sync(ctm_ctx* ctx);
int ctm_trim(ctm_ctx* ctx);   /* release the context's workspace arena (regrown on demand); call between engine calls */
int ctm_set_option(ctm_ctx* ctx, const char* key, double value);
/*   truncation:  "jacobi_tol","jacobi_max_sweeps","jacobi_block","jacobi_inner_sweeps","jacobi_inner_sweeps_many","jacobi_verbose","eig64_pingpong",
 *                "si_enable","si_min_n","si_max_iter","si_tol","si_rr_sweeps","si_warm_skip_calls","si_block32","rank_tol","lz_enable","lz_min_k","lz_switch_steps"
 *   kernels:     "use_layer2","layer2_reg","layer2_cplx","gemm_fast","gemm_strip","strip_target_wgs","gemm_split_rem",
 *                "splitk_max_tiles","splitk_target_wgs","einsum_in_relayout","z_spectators_first","chain_as_strips","gemm_log"
 *   round 4:     "lz_block" (rows per block of the real block Krylov recurrence: 32 / 64), "lz_block_c" (complex), "xgemm_stack_rows",
 *                "rows_min_klen","rows_min_klen_hbm","rows_target_wgs","rows_quantise","rows_deep_prefetch" (K-slice rule / pipeline of the row-block GEMM),
 *                "jacobi_cross_only","jacobi_rot_apply","jacobi_persist" (variants of the many-panel block Jacobi), "eigh_orth_extra_blocks",
 *                "eigh_orth_double","eigh_orth_double_min_ratio" (two shifted applications per Cholesky-QR step of the symmetric orthogonal iteration)
 *   measurement: "gemm_timing","timing_min_flops","profile"                                                                   */
int ctm_get_stat(ctm_ctx* ctx, const char* key, double* value);
/*   "last_sweeps","last_offnorm","total_sweeps","jacobi_calls","si_hits","si_fallbacks","si_total_iters","si_last_iters",
 *   "si_last_rank","si_warm_starts","si_warm_skips","corner_cache_hits","lz_hits","lz_total_steps","gemm_flops","gemm_calls","layer2_flops","layer2_calls",
 *   "arena_high","k_ms0..3","k_flops0..3","k_calls0..3","eigh_warm_hits","eigh_orth_hits","eigh_orth_fails","eigh_orth_doubled"                                                                 */
int ctm_timers(ctm_ctx* ctx, double* out8, int reset);             /* corners,halves,svd,proj,absorb,norm,rdm,eig (s) */
/* GEMM launches timed with HIP events while the option "gemm_timing" is on: quadruples (kind, start_ms, end_ms, flops) on a
 * process-wide clock (kind 0 = 128x128-tile GEMM kernels, 1 = 64x64-tile GEMM kernel, 2 = fused two-layer kernel, 3 = streaming
 * row-block kernels with <= 32 rows, whose fourth value is the ALGORITHMIC BYTES instead of flops: HBM-bound; 4 = the same kernels
 * with 33..64 rows: MFMA-bound, flops).  out may be NULL to query
 * *count (launches).  The same classes index the stats "k_ms<c>", "k_flops<c>", "k_calls<c>". */
int ctm_gemm_intervals(ctm_ctx* ctx, double* out, long long capacity_doubles, long long* count);

/* ---- communicator of a rank group that shares ONE unit (multi-GPU: more ranks than sites; reference independence argument
 * ctm/generic/ctmrg.py:238-275 -- the per-site units of a move are independent, so 8 GPUs on a 4-site cell can only be used by
 * splitting a unit) -------------------------------------------------------------------------------------------------------------
 * rccl_comm: an ncclComm_t (RCCL over xGMI) spanning the `nranks` ranks of the group, `rank` = this process's rank in it; NULL with
 * nranks == 1 detaches.  The context issues its collectives on ITS OWN stream (ctm_create), so they are stream-ordered with the
 * kernels that produce and consume the buffers; nothing is retained beyond the handle.
 * Contract of the column split a group executes (ABI frozen here, nranks == 1 is what is implemented):
 *   - every rank of the group keeps the column block [rank n/nranks, (rank+1) n/nranks) of each n x n enlarged corner of the unit
 *     (corner construction, corner cache and absorb split by OUTPUT columns: no communication);
 *   - every corner is applied in both orientations (M = cB^T cA^T cC cD and M^T), so a corner pass is one of two kinds:
 *       B(<= 64 x n) * corner[:, J]      yields this rank's columns J of the product: ONE in-place ncclAllGather of (rows x n/nranks)
 *                                        doubles per rank before the next pass;
 *       B[:, J] * (corner[:, J])^T       yields a partial sum of the whole (rows x n) product: ONE ncclAllReduce (sum) of it;
 *     complex128: both planes in the same call; all on the context's stream -- 8 passes per block step of the Krylov recurrence
 *     (4 of each kind), ~170 per full-rank unit at n = 16384; every rank of the group ends each pass with the same bits;
 *   - the 64-row orthonormalisations and the Ritz extraction are replicated on every rank of the group (bit-identical inputs,
 *     deterministic kernels: no broadcast of their results);
 *   - projectors P, Pt come out replicated; the host layer exchanges them between groups as it does between single ranks
 *     (parallel.exchange).
 * What this build executes (round 6) is the first kind only, on REPLICATED corners: every rank of the group builds and keeps the whole
 * corners (no memory saving, corner construction and absorb not split) and computes, in every corner pass of a float64 unit, the column
 * block [rank n'/nranks, (rank+1) n'/nranks) of the (rows x n') product, followed by one all-gather of (rows x n'/nranks) doubles per rank;
 * every rank ends each pass with the same bits, everything else of the unit is replicated.  complex128 units are not split (every rank
 * computes them in full).  nranks is 1 or 2.
 * ctm_set_comm: the all-gathers are ncclAllGather calls on the context's stream (librccl is resolved at run time with dlopen: the library
 * does not link it) -- CTM_ERR_UNSUPPORTED when it cannot be resolved.  Exercised with a ONE-rank communicator only (no multi-GPU box in
 * the build loop): with nranks == 1 and a communicator the passes go through the same code (one part, all-gather onto itself).
 * ctm_set_comm_ops: the same split with HOST-DRIVEN all-gathers -- the test harness of the build loop (two gloo ranks sharing one device,
 * tests/test_gpu_dist.py) and any transport that is not RCCL: the context writes `count` doubles into send_buf (device, caller-owned,
 * capacity_doubles each for send_buf and recv_buf / nranks), drains its stream and calls allgather(user, count); on return recv_buf must hold
 * the nranks contributions back to back (rank-major), visible to the context's stream.  allgather == NULL detaches.
 * CTM_ERR_BADARG for rank outside [0, nranks) or nranks > 2. */
int ctm_set_comm(ctm_ctx* ctx, void* rccl_comm /* ncclComm_t or NULL */, int rank, int nranks);
typedef int (*ctm_allgather_fn)(void* user, long long count);
int ctm_set_comm_ops(ctm_ctx* ctx, ctm_allgather_fn allgather, void* user, double* send_buf, double* recv_buf, long long capacity_doubles,
                     int rank, int nranks);

/* ---- primitives (replace tn_interface.py:3-27 contract/mm/permute) ------------------------------ */
/* C[M,N] = alpha op(A) op(B) + beta C ; row-major; trans = 0 ('N') or 1 ('T', plain transpose); CTM_C128 also 2 ('C', conjugate
 * transpose), dense operands and beta == 0 */
int ctm_gemm(ctm_ctx* ctx, int transA, int transB, int M, int N, int K, double alpha, const double* A, long long lda,
             const double* B, long long ldb, double beta, double* C, long long ldc);
int ctm_permute(ctm_ctx* ctx, const double* in, double* out, int nd, const long long* dims, const int* perm);
/* Generic tensor-network contraction "i0,i1,...->o" (single-letter indices, evaluated left to right; a consecutive (a, conj a)
 * pair of identical 5-index site operands goes through the fused two-layer kernel): the engine behind tn_interface.contract /
 * einsum (tn_interface.py:3-27), used by the host layer for the observables outside the fixed CTM networks (e.g. the
 * transfer-matrix correlators of ctm/generic/corrf.py).  dims holds the extents of all operands back to back; conj[i] != 0
 * reads operand i conjugated (CTM_C128).  out has the extents of the output indices. */
int ctm_einsum(ctm_ctx* ctx, const char* expr, int ntensors, const double* const* tensors, const int* ndims, const long long* dims,
               const int* conj, double* out);
/* x /= max|x| (ord_inf=1, ctmrg.py:210-230) ; scale_out (device or NULL) receives the norm */
int ctm_normalize_inf(ctm_ctx* ctx, double* x, long long n);

/* ---- chi-truncation (linalg/custom_svd.py:38-101, svd_gesdd.py:77-96; custom_eig.py:7-65, eig_sym.py:25-34) */
/* M is n x n; U,V are n x chi (row-major), S is chi.  Columns beyond the last complete multiplet are zeroed. */
int ctm_truncated_svd(ctm_ctx* ctx, const double* M, int n, int chi, const ctm_trunc_cfg* cfg, double* U, double* S,
                      double* V);
/* Same, warm started from the decomposition of a nearby matrix: `basis` as for ctm_projectors_4x4_ws ((min(chi+1,n) + 1) * n
 * doubles, CTM_C128 (2 min(chi+1,n) + 1) * n; zero-filled before the first call, passed again for the next matrix of the
 * sequence).  The result does not depend on the basis (residual-verified, and a full block started from it is not accepted
 * before its guard rows have seen the operator three times); only the work does.
 * chi >= n (FULL decomposition, the SVD node of the differentiable route): the workspace then keeps the n left vectors; the row
 * Jacobi of the next call starts from W M (rows almost orthogonal when the matrix moved little) instead of M. */
int ctm_truncated_svd_ws(ctm_ctx* ctx, const double* M, int n, int chi, const ctm_trunc_cfg* cfg, double* U, double* S,
                         double* V, double* basis);
/* A symmetric (lower triangle referenced); D (chi, signed, ordered by |D| descending), U n x chi */
int ctm_truncated_eigh(ctm_ctx* ctx, const double* A, int n, int chi, const ctm_trunc_cfg* cfg, double* D, double* U);
/* Same for a sequence of nearby matrices (the enlarged corner of consecutive C4v moves): `basis` is an opaque caller-owned device
 * workspace of (min(n, min(chi+1,n) + 8) + 1) * n doubles (CTM_C128: (2 min(...) + 1) * n, real plane then imaginary plane), zero-filled
 * before the first call: the vectors, then ONE header row of n doubles in which the solver keeps the adaptive state of this sequence
 * (contraction rate of its last accepted solve, back-off counters) -- the state is born and dies with the workspace.
 * A restart from the previous invariant subspace is accepted only if (a) every pair of a Rayleigh-Ritz inside it passes the
 * residual threshold of the cold solver and (b) a block of fresh pseudo-random rows iterated three times on the deflated matrix finds
 * nothing above the smallest accepted |lambda|; otherwise the regular iteration runs.  The result does not depend on the basis.
 * chi >= n (FULL decomposition, the SYMEIG node of the differentiable route): the workspace ((n + 1) * n doubles) keeps all eigenvectors
 * and the Jacobi sweeps of the next call start from W (A + shift I). */
int ctm_truncated_eigh_ws(ctm_ctx* ctx, const double* A, int n, int chi, const ctm_trunc_cfg* cfg, double* D, double* U,
                          double* basis);
/* SVD of a real SYMMETRIC matrix through its eigendecomposition (linalg/svd_symeig.py:12-34 SVDSYMEIG.forward and
 * linalg/custom_svd.py:143-208 truncated_svd_symeig): A = U D U^T ordered by |D| descending, S = |D|, V = U sign(D);
 * U, V n x min(chi,n), S min(chi,n); chi = n gives the full decomposition; cfg NULL: no multiplet back-off.  CTM_F64 only. */
int ctm_svd_symeig(ctm_ctx* ctx, const double* A, int n, int chi, const ctm_trunc_cfg* cfg, double* U, double* S, double* V);
/* singular values of an n x n matrix, descending (ENV.get_spectra, env.py:204-209) */
int ctm_svdvals(ctm_ctx* ctx, const double* M, int n, double* S);

/* ---- adjoints of the two decompositions (the adjoint of a whole move is assembled by the host layer from these and from ctm_einsum
 * nodes: peps-torch_amd/linalg/native_einsum.py, ctm/generic/ctm_ad.py) ------ */
/* SVDGESDD.backward (linalg/svd_gesdd.py:209-328): A = U diag(S) V^H with U m x k, V n x k (k <= min(m,n); thin factors get the
 * (1 - U U^H), (1 - V V^H) terms exactly as the reference), gradients gU, gS, gV (any may be NULL), regularisation eps (the
 * reference's ad_decomp_reg, relative to S[0]); dA is m x n.  CTM_C128: complex factors / gradients, S and gS real. */
int ctm_svd_backward(ctm_ctx* ctx, const double* U, const double* S, const double* V, const double* gU, const double* gS,
                     const double* gV, int m, int n, int k, double eps, double* dA);
/* SYMEIG.backward (linalg/eig_sym.py:57-75): A = U diag(D) U^H, U n x k; dA = U (diag(gD) + F o (U^H gU)) U^H with
 * F_ij = x / (x^2 + reg), x = D_j - D_i; gD, gU may be NULL. */
int ctm_eigh_backward(ctm_ctx* ctx, const double* D, const double* U, const double* gD, const double* gU, int n, int k, double reg,
                      double* dA);

/* ---- generic directional move units ---------------------------------------------------------------- */
/* c2x2_{LU,RU,RD,LD}_sl_c (ctm_components.py:372-434,532-586,683-733,832-884).
 * adims = {p,Du,Dl,Dd,Dr}.  out: (chi*Dx^2) x (chi*Dy^2) [x p x p if open]. */
int ctm_c2x2(ctm_ctx* ctx, int corner, int open, const double* C, const double* T1, const double* T2, const double* a,
             int chi, const int* adims, double* out);
/* halves_of_4x4_CTM_MOVE_<DIR>_c (ctm_components.py:52-75,117-139,180-201,244-265): 16 tensors in the
 * reference's order (C,T1,T2,a) x 4 corners; adims 4x5; R, Rt are n x n. */
int ctm_halves(ctm_ctx* ctx, int dir, const double* const* tensors16, int chi, const int* adims4x5, double* R,
               double* Rt);
/* ctm_get_projectors_from_matrices (ctm_projectors.py:142-293): P, Pt n x chi, S chi (may be NULL) */
int ctm_projectors(ctm_ctx* ctx, const double* R, const double* Rt, int n, int chi, const ctm_trunc_cfg* cfg, double* P,
                   double* Pt, double* S);
/* The same for RECTANGULAR halves R, Rt (a x b, row-major): bond dimensions that differ along one cut -- the reference only asserts
 * R.shape == Rt.shape (ctm_projectors.py:209).  a = the truncated bond (rows of P, Pt), b = the far side: M = R^T Rt is b x b, P and Pt
 * are a x min(chi, b), S holds min(chi, b) values.  The fused entries below take square halves (a == b) only and return
 * CTM_ERR_UNSUPPORTED otherwise; the host layer then builds the halves (ctm_halves: any shapes) and calls this. */
int ctm_projectors_rect(ctm_ctx* ctx, const double* R, const double* Rt, int a, int b, int chi, const ctm_trunc_cfg* cfg, double* P,
                        double* Pt, double* S);
/* ctm_get_projectors_4x4 (ctm_projectors.py:14-64) as ONE call: the four enlarged corners of the move are built and
 * M = R^T Rt is applied implicitly (R, Rt, M are only materialised if the leading-chi iteration falls back to the full
 * decomposition).  tensors16 / adims4x5 as for ctm_halves. */
int ctm_projectors_4x4(ctm_ctx* ctx, int dir, const double* const* tensors16, int chi, const int* adims4x5,
                       const ctm_trunc_cfg* cfg, double* P, double* Pt, double* S);
/* Same, warm started: `basis` is an opaque caller-owned device workspace of (min(chi+1,n) + 1) * n doubles (CTM_C128:
 * (2 min(chi+1,n) + 1) * n; the last row is a header the engine keeps its start-up policy in), zero-filled before the first call and passed again for the same (direction, site) on later sweeps.
 * Optional Ritz region: a caller that allocates R more doubles BEHIND the header row writes R into header word 10 (after the zero fill); the block
 * Krylov solver then keeps the accumulated rotations of the unit's last Ritz extraction there (16 words + m x m doubles, m ~ 2.7 chi rows of
 * Krylov basis; R >= 16 + (4.5 (chi+1) + 64)^2 covers it) and starts the next extraction from them.  Word 10 == 0: no region, every extraction starts cold.  It carries
 * the right singular row basis of the previous call: the leading-chi iteration starts from it instead of a random block
 * (the result is residual-verified either way, so a stale or zero basis only costs iterations).  basis == NULL: cold. */
int ctm_projectors_4x4_ws(ctm_ctx* ctx, int dir, const double* const* tensors16, int chi, const int* adims4x5,
                          const ctm_trunc_cfg* cfg, double* P, double* Pt, double* S, double* basis);
/* Same with caller-kept enlarged corners.  corner_buf: NULL, or four device buffers (NULL entries allowed) of n0*n1 doubles
 * (twice that for CTM_C128; 16-byte aligned; engine-internal layout, opaque) for the four corners in tensors16 order;
 * corner_valid[i] != 0: buffer i holds that corner as computed by an earlier call from the SAME (C, T1, T2, a) -- it is used
 * as it is; otherwise the corner is computed into the buffer.  An enlarged corner only depends on its own corner matrix and
 * two T tensors, and a directional move replaces two corner matrices and one T per site (ctmrg.py:302-319): the corners of
 * the opposite side stay valid, i.e. half of the reference's corner contractions (ctm_components.py:10-265 rebuilds all
 * four in every move) are repeats.  The host layer keeps the buffers with the environment and checks tensor identity. */
int ctm_projectors_4x4_cc(ctm_ctx* ctx, int dir, const double* const* tensors16, int chi, const int* adims4x5,
                          const ctm_trunc_cfg* cfg, double* P, double* Pt, double* S, double* basis,
                          double* const* corner_buf, const int* corner_valid);
/* absorb_truncate_CTM_MOVE_<DIR>_c (ctmrg.py:343-438,459-564,585-680,701-804), 'sl' mode, followed by
 * move_normalize_c (ctmrg.py:210-230): normalize = 0 none, 1 'inf' (max-abs), 2 vector 2-norm.
 * tensors10 = C1,T1,T,T2,C2,A,P2,Pt2,P1,Pt1 */
int ctm_absorb(ctm_ctx* ctx, int dir, const double* const* tensors10, int chi, const int* adims, int normalize,
               double* nC1, double* nC2, double* nT);
/* Same with different dimensions of the incoming environment (chi_in: C, T and the row blocks of the projectors) and of the
 * truncated bond (chi_out: projector columns).  nT is (chi_out, D^2, chi_out)-shaped per direction; nC1 and nC2 have one new and
 * one old leg: new leg first for nC1 of UP/LEFT/RIGHT and nC2 of LEFT, second otherwise (the output index order of the
 * reference's contractions, ctmrg.py:351-374 etc.).  Used by the host layer to run a move on the non-zero block of an
 * environment whose trailing singular values were masked (chi_eff < chi). */
int ctm_absorb_x(ctm_ctx* ctx, int dir, const double* const* tensors10, int chi_in, int chi_out, const int* adims, int normalize,
                 double* nC1, double* nC2, double* nT);

/* ---- a whole directional move of the generic CTMRG in ONE call (ctm/generic/ctmrg.py:233-283 ctm_MOVE_c: raw tensors of the old
 * environment in, nC1 / nC2 / nT of every site out) -------------------------------------------------------------------------------
 * One unit per site of the unit cell.  Phase A runs ctm_projectors_4x4_cc for every unit, phase B ctm_absorb_x + normalisation for
 * every unit with its own projectors (P, Pt) and those of unit `nb` (the neighbouring site along the move, ctmrg.py:351-425); no host
 * code runs in between.  The units of a phase run concurrently, one thread of this library per worker context (`workers`: nworkers
 * DISTINCT contexts of the same dtype as ctx, each with its own arena and stream; ctx itself is not among them); nworkers = 0: serially
 * on ctx.  Worker streams wait for the work queued on ctx's stream when the call starts; the call returns after every stream has
 * drained.  skip_zero_columns != 0: an absorb whose projectors (own and neighbour's) have at most chi/2 non-zero columns (S/S[0] >
 * cfg->svd_reltol is a prefix of the descending spectrum) runs on that prefix, rounded up to a multiple of 16, and pads the result
 * with the zeros the full product would have produced (same numbers).  All pointers: device memory, layouts as in the entries named. */
typedef struct ctm_move_unit {
    const double* proj[16];      /* in: tensors16 of ctm_projectors_4x4 for this site's 2x2 window */
    int proj_adims[20];          /* in: adims4x5 */
    double* basis;               /* in/out: warm-start workspace of this (direction, site) unit, or NULL */
    double* corner_buf[4];       /* in/out: caller-kept enlarged corners (see ctm_projectors_4x4_cc); used when use_corner_cache != 0 */
    int corner_valid[4];
    int use_corner_cache;
    const double* absorb[6];     /* in: C1, T1, T, T2, C2, A of ctm_absorb (tensors10 without the projectors) */
    int absorb_adims[5];
    int nb;                      /* in: index of the unit whose projectors are P1, Pt1 of this site's absorb */
    long long n_rows;            /* in: rows of P / Pt = chi * D_cut^2 of this unit */
    double* P; double* Pt;       /* out: n_rows x min(chi, n_rows) */
    double* S;                   /* out: min(chi, n_rows) */
    double* nC1; double* nC2; double* nT;   /* out: as ctm_absorb_x with chi_in = chi, chi_out = min(chi, n_rows) */
    int ncol;                    /* out: number of non-zero projector columns of this unit */
} ctm_move_unit;
int ctm_move(ctm_ctx* ctx, ctm_ctx* const* workers, int nworkers, int dir, int nunits, ctm_move_unit* units, int chi,
             const ctm_trunc_cfg* cfg, int normalize, int skip_zero_columns);

/* ---- one-site C4v move (ctm/one_site_c4v/ctmrg_c4v.py:325-463; ctm_components_c4v.py:52-130) ---------- */
int ctm_c2x2_c4v(ctm_ctx* ctx, int open, const double* a, const double* C, const double* T, int chi, int p, int D,
                 double* out);
int ctm_move_c4v(ctm_ctx* ctx, const double* a, const double* C, const double* T, int chi, int p, int D,
                 const ctm_trunc_cfg* cfg, double* C_out, double* T_out, double* D_out /* chi eigenvalues or NULL */);
/* Same, warm started: `basis` is an opaque caller-owned device workspace of (min(n, min(chi+1,n)+8) + 1) * n doubles (CTM_C128: (2 min(...) + 1) * n), zero-filled before the
 * first sweep and passed again on every later one (invariant subspace of the previous enlarged corner; see ctm_projectors_4x4_ws).
 * Once the enlarged corner is stationary a sweep restarts from that subspace (residual test on every kept pair + a deflated probe for
 * missed directions, see ctm_truncated_eigh_ws) instead of iterating; the returned tensors do not depend on the workspace. */
int ctm_move_c4v_ws(ctm_ctx* ctx, const double* a, const double* C, const double* T, int chi, int p, int D,
                    const ctm_trunc_cfg* cfg, double* C_out, double* T_out, double* D_out, double* basis);

/* Same with the normalisation of the new T selectable (_move_normalize_c, ctmrg_c4v.py:182-197): normalize = 1 'inf' (max-abs),
 * 2 vector 2-norm; C is always divided by |C[0,0]|. */
int ctm_move_c4v_x(ctm_ctx* ctx, const double* a, const double* C, const double* T, int chi, int p, int D,
                   const ctm_trunc_cfg* cfg, int normalize, double* C_out, double* T_out, double* D_out, double* basis);

/* ---- RDMs (ctm/generic/rdm.py:1362-1592, 71-302, 304-500, 622-826; one_site_c4v/rdm_c4v.py) ---------- */
/* rdm2x2: tensors16 = (C,T1,T2,a) for LU(coord), RU(coord+x), RD(coord+x+y), LD(coord+y); out p^8 raw
 * (un-normalised, index order s0 s1 s2 s3 s0' s1' s2' s3'); symmetrisation/normalisation is host-side. */
int ctm_rdm2x2(ctm_ctx* ctx, const double* const* tensors16, int chi, const int* adims4x5, double* out);
/* One part of the same contraction: only the lower-half slices cl in [lo0, lo1), cl = (s2 t2) p^2 + (s3 t3) (p^4 in all; s2 is
 * the site coord+y, s3 the site coord+x+y), are built; out receives the raw block R[(s0 t0 s1 t1), lo0:lo1] (p^4 x (lo1 - lo0),
 * row-major).  The blocks of all ranges put side by side give R[s0 t0 s1 t1 ; s2 t2 s3 t3], whose permutation
 * (0,2,4,6,1,3,5,7) is ctm_rdm2x2's output.  Peak workspace
 * n^2 (2 p^2 + 1 + lo1 - lo0) elements instead of n^2 (p^4 + 2 p^2 + 1): the unit that is sharded over the GPUs of a rank group
 * (models/j1j2.py:238-240 loops over sites; SURVEY 8e splits a site's RDM over a GPU pair) or looped over on one GPU when the
 * open halves do not fit (D = 8, chi = 384, complex128). */
int ctm_rdm2x2_part(ctm_ctx* ctx, const double* const* tensors16, int chi, const int* adims4x5, int lo0, int lo1, double* out);
/* rdm1x1 / rdm2x1 / rdm1x2 through the open-corner route; tensor lists documented in INTEGRATION.md */
int ctm_rdm1x1(ctm_ctx* ctx, const double* const* tensors9 /* C1,C2,C3,C4,T1,T2,T3,T4,a */, int chi, const int* adims,
               double* out /* p^2 */);
int ctm_rdm2x1(ctm_ctx* ctx, const double* const* tensors12, int chi, const int* adims2x5, double* out /* p^4 */);
int ctm_rdm1x2(ctm_ctx* ctx, const double* const* tensors12, int chi, const int* adims2x5, double* out /* p^4 */);
/* C4v: which = 0 rdm2x1_sl, 1 rdm2x2_NN_lowmem_sl, 2 rdm2x2_NNN_lowmem_sl, 3 rdm2x2 ; raw output */
int ctm_rdm_c4v(ctm_ctx* ctx, int which, const double* a, const double* C, const double* T, int chi, int p, int D,
                double* out);

/* ---- env initialisation pieces (ctm/generic/env.py:367-536; env_c4v.py:262-311) ---------------------- */
/* double-layer partial trace: kind 0..3 corners C(-1,-1),C(1,-1),C(1,1),C(-1,1); 4..7 T(0,-1),T(-1,0),T(0,1),T(1,0);
 * out is the un-padded, max-abs-normalised tensor of the NEIGHBOUR site `a`. */
int ctm_init_piece(ctm_ctx* ctx, int kind, const double* a, const int* adims, double* out);

#ifdef __cplusplus
}
#endif
#endif
