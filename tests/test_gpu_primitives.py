"""GPU parity of the native primitives (GEMM / permute / truncated SVD / eigh) through the C-ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (64, 64, 64), (128, 128, 128), (130, 70, 33), (257, 129, 515), (8, 300, 17), (512, 384, 1024), (1, 1, 1),
                                   # K split of the vectorised kernel (few 128x128 tiles, long K); 128 q + r rows / columns split into
                                   # a vectorised part and a strip
                                   (128, 4096, 2048), (384, 2048, 4096), (257, 2048, 2048), (4096, 300, 1024), (300, 2048, 2048),
                                   # streaming strip kernel (<= 64 rows times a big operand read once), every row-tile count, odd K split
                                   (32, 4096, 1024), (64, 2048, 2048), (17, 2048, 2064), (1, 4096, 1024), (48, 2080, 2048), (33, 6400, 1040)])
@pytest.mark.parametrize("tA", [False, True])
@pytest.mark.parametrize("tB", [False, True])
def test_gemm(eng, M, N, K, tA, tB):
    rng = np.random.default_rng(M * 1000 + N * 10 + K)
    A = rng.standard_normal((K, M) if tA else (M, K))
    B = rng.standard_normal((N, K) if tB else (K, N))
    ref = (A.T if tA else A) @ (B.T if tB else B)
    out = eng.gemm(dev(A), dev(B), tA, tB).cpu().numpy()
    assert np.abs(out - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()) * np.sqrt(K)


@pytest.mark.parametrize("shape,perm", [((5, 7), (1, 0)), ((64, 48), (1, 0)), ((3, 4, 5), (2, 0, 1)), ((8, 2, 2, 8, 3, 3), (1, 2, 4, 0, 3, 5)),
                                        ((6, 3, 3, 6, 2, 3, 3, 2), (1, 3, 6, 0, 4, 7, 2, 5)), ((40, 33, 17), (2, 1, 0)), ((4, 5, 6), (0, 1, 2)),
                                        ((16, 9, 16, 9), (2, 3, 0, 1)), ((2, 2, 2, 2, 2, 2, 2, 2), (0, 2, 4, 6, 1, 3, 5, 7))])
def test_permute(eng, shape, perm):
    x = np.random.default_rng(1).standard_normal(shape)
    out = eng.permute(dev(x), perm).cpu().numpy()
    assert np.array_equal(out, np.ascontiguousarray(x.transpose(perm)))


def _graded(n, rng, decay):
    Q1, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.exp(-decay * np.arange(n))
    return (Q1 * s) @ Q2.T, s


@pytest.mark.parametrize("n,chi,decay", [(24, 8, 0.7), (64, 16, 0.3), (100, 20, 0.2), (288, 32, 0.05), (512, 64, 0.03), (1024, 64, 0.02), (1536, 96, 0.05)])
def test_truncated_svd(eng, n, chi, decay):
    rng = np.random.default_rng(n)
    M, s = _graded(n, rng, decay)
    U, S, V = (t.cpu().numpy() for t in eng.truncated_svd(dev(M), chi))
    Sr = np.linalg.svd(M, compute_uv=False)[:chi]
    assert np.abs(S - Sr).max() <= 1e-13 * Sr[0]
    # orthonormality and reconstruction of the leading block (svd_symeig.py:88 style bound)
    assert np.abs(U.T @ U - np.eye(chi)).max() < 1e-12
    assert np.abs(V.T @ V - np.eye(chi)).max() < 1e-12
    assert np.abs(U.T @ M @ V - np.diag(S)).max() < 1e-13 * Sr[0] * n


def test_truncated_svd_random_dense(eng):
    rng = np.random.default_rng(5)
    M = rng.random((200, 200))
    U, S, V = (t.cpu().numpy() for t in eng.truncated_svd(dev(M), 40))
    Sr = np.linalg.svd(M, compute_uv=False)[:40]
    assert np.abs(S - Sr).max() <= 1e-13 * Sr[0]
    assert np.abs(U.T @ M @ V - np.diag(S)).max() < 1e-12 * Sr[0]


def test_truncated_svd_golden(eng):
    from conftest import golden
    g = golden("decomp")
    cfg = eng.cfg(eps_multiplet=1e-8)
    for nm in "abc":
        M = g[f"svd_{nm}_M"]
        U, S, V = (t.cpu().numpy() for t in eng.truncated_svd(dev(M), 8, cfg))
        assert np.abs(S - g[f"svd_{nm}_S"]).max() < 1e-13, nm
        assert ((S == 0) == (g[f"svd_{nm}_S"] == 0)).all(), nm          # multiplet back-off / rank deficiency
        nzc = S > 1e-10
        # sign-fixed vectors agree entrywise where the spectrum is non-degenerate (case a)
        if nm == "a":
            assert np.abs(np.abs(U[:, nzc]) - np.abs(g["svd_a_U"][:, nzc])).max() < 1e-9
            assert np.abs(U[:, nzc] - g["svd_a_U"][:, nzc]).max() < 1e-9


@pytest.mark.parametrize("n,chi", [(24, 6), (81, 20), (144, 16), (256, 64), (768, 48), (1024, 64)])
def test_truncated_eigh(eng, n, chi):
    rng = np.random.default_rng(n + 1)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.exp(-0.1 * np.arange(n)) * np.where(rng.random(n) < 0.3, -1.0, 1.0)
    H = (Q * lam) @ Q.T
    H = 0.5 * (H + H.T)
    D, U = (t.cpu().numpy() for t in eng.truncated_eigh(dev(H), chi, eng.cfg(keep_multiplets=False)))
    w = np.linalg.eigvalsh(H)
    w = w[np.argsort(-np.abs(w))][:chi]
    assert np.abs(D - w).max() < 1e-13
    assert np.abs(U.T @ U - np.eye(chi)).max() < 1e-12
    assert np.abs(H @ U - U * D[None, :]).max() < 1e-12


def test_eigh_golden(eng):
    from conftest import golden
    g = golden("decomp")
    H = g["eig_H"]
    for nm, ch in (("a", 4), ("b", 6)):
        D, U = (t.cpu().numpy() for t in eng.truncated_eigh(dev(H), ch))
        assert np.abs(D - g[f"eig_{nm}_D"]).max() < 1e-13
        assert ((D == 0) == (g[f"eig_{nm}_D"] == 0)).all()


def test_svdvals(eng):
    rng = np.random.default_rng(3)
    C = rng.standard_normal((18, 18))
    S = eng.svdvals(dev(C)).cpu().numpy()
    assert np.abs(S - np.linalg.svd(C, compute_uv=False)).max() < 1e-13 * S[0]


@pytest.mark.parametrize("n,chi", [(24, 24), (81, 20), (300, 300), (768, 48)])
def test_svd_symeig(eng, n, chi):
    """ctm_svd_symeig (reference linalg/svd_symeig.py:12-34, custom_svd.py:143-208) vs the oracle restatement."""
    from oracle import ctm_oracle as O
    rng = np.random.default_rng(n + 7)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.exp(-0.07 * np.arange(n)) * np.where(rng.random(n) < 0.4, -1.0, 1.0)
    H = (Q * lam) @ Q.T
    H = 0.5 * (H + H.T)
    U, S, V = (t.cpu().numpy() for t in eng.svd_symeig(dev(H), chi))
    Uo, So, Vo = O.truncated_svd_symeig(H, chi)
    assert np.abs(S - So).max() < 1e-13 and (np.diff(S) <= 0).all()
    # V = U sign(D) exactly, column by column
    sg = np.sign(np.sum(U * V, axis=0))
    assert np.array_equal(V, U * sg[None, :])
    assert np.array_equal(sg, np.sign(np.sum(Uo * Vo, axis=0)))                # the eigenvalue signs
    assert np.abs(U.T @ U - np.eye(chi)).max() < 1e-12
    assert np.abs(H @ V - U * S[None, :]).max() < 1e-12                         # M v = s u  (M symmetric: M u sign = s u)
    if chi == n:
        assert np.linalg.norm(H - (U * S) @ V.T) < S[0] * n * n * 1e-14          # svd_symeig.py:88


def test_truncated_svd_symeig_host_wrapper(eng):
    from linalg.custom_svd import truncated_svd_symeig
    from linalg.svd_symeig import SVDSYMEIG
    from conftest import golden
    H = golden("decomp")["eig_H"]
    for nm, ch in (("a", 4), ("b", 6)):
        U, S, V = (t.cpu().numpy() for t in truncated_svd_symeig(dev(H), ch, keep_multiplets=True, eps_multiplet=1e-12))
        D = golden("decomp")[f"eig_{nm}_D"]
        assert np.abs(S - np.abs(D)).max() < 1e-13 and ((S == 0) == (D == 0)).all()
    U, S, V = (t.cpu().numpy() for t in SVDSYMEIG.apply(dev(H)))
    assert np.linalg.norm(H - (U * S) @ V.T) < S[0] * H.shape[0] ** 2 * 1e-14


@pytest.mark.parametrize("rows_kernel", [True, False], ids=["lds-row-block", "streaming-strip"])
def test_row_block_times_big_operand_kernels(eng, rows_kernel):
    """The two kernels behind (<= 64 rows) x (big operand): LDS-tiled row-block kernel and LDS-free streaming strip kernel, both
    operand layouts, against torch's fp64 matmul."""
    import torch
    n = 2048
    g = torch.Generator(device="cuda").manual_seed(3)
    B = torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g)
    old = (1, 1)              # the defaults of csrc/ctm_common.h
    eng.set_option("rows_kernel_min_m", 1 if rows_kernel else 1000)
    eng.set_option("rows_kernel_min_m_kc", 1 if rows_kernel else 1000)
    try:
        for m in (8, 16, 31, 32, 33, 48, 64):
            A = torch.randn(m, n, dtype=torch.float64, device="cuda", generator=g)
            for tB in (False, True):
                C = eng.gemm(A, B, transB=tB)
                ref = A @ (B.t() if tB else B)
                assert (C - ref).abs().max().item() < 1e-11 * ref.abs().max().item(), (m, tB)
    finally:
        eng.set_option("rows_kernel_min_m", old[0]); eng.set_option("rows_kernel_min_m_kc", old[1])


def test_in_launch_split_k_combine_equals_the_separate_reduce_kernel(eng):
    """The row-block kernel sums its K-slice partials inside the launch (last workgroup of a column tile, fixed slice order); the
    separate reduce kernel sums the same partials in the same order: bit-identical products, in both operand layouts, for several row
    counts, and repeatedly (a stale or half-visible slab would show as a difference)."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(7)
    n, k = 4096, 8192
    B = torch.randn(k, n, dtype=torch.float64, device="cuda", generator=g)
    Bt = B.t().contiguous()
    try:
        for m in (16, 32, 48, 64):
            A = torch.randn(m, k, dtype=torch.float64, device="cuda", generator=g)
            eng.set_option("rows_fused_reduce", 0)
            ref, reft = eng.gemm(A, B), eng.gemm(A, Bt, transB=True)
            eng.set_option("rows_fused_reduce", 1)
            for _ in range(5):
                assert torch.equal(eng.gemm(A, B), ref), m
                assert torch.equal(eng.gemm(A, Bt, transB=True), reft), m
            assert (ref - A @ B).abs().max().item() < 1e-11 * ref.abs().max().item()
    finally:
        eng.set_option("rows_fused_reduce", 1)


@pytest.mark.parametrize("cplx", [False, True])
def test_full_decomposition_of_a_rank_deficient_matrix_returns_orthonormal_factors(eng, cplx):
    """chi = n on a matrix of rank n/3 (plus singular values at the rounding level): U and V are complete orthonormal bases, as
    LAPACK returns them -- the differentiable full decomposition feeds both to the regularised backward."""
    n, r = 96, 32
    g = torch.Generator().manual_seed(17)
    dt = torch.complex128 if cplx else torch.float64
    A = torch.randn(n, r, generator=g, dtype=dt) * (0.5 ** torch.arange(r, dtype=torch.float64)).to(dt)
    B = torch.randn(r, n, generator=g, dtype=dt)
    M = (A @ B).cuda()
    U, S, V = eng.truncated_svd(M, n, eng.cfg(keep_multiplets=False))
    I = torch.eye(n, dtype=dt, device=M.device)
    assert float((U.conj().T @ U - I).abs().max()) < 1e-12
    assert float((V.conj().T @ V - I).abs().max()) < 1e-12
    Sref = torch.linalg.svdvals(M.cpu())
    assert float((S.cpu() - Sref).abs().max()) < 1e-12 * float(Sref[0])
    assert float((U * S.to(dt) @ V.conj().T - M).abs().max()) < 1e-12 * float(Sref[0])


@pytest.mark.parametrize("cplx", [False, True])
def test_full_svd_warm_start_and_rank_deficient_completion(eng, cplx):
    """chi = n (the differentiable route's SVD node) with a workspace: the second call on a slightly changed matrix starts the sweeps
    from the previous left vectors (fewer sweeps, same decomposition); the matrix has rank n - 40, so V's last 40 columns are an
    orthonormal completion (float64: pivoted projector rows + Newton-Schulz) -- U, V unitary, U S V^H = M, S as LAPACK's."""
    n, r = 320, 280
    g = torch.Generator().manual_seed(23)
    dt = torch.complex128 if cplx else torch.float64
    A = torch.randn(n, r, generator=g, dtype=dt) @ torch.diag(torch.logspace(0, -6, r, dtype=torch.float64).to(dt)) @ torch.randn(r, n, generator=g, dtype=dt)
    P = torch.randn(n, r, generator=g, dtype=dt)
    M0 = A.cuda()
    M1 = ((A + 1e-5 * P @ torch.linalg.pinv(torch.randn(n, r, generator=g, dtype=dt)) @ A)).cuda()      # same row space, slightly moved
    cfgT = eng.cfg(keep_multiplets=False)
    basis = eng.warm_basis(n, n, dt)
    eng.timers(reset=True)
    eng.truncated_svd(M0, n, cfgT, basis=basis)
    cold = eng.stat("total_sweeps")
    assert eng.stat("eigh_warm_hits") == 0
    eng.timers(reset=True)
    U, S, V = eng.truncated_svd(M1, n, cfgT, basis=basis)
    assert eng.stat("eigh_warm_hits") == 1 and eng.stat("total_sweeps") <= cold
    assert eng.stat("svd_polar_completions") == 1 and eng.stat("svd_eig_completions") == 0      # pivoted projector rows + Newton-Schulz
    I = torch.eye(n, device=U.device, dtype=U.dtype)
    assert float((U.conj().T @ U - I).abs().max()) < 1e-11
    assert float((V.conj().T @ V - I).abs().max()) < 1e-11
    s0 = float(S[0])
    assert float(((U * S.to(U.dtype)) @ V.conj().T - M1).abs().max()) < 1e-11 * s0
    ref = torch.linalg.svdvals(M1.cpu())
    assert float((S.cpu() - ref).abs().max()) < 1e-12 * s0
    assert int((S > 1e-11 * s0).sum()) == r


def test_lazily_conjugated_tensors_are_read_by_value(eng):
    """torch keeps `x.conj()` as a flag on x's memory (so does the gradient that flows back through `V.conj().transpose(-2, -1)`):
    every entry point materialises it before taking the pointer."""
    g = torch.Generator().manual_seed(31)
    A = torch.randn(48, 48, generator=g, dtype=torch.complex128).cuda()
    B = torch.randn(48, 48, generator=g, dtype=torch.complex128).cuda()
    Ac = A.conj()
    assert Ac.is_conj() and Ac.data_ptr() == A.data_ptr()
    assert float((eng.gemm(Ac, B) - A.conj().resolve_conj() @ B).abs().max()) < 1e-12
    assert float((eng.einsum('ab,bc->ac', Ac, B.conj()) - (A.conj() @ B.conj()).resolve_conj()).abs().max()) < 1e-12
    s1 = eng.svdvals(Ac)
    assert float((s1.cpu() - torch.linalg.svdvals(A.cpu())).abs().max()) < 1e-12
    U, S, V = eng.truncated_svd(Ac, 48, eng.cfg(keep_multiplets=False))
    assert float(((U * S.to(U.dtype)) @ V.conj().T - A.conj().resolve_conj()).abs().max()) < 1e-11
    gU = torch.randn(48, 48, generator=g, dtype=torch.complex128).cuda()
    d1 = eng.svd_backward(U, S, V, gU=gU.conj(), gS=None, gV=None)
    d2 = eng.svd_backward(U, S, V, gU=gU.conj().resolve_conj(), gS=None, gV=None)
    assert float((d1 - d2).abs().max()) == 0.0
