"""GPU: the BASELINE configurations at their FULL sizes (C4v D4 chi64; generic 2x2 D6 chi128, D8 chi256; one unit of
D8 chi384 complex128), checked through size-independent properties -- the oracle cannot run at these sizes in seconds:

 * singular triplets of the truncation: residuals |M v - s u|, |M^T u - s v| <= 1e-12 s0, orthonormal U and V, and
   completeness of the leading chi (norm of the deflated operator M - U S V^T, estimated by power iteration, does not
   exceed the smallest kept value);
 * the fused, implicit-operator projector path against the explicit n x n matrices: same singular values (and the
   same as host LAPACK's svdvals where that finishes in seconds), biorthogonality P^T Pt = 1 on the kept subspace;
 * invariances of a whole sweep: scale invariance under a -> 2a (exact in binary floating point up to the max-abs
   normalisation), every new tensor has max-abs 1, and a 1x1 state tiled on the 2x2 cell gives the same environment
   on all four sites (checks the concurrent scheduling of the site units);
 * C4v: the eigenvalues of the move equal torch.linalg.eigh of the enlarged corner, ordered by magnitude, signs kept;
   the new T is symmetric in its two environment legs.
Tolerances are relative to the largest singular/eigen value; the north-star bound is 1e-10."""
import numpy as np
import pytest
import torch
from helpers import dev

pytestmark = pytest.mark.gpu
UP = (0, -1)


def _state(D, seed, cplx=False, signed=False, tiled=False):
    from ipeps.ipeps import IPEPS
    rng = np.random.default_rng(seed)
    sites = {}
    for y in range(2):
        for x in range(2):
            if tiled and sites:
                sites[(x, y)] = sites[(0, 0)].copy(); continue
            A = rng.random((2, D, D, D, D)) - (0.5 if signed else 0.0)
            if cplx:
                A = A + 1j * (rng.random((2, D, D, D, D)) - (0.5 if signed else 0.0))
            sites[(x, y)] = A / np.abs(A).max()
    return IPEPS({k: dev(v) for k, v in sites.items()})


def _sweep(st, env, n=1):
    import config as cfg
    from ctm.generic import ctmrg
    for _ in range(n):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _r in range(2):
                ctmrg.ctm_MOVE(d, st, env)


def _deflated_norm(M, U, S, V, iters=12, seed=0):
    """Power-iteration estimate of |M - U diag(S) V^H|_2 (a lower bound that converges from below)."""
    g = torch.Generator(device=M.device); g.manual_seed(seed)
    x = torch.randn(M.shape[1], 4, dtype=torch.float64, device=M.device, generator=g).to(M.dtype)
    est = 0.0
    for _ in range(iters):
        x = x / torch.linalg.vector_norm(x, dim=0, keepdim=True)
        y = M @ x - U @ (S.to(M.dtype)[:, None] * (V.conj().T @ x))
        est = float(torch.linalg.vector_norm(y, dim=0).max())
        x = M.conj().T @ y - V @ (S.to(M.dtype)[:, None] * (U.conj().T @ y))
    return est


def _check_unit(eng, st, env, chi, coord=(0, 0), direction=UP, torch_svdvals=False):
    from ctm.generic.ctm_components import _halves_t
    t16 = _halves_t(direction, coord, st, env)
    R, Rt = eng.halves(direction, t16)
    M = eng.gemm(R, Rt, transA=True)                                   # M = R^T Rt, plain transpose (ctm_projectors.py:263)
    n = M.shape[0]
    P, Pt, S = eng.projectors_4x4(direction, t16, chi, return_S=True)   # fused path: the n x n matrices are never formed
    U, S2, V = eng.truncated_svd(M, chi)                                # explicit matrix
    s0 = float(S[0])
    assert float((S - S2).abs().max()) < 1e-11 * s0
    if torch_svdvals:
        ref = torch.linalg.svdvals(M.cpu())[:chi].to(S.device)       # LAPACK on the host, the reference's own route
        kept = (S > 0)
        assert float((S - ref)[kept].abs().max()) < 1e-11 * s0
    k = int((S2 > 0).sum())
    U, V, Sk = U[:, :k], V[:, :k], S2[:k]
    I = torch.eye(k, dtype=M.dtype, device=M.device)
    assert float((U.conj().T @ U - I).abs().max()) < 1e-12
    assert float((V.conj().T @ V - I).abs().max()) < 1e-12
    assert float((M @ V - U * Sk.to(M.dtype)).abs().max()) < 1e-12 * s0
    assert float((M.conj().T @ U - V * Sk.to(M.dtype)).abs().max()) < 1e-12 * s0
    # completeness: nothing larger than the smallest kept value is left in the deflated operator (up to the multiplet
    # back-off, which may drop values within 1e-8 relative gap of the cut; and to the numerical rank of M)
    rest = _deflated_norm(M, U, Sk, V)
    assert rest <= max(float(Sk[-1]) * (1 + 1e-6), 1e-12 * s0), (rest, float(Sk[-1]))
    # biorthogonality of the fused projectors on the kept subspace: P^T Pt = S^-1/2 U^H M V S^-1/2 = 1, entries weighted
    # by sqrt(s_i s_j)/s0 (the 1/sqrt(s) of the smallest kept triplets amplifies their rounding)
    kk = int((S > 1e-8 * s0).sum())                                     # projector columns below svd_reltol are zero (ctm_projectors.py:266-270)
    G = P[:, :kk].T @ Pt[:, :kk] - torch.eye(kk, dtype=M.dtype, device=M.device)
    w = torch.sqrt(S[:kk] / s0)
    assert float((G.abs() * w[:, None] * w[None, :]).max()) < 1e-11
    return n


@pytest.mark.parametrize("D,chi,signed", [(6, 128, False), (6, 128, True), (8, 256, False)], ids=["D6chi128", "D6chi128-signed", "D8chi256"])
def test_generic_unit_at_full_size(eng, D, chi, signed):
    from ctm.generic.env import ENV, init_env
    st = _state(D, 3, signed=signed)
    env = ENV(chi, st); init_env(st, env)
    _sweep(st, env, 2 if D == 6 else 1)
    n = _check_unit(eng, st, env, chi, torch_svdvals=(D == 6))
    assert n == chi * D * D
    eng.trim()


def test_unit_of_the_complex_config_at_full_size(eng):
    """One unit of generic D8 chi384 complex128 (n = 24576): fused projectors only (P^T Pt = 1, plain transpose)."""
    from ctm.generic.env import ENV, init_env
    from ctm.generic.ctm_components import _halves_t
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~100 GB of free HBM")
    chi, D = 384, 8
    st = _state(D, 4, cplx=True)
    env = ENV(chi, st); init_env(st, env)
    t16 = _halves_t(UP, (0, 0), st, env)
    P, Pt, S = eng.projectors_4x4(UP, t16, chi, return_S=True)
    assert P.shape == (chi * D * D, chi) and P.dtype == torch.complex128
    s0 = float(S[0]); kk = int((S > 1e-8 * s0).sum())
    assert kk >= 1 and bool((S[:kk - 1] >= S[1:kk]).all())
    G = P[:, :kk].T @ Pt[:, :kk] - torch.eye(kk, dtype=P.dtype, device=P.device)
    w = torch.sqrt(S[:kk] / s0)
    assert float((G.abs() * w[:, None] * w[None, :]).max()) < 1e-11
    eng.trim()


def test_sweep_invariances_at_full_size(eng):
    from ctm.generic.env import ENV, init_env
    D, chi = 6, 128
    st = _state(D, 8, tiled=True)
    env = ENV(chi, st); init_env(st, env)
    _sweep(st, env, 2)
    for k, t in list(env.C.items()) + list(env.T.items()):
        assert abs(float(t.abs().max()) - 1.0) < 1e-14, k
    # 1x1 state tiled on 2x2: identical environments on all sites
    for (c, v), t in env.C.items():
        assert float((t.abs() - env.C[((0, 0), v)].abs()).abs().max()) < 1e-9, (c, v)
    spec = env.get_spectra()
    for (c, v), s in spec.items():
        assert float((s - spec[((0, 0), v)]).abs().max()) < 1e-11, (c, v)
    # a -> 2a: the max-abs normalised environment does not change
    from ipeps.ipeps import IPEPS
    st2 = IPEPS({k: 2.0 * v for k, v in st.sites.items()})
    env2 = ENV(chi, st2); init_env(st2, env2)
    _sweep(st2, env2, 2)
    spec2 = env2.get_spectra()
    for k in spec:
        assert float((spec[k] - spec2[k]).abs().max()) < 1e-12, k


def test_c4v_move_at_full_size(eng):
    from groups.pg import make_c4v_symm
    D, chi = 4, 64
    rng = np.random.default_rng(2)
    A = torch.from_numpy(rng.random((2, D, D, D, D))).cuda()
    A = make_c4v_symm(A); A = A / A.abs().max()
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    from ctm.one_site_c4v import ctmrg_c4v
    st = IPEPS_C4V(A)
    env = ENV_C4V(chi, st); init_env(st, env)
    for _ in range(6):
        ctmrg_c4v.ctm_MOVE_sl(A, env)
    C_, T = env.get_C(), env.get_T()
    X = eng.c2x2_c4v(A, C_, T)
    assert float((X - X.T).abs().max()) < 1e-12 * float(X.abs().max())
    nC, nT, Dv = eng.move_c4v(A, C_, T)
    w = torch.linalg.eigvalsh(0.5 * (X + X.T))
    w = w[torch.argsort(w.abs(), descending=True)][:chi]
    kept = Dv != 0
    assert float((Dv - w)[kept].abs().max()) < 1e-12 * float(w.abs().max())
    assert float((nT - nT.permute(1, 0, 2)).abs().max()) < 1e-13
    assert abs(float(nT.abs().max()) - 1.0) < 1e-14
