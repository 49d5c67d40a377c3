"""GPU: complex128 primitives of the native engine (planar complex GEMMs, complex one-sided Jacobi / block power
iteration, phase fixing) vs numpy / LAPACK.  The end-to-end complex CTM cases run in test_gpu_generic.py through the
committed reference fixture generic_D2_chi8_c128."""
import numpy as np
import pytest
import torch
from helpers import dev, relerr

pytestmark = pytest.mark.gpu


def crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def test_gemm_permute_normalize_c128(eng):
    rng = np.random.default_rng(0)
    A, B = crand(rng, 70, 45), crand(rng, 45, 33)
    assert relerr(eng.gemm(dev(A), dev(B)), A @ B) < 1e-13
    assert relerr(eng.gemm(dev(A.T.copy()), dev(B), 1, 0), A @ B) < 1e-13
    assert relerr(eng.gemm(dev(A.conj().T.copy()), dev(B), 2, 0), A @ B) < 1e-13
    assert relerr(eng.gemm(dev(A), dev(B.conj().T.copy()), 0, 2), A @ B) < 1e-13
    x = crand(rng, 3, 4, 5, 6)
    assert relerr(eng.permute(dev(x), (2, 0, 3, 1)), x.transpose(2, 0, 3, 1)) == 0.0
    y = dev(x)
    eng.normalize_inf_(y)
    assert relerr(y, x / np.abs(x).max()) < 1e-15


@pytest.mark.parametrize("n,chi", [(24, 8), (96, 20), (200, 33)])
def test_truncated_svd_c128_full(eng, n, chi):
    rng = np.random.default_rng(n)
    M = crand(rng, n, n) * (0.7 ** np.arange(n))[None, :]
    U, S, V = (t.cpu().numpy() for t in eng.truncated_svd(dev(M), chi, eng.cfg(keep_multiplets=False)))
    Ur, Sr, Vh = np.linalg.svd(M)
    assert np.abs(S - Sr[:chi]).max() < 1e-12 * Sr[0]
    # triplets: M V = U S, U^H M = S V^H ; orthonormality; phase convention (max-|U| entry real positive)
    assert np.abs(M @ V - U * S).max() < 1e-11 * Sr[0]
    assert np.abs(U.conj().T @ M - S[:, None] * V.conj().T).max() < 1e-11 * Sr[0]
    assert np.abs(U.conj().T @ U - np.eye(chi)).max() < 1e-12
    assert np.abs(V.conj().T @ V - np.eye(chi)).max() < 1e-12
    piv = U[np.abs(U).argmax(axis=0), np.arange(chi)]
    assert np.abs(piv.imag).max() < 1e-12 and (piv.real > 0).all()
    sv = eng.svdvals(dev(M)).cpu().numpy()
    assert np.abs(sv - Sr).max() < 1e-12 * Sr[0]


def test_truncated_svd_c128_iterative(eng):
    """n >= 256 (si_min_n): leading-chi triplets by the complex block power iteration (residual-verified)."""
    rng = np.random.default_rng(5)
    n, chi = 640, 24
    Q1, _ = np.linalg.qr(crand(rng, n, n)); Q2, _ = np.linalg.qr(crand(rng, n, n))
    sv = 0.6 ** np.arange(n)
    M = (Q1 * sv) @ Q2.conj().T
    h0 = eng.stat("si_hits")
    U, S, V = (t.cpu().numpy() for t in eng.truncated_svd(dev(M), chi, eng.cfg(keep_multiplets=False)))
    assert eng.stat("si_hits") == h0 + 1
    assert np.abs(S - sv[:chi]).max() < 1e-12
    assert np.abs(M @ V - U * S).max() < 1e-11
    assert np.abs(U.conj().T @ U - np.eye(chi)).max() < 1e-11


def test_projectors_c128_fused_vs_explicit(eng):
    from oracle import ctm_oracle as O
    rng = np.random.default_rng(7)
    n, chi = 64, 12
    R, Rt = crand(rng, n, n), crand(rng, n, n)
    P, Pt, S = eng.projectors(dev(R), dev(Rt), chi, return_S=True)
    Pr, Ptr, Sr = O.projectors_from_matrices(R, Rt, chi, return_S=True)
    assert relerr(S, Sr) < 1e-12
    assert relerr(P @ Pt.t(), Pr @ Ptr.T) < 1e-9
    assert relerr(P.abs(), np.abs(Pr)) < 1e-9
    G = (Pt.t() @ P).cpu().numpy()
    assert np.abs(G - np.eye(chi)).max() < 1e-9


def test_projectors_4x4_c128_implicit_operator(eng):
    """n = chi D^2 = 512: the fused path (four corners, M = R^T Rt applied implicitly inside the complex block power
    iteration) equals the explicit halves -> projectors route and the oracle."""
    from oracle import ctm_oracle as O
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from ctm.generic.ctm_components import _halves_t
    rng = np.random.default_rng(11)
    D, chi = 4, 32
    sites = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, D, D, D, D)) + 1j * rng.random((2, D, D, D, D))
            sites[(x, y)] = A / np.abs(A).max()
    st = IPEPS({k: dev(v) for k, v in sites.items()})
    env = ENV(chi, st); init_env(st, env)
    for d in [(0, -1), (-1, 0), (0, 1), (1, 0)] * 2:
        ctmrg.ctm_MOVE(d, st, env)                       # dense complex environment
    ost = O.State(sites); oe = O.Env(chi)
    oe.C = {k: v.cpu().numpy() for k, v in env.C.items()}; oe.T = {k: v.cpu().numpy() for k, v in env.T.items()}
    h0 = eng.stat("si_hits")
    for d in [(0, -1), (1, 0)]:
        t16 = _halves_t(d, (0, 0), st, env)
        R, Rt = eng.halves(d, t16)
        Ro, Rto = O.halves(d, (0, 0), ost, oe)
        assert relerr(R, Ro) < 1e-12 and relerr(Rt, Rto) < 1e-12
        P, Pt, S = eng.projectors(R, Rt, chi, return_S=True)
        P2, Pt2, S2 = eng.projectors_4x4(d, t16, chi, return_S=True)
        Po, Pto, So = O.projectors_from_matrices(Ro, Rto, chi, return_S=True)
        assert relerr(S, So) < 1e-11 and relerr(S2, So) < 1e-11
        assert relerr(P2 @ Pt2.t(), Po @ Pto.T) < 1e-6 and relerr(P @ Pt.t(), Po @ Pto.T) < 1e-6
    assert eng.stat("si_hits") >= h0 + 4
