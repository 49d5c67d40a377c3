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
    """n >= 512: leading-chi triplets by the complex block power iteration (residual-verified)."""
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
