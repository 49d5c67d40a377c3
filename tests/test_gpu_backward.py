"""GPU: adjoints of the two decompositions (first part of the backward pass): ctm_svd_backward / ctm_eigh_backward vs
the reference's SVDGESDD.backward / SYMEIG.backward outputs (golden, real and complex, square and thin factors), and the
torch.autograd Functions built on them vs finite differences of the native forward."""
import numpy as np
import pytest
import torch
from conftest import golden
from helpers import dev, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["sq_f64", "thin_f64", "sq_c128", "thin_c128"])
def test_svd_backward_golden(eng, tag):
    g = golden("backward")
    a = {nm: dev(g[f"svd_{tag}_{nm}"]) for nm in ("U", "S", "V", "gU", "gS", "gV")}
    ref = g[f"svd_{tag}_dA"]
    assert relerr(eng.svd_backward(a["U"], a["S"], a["V"], a["gU"], a["gS"], a["gV"], eps=1e-12), ref) < 1e-12
    # gradients may be absent: dA is linear in (gU, gS, gV)
    parts = [eng.svd_backward(a["U"], a["S"], a["V"], gU=a["gU"], eps=1e-12), eng.svd_backward(a["U"], a["S"], a["V"], gS=a["gS"], eps=1e-12),
             eng.svd_backward(a["U"], a["S"], a["V"], gV=a["gV"], eps=1e-12)]
    assert relerr(parts[0] + parts[1] + parts[2], ref) < 1e-12


@pytest.mark.parametrize("tag", ["f64", "c128"])
def test_eigh_backward_golden(eng, tag):
    g = golden("backward")
    a = {nm: dev(g[f"eig_{tag}_{nm}"]) for nm in ("D", "U", "gD", "gU")}
    assert relerr(eng.eigh_backward(a["D"], a["U"], a["gD"], a["gU"], reg=1e-12), g[f"eig_{tag}_dA"]) < 1e-12


def test_autograd_functions_against_finite_differences(eng):
    """d/dA of sum(w_S * S) + gauge-invariant function of (U, V) through linalg.svd_gesdd.SVDGESDD, and of an eigenvalue functional
    through linalg.eig_sym.SYMEIG, against central differences of the native forward."""
    from linalg.svd_gesdd import SVDGESDD
    from linalg.eig_sym import SYMEIG
    rng = np.random.default_rng(9)
    n = 12
    A = dev(rng.standard_normal((n, n))).requires_grad_(True)
    w = dev(rng.standard_normal(n))
    W = dev(rng.standard_normal((n, n)))

    def f(M):
        U, S, V = SVDGESDD.apply(M, 1e-12)
        return (w * S).sum() + ((U * S) @ V.t() * W).sum()          # second term = <A, W>: its gradient is W
    L = f(A); L.backward()
    gnum = torch.zeros_like(A)
    h = 1e-6
    with torch.no_grad():
        for i in range(n):
            for j in range(0, n, 5):
                E = torch.zeros_like(A); E[i, j] = h
                gnum[i, j] = (f(A + E) - f(A - E)) / (2 * h)
    mask = gnum != 0
    assert float((A.grad - gnum)[mask].abs().max()) < 1e-6
    H = dev(rng.standard_normal((n, n))); H = (0.5 * (H + H.t())).requires_grad_(True)

    def fe(M):
        D, U = SYMEIG.apply(M, 1e-12)
        return (w * D).sum() + ((U * D) @ U.t() * W).sum()
    Le = fe(H); Le.backward()
    gsym = 0.5 * (H.grad + H.grad.t())
    with torch.no_grad():
        for (i, j) in ((0, 0), (2, 5), (7, 3), (11, 11)):
            E = torch.zeros_like(H); E[i, j] += h / 2; E[j, i] += h / 2           # symmetric perturbation
            num = (fe(H + E) - fe(H - E)) / (2 * h)
            ana = float(gsym[i, j]) if i == j else float(gsym[i, j])
            assert abs(float(num) - ana) < 1e-6, (i, j)
