"""GPU: the reference's OWN unit tests of its differentiable decompositions, run on the native nodes -- linalg/eig_sym.py:80-127
(test_SYMEIG_random, _3x3degenerate, _rank_deficient), linalg/svd_gesdd.py:623-693 (test_SVDGESDD_random on the square case the
hot path has, test_SVDGESDD_COMPLEX_random with nearly and exactly degenerate pairs), linalg/svd_symeig.py:82-151
(test_SVDSYMEIG_random / _3x3degenerate / _rank_deficient: forward only, as that route is here).  Same sizes, same tolerances."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_SYMEIG_random(eng):
    from linalg.eig_sym import SYMEIG
    m = 50
    g = torch.Generator().manual_seed(1)
    M = torch.rand(m, m, generator=g, dtype=torch.float64).cuda()
    M = 0.5 * (M + M.t())
    D, U = SYMEIG.apply(M, 1.0e-12)
    assert torch.norm(M - U @ torch.diag(D) @ U.t()) < D[0].abs() * (m ** 2) * 1e-14
    assert float((D.abs()[:-1] - D.abs()[1:]).min()) >= 0.0                     # ordered by |D| descending (eig_sym.py:25-34)
    M.requires_grad_(True)

    def force_sym_eig(M):
        M = 0.5 * (M + M.t())
        return SYMEIG.apply(M, 1.0e-12)
    assert torch.autograd.gradcheck(force_sym_eig, M, eps=1e-6, atol=1e-4)


def test_SYMEIG_3x3degenerate(eng):
    from linalg.eig_sym import SYMEIG
    M = torch.zeros((3, 3), dtype=torch.float64)
    M[0, 1] = M[0, 2] = M[1, 2] = 1.
    M = (0.5 * (M + M.t())).cuda()
    D, U = SYMEIG.apply(M, 1.0e-12)
    assert torch.norm(M - U @ torch.diag(D) @ U.t()) < D[0].abs() * (M.size()[0] ** 2) * 1e-14
    assert float((D.cpu() - torch.tensor([1.0, -0.5, -0.5], dtype=torch.float64)).abs().max()) < 1e-14
    # a function that does not depend on the basis chosen inside the degenerate pair: the regularised backward differentiates it
    M.requires_grad_(True)

    def invariant(M):
        M = 0.5 * (M + M.t())
        D, U = SYMEIG.apply(M, 1.0e-12)
        return (U[:, :1] @ U[:, :1].t()), D.sum()
    assert torch.autograd.gradcheck(invariant, M, eps=1e-6, atol=1e-4)


def test_SYMEIG_rank_deficient(eng):
    from linalg.eig_sym import SYMEIG
    m, r = 50, 10
    g = torch.Generator().manual_seed(2)
    M = torch.rand((m, m), generator=g, dtype=torch.float64)
    M = M + M.t()
    D, U = torch.linalg.eigh(M)
    D[-r:] = 0
    M = (U @ torch.diag(D) @ U.t()).cuda()
    D, U = SYMEIG.apply(M, 1.0e-12)
    assert torch.norm(M - U @ torch.diag(D) @ U.t()) < D[0].abs() * (m ** 2) * 1e-14
    assert int((D.abs() < 1e-12 * D[0].abs()).sum()) == r
    I = torch.eye(m, dtype=torch.float64, device=U.device)
    assert float((U.t() @ U - I).abs().max()) < 1e-12


def test_SVDGESDD_random(eng):
    from linalg.svd_gesdd import SVDGESDD
    g = torch.Generator().manual_seed(3)
    A = torch.rand(40, 40, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a: SVDGESDD.apply(a, 1.0e-12), (A,), eps=1e-6, atol=1e-5)


def test_SVDGESDD_COMPLEX_random(eng):
    from linalg.svd_gesdd import SVDGESDD

    def test_f_1(M):
        U, S, V = SVDGESDD.apply(M, 1.0e-12)
        return torch.sum(S[0:1])

    def test_f_2(M):
        U, S, V = SVDGESDD.apply(M, 1.0e-12)
        T = U @ V.conj().transpose(-2, -1)
        return T.norm()

    m = 25
    g = torch.Generator().manual_seed(4)
    A = torch.rand((m, m), generator=g, dtype=torch.float64) + 1j * torch.rand((m, m), generator=g, dtype=torch.float64)
    U, S, Vh = torch.linalg.svd(A)
    for split_scale in [10.0, 1.0, 0.1, 0.01, 0.]:
        tot_scale = 1000
        d0 = torch.rand(m // 2, generator=g, dtype=torch.float64)
        splits = torch.rand(m // 2, generator=g, dtype=torch.float64)
        S = S.clone()
        for i in range(m // 2):
            S[2 * i] = tot_scale * d0[i]
            S[2 * i + 1] = tot_scale * d0[i] + split_scale * splits[i]
        A = ((U * S.to(U.dtype)) @ Vh).cuda().requires_grad_(True)
        if split_scale > 0.:      # S[0] of an exactly degenerate leading pair is not differentiable (the reference's own run of this
            assert torch.autograd.gradcheck(test_f_1, A, eps=1e-6, atol=1e-4), split_scale      # case fails there too; its __main__ skips the test)
        assert torch.autograd.gradcheck(test_f_2, A, eps=1e-6, atol=1e-4), split_scale


@pytest.mark.parametrize("case", ["random", "3x3degenerate", "rank_deficient"])
def test_SVDSYMEIG(eng, case):
    from linalg.custom_svd import truncated_svd_symeig
    g = torch.Generator().manual_seed(5)
    if case == "random":
        M = torch.rand(50, 50, generator=g, dtype=torch.float64); M = 0.5 * (M + M.t())
    elif case == "3x3degenerate":
        M = torch.zeros((3, 3), dtype=torch.float64); M[0, 1] = M[0, 2] = M[1, 2] = 1.; M = 0.5 * (M + M.t())
    else:
        M = torch.rand((50, 50), generator=g, dtype=torch.float64); M = M + M.t()
        D, U = torch.linalg.eigh(M); D[-10:] = 0; M = U @ torch.diag(D) @ U.t()
    M = M.cuda()
    m = M.shape[0]
    U, S, V = truncated_svd_symeig(M, m)
    assert torch.norm(M - U @ torch.diag(S) @ V.t()) < S[0] * (m ** 2) * 1e-14
    assert float(S.min()) >= 0.0 and float((S[:-1] - S[1:]).min()) >= -1e-14 * float(S[0])
