"""truncated_svd_gesdd with the reference's signature (linalg/custom_svd.py:38-101), executed by the
native block-Jacobi SVD (full-decomposition semantics: exact leading chi triplets, sign fix,
multiplet-aware truncation).  With a matrix that requires grad: the differentiable full decomposition."""
import torch
from backend import get_engine
from linalg.native_einsum import needs_grad


def truncated_svd_gesdd(M, chi, abs_tol=1.0e-14, rel_tol=None, ad_decomp_reg=1.0e-12,
                        keep_multiplets=False, eps_multiplet=1.0e-12, verbosity=0, diagnostics=None, basis=None):
    """basis (differentiable route only): warm-start workspace of the full decomposition, see SVDGESDD.forward."""
    eng = get_engine()
    if needs_grad(M):
        # the reference's own route (custom_svd.py:38-101): FULL decomposition through the differentiable SVDGESDD (native forward
        # with fix_svd_signs, regularised native backward), truncation by slicing / zeroing the cut multiplet
        from linalg.svd_gesdd import SVDGESDD
        from linalg.custom_eig import _multiplet_chi
        U, S, V = SVDGESDD.apply(M, ad_decomp_reg, None, basis)
        if keep_multiplets and chi < S.shape[0]:
            chi_new = _multiplet_chi(S, chi, eps_multiplet, abs_tol)
            mask = torch.zeros(chi, dtype=S.dtype, device=S.device)
            mask[:chi_new + 1] = 1.0
            return U[:, :chi] * mask.to(U.dtype)[None, :], S[:chi] * mask, V[:, :chi] * mask.to(V.dtype)[None, :]
        k = min(chi, S.shape[0])
        return U[:, :k], S[:k], V[:, :k]
    cfg = eng.cfg(eps_multiplet=eps_multiplet, multiplet_abstol=abs_tol, keep_multiplets=keep_multiplets)
    return eng.truncated_svd(M, chi, cfg)


def truncated_svd_symeig(M, chi, abs_tol=1.0e-14, rel_tol=None, keep_multiplets=False, eps_multiplet=1.0e-12, verbosity=0):
    """Leading chi singular triplets of a SYMMETRIC matrix from its eigendecomposition M = U D U^T: S = |D|, V = U sign(D)
    (reference linalg/custom_svd.py:143-208, svd_symeig.py:12-34), with the same multiplet-aware truncation.  Forward only."""
    eng = get_engine()
    cfg = eng.cfg(eps_multiplet=eps_multiplet, multiplet_abstol=abs_tol, keep_multiplets=keep_multiplets)
    return eng.svd_symeig(M, chi, cfg)
