"""truncated_svd_gesdd with the reference's signature (linalg/custom_svd.py:38-101), executed by the
native block-Jacobi SVD (full-decomposition semantics: exact leading chi triplets, sign fix,
multiplet-aware truncation).  Forward only (no autograd)."""
from backend import get_engine


def truncated_svd_gesdd(M, chi, abs_tol=1.0e-14, rel_tol=None, ad_decomp_reg=1.0e-12,
                        keep_multiplets=False, eps_multiplet=1.0e-12, verbosity=0, diagnostics=None):
    eng = get_engine()
    cfg = eng.cfg(eps_multiplet=eps_multiplet, multiplet_abstol=abs_tol, keep_multiplets=keep_multiplets)
    return eng.truncated_svd(M, chi, cfg)


def truncated_svd_symeig(M, chi, abs_tol=1.0e-14, rel_tol=None, keep_multiplets=False, eps_multiplet=1.0e-12, verbosity=0):
    """Leading chi singular triplets of a SYMMETRIC matrix from its eigendecomposition M = U D U^T: S = |D|, V = U sign(D)
    (reference linalg/custom_svd.py:143-208, svd_symeig.py:12-34), with the same multiplet-aware truncation.  Forward only."""
    eng = get_engine()
    cfg = eng.cfg(eps_multiplet=eps_multiplet, multiplet_abstol=abs_tol, keep_multiplets=keep_multiplets)
    return eng.svd_symeig(M, chi, cfg)
