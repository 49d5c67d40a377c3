"""truncated_eig_sym with the reference's signature (linalg/custom_eig.py:7-65; eig_sym.py:25-34):
eigh of the lower triangle, ordered by |D| descending, multiplet-aware truncation.  Forward only."""
from backend import get_engine


def truncated_eig_sym(M, chi, abs_tol=1.0e-14, rel_tol=None, ad_decomp_reg=1.0e-12,
                      keep_multiplets=False, eps_multiplet=1.0e-12, verbosity=0):
    eng = get_engine()
    cfg = eng.cfg(eps_multiplet=eps_multiplet, multiplet_abstol=abs_tol, keep_multiplets=keep_multiplets)
    return eng.truncated_eigh(M, chi, cfg)
