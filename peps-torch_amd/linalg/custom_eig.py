"""truncated_eig_sym with the reference's signature (linalg/custom_eig.py:7-65; eig_sym.py:25-34):
eigh of the lower triangle, ordered by |D| descending, multiplet-aware truncation.

Forward only: one native call (leading-chi solver).  When M requires grad: the reference's own route -- the FULL decomposition
through the differentiable SYMEIG (native forward and regularised backward, eig_sym.py:57-75), then the truncation by slicing."""
import torch
from backend import get_engine
from linalg.native_einsum import needs_grad


def _multiplet_chi(D, chi, eps_multiplet, abs_tol):
    """custom_eig.py:36-52: index (inclusive) of the last kept value so that no multiplet of |D| is cut at chi."""
    S = D.detach().abs().cpu().double()
    g = S[:chi + 1].clone()
    g[g < abs_tol] = 0.
    gaps = (g[:chi] - S[1:chi + 1]) / (g[:chi] + 1.0e-16)
    gaps[gaps > 1.0] = 0.
    chi_new = chi
    if gaps[chi - 1] < eps_multiplet:
        for i in range(chi - 1, -1, -1):
            if gaps[i] > eps_multiplet:
                chi_new = i
                break
    return chi_new


def truncated_eig_sym(M, chi, abs_tol=1.0e-14, rel_tol=None, ad_decomp_reg=1.0e-12,
                      keep_multiplets=False, eps_multiplet=1.0e-12, verbosity=0, basis=None):
    """basis (differentiable route only): warm-start workspace of the full decomposition, see SYMEIG.forward."""
    eng = get_engine()
    if needs_grad(M):
        from linalg.eig_sym import SYMEIG
        D, U = SYMEIG.apply(M, ad_decomp_reg, basis)
        n = D.shape[0]
        if keep_multiplets and chi < n:
            chi_new = _multiplet_chi(D, chi, eps_multiplet, abs_tol)
            mask = torch.zeros(chi, dtype=D.dtype, device=D.device)
            mask[:chi_new + 1] = 1.0
            return D[:chi] * mask, U[:, :chi] * mask.to(U.dtype)[None, :]
        k = min(chi, n)
        return D[:k], U[:, :k]
    cfg = eng.cfg(eps_multiplet=eps_multiplet, multiplet_abstol=abs_tol, keep_multiplets=keep_multiplets)
    return eng.truncated_eigh(M, chi, cfg)
