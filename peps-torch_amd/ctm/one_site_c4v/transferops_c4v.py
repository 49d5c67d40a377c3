"""Spectrum of the width-1 transfer operator of the one-site C4v network (reference ctm/one_site_c4v/transferops_c4v.py:10-68).
ARPACK (scipy, host) drives the iteration; every matrix-vector product is one native transfer step (corrf_c4v.apply_TM_1sO)."""
import numpy as np
import torch
from scipy.sparse.linalg import LinearOperator, eigs
from ctm.one_site_c4v import corrf_c4v


def get_Top_spec_c4v(n, state, env_c4v, normalize=True, eigenvectors=False, verbosity=0):
    """Leading n eigenvalues by modulus (normalised to |lambda_0| = 1 unless normalize=False) as an n x 2 tensor (re, im)."""
    chi = env_c4v.chi
    ad = next(iter(state.sites.values())).size(4)
    dev, dt = env_c4v.device, env_c4v.dtype

    def _mv(v):
        V = torch.as_tensor(np.ascontiguousarray(v), device=dev).to(dt).view(chi, ad * ad, chi)
        V = corrf_c4v.apply_TM_1sO(state, env_c4v, V)
        return V.reshape(-1).detach().cpu().numpy()

    dim = chi * ad * ad * chi
    T = LinearOperator((dim, dim), matvec=_mv, dtype="complex128" if dt.is_complex else "float64")
    if eigenvectors:
        vals, vecs = eigs(T, k=n, v0=None, return_eigenvectors=True)
    else:
        vals = eigs(T, k=n, v0=None, return_eigenvectors=False)
    order = np.argsort(np.abs(vals))[::-1]
    vals = vals[order]
    if normalize:
        vals = (1.0 / np.abs(vals[0])) * vals
    L = torch.zeros((n, 2), dtype=torch.float64, device=state.device)
    L[:, 0] = torch.as_tensor(np.real(vals))
    L[:, 1] = torch.as_tensor(np.imag(vals))
    if eigenvectors:
        return L, torch.as_tensor(vecs[:, order], device=state.device)
    return L
