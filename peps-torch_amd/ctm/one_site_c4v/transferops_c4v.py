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


def get_Top2_spec_c4v(n, state, env_c4v, verbosity=0):
    """Leading n eigenvalues of the width-2 transfer operator T - aa - aa - T (:70-117), normalised to |lambda_0| = 1; ordered as
    ARPACK returns them reversed, like the reference."""
    chi = env_c4v.chi
    ad = next(iter(state.sites.values())).size(4)
    dev, dt = env_c4v.device, env_c4v.dtype

    def _mv(v):
        V = torch.as_tensor(np.ascontiguousarray(v), device=dev).to(dt).view(chi, ad * ad, ad * ad, chi)
        V = corrf_c4v.apply_TM_1sO_2(state, env_c4v, V)
        return V.reshape(-1).detach().cpu().numpy()

    dim = chi * (ad ** 4) * chi
    T = LinearOperator((dim, dim), matvec=_mv, dtype="complex128" if dt.is_complex else "float64")
    vals = eigs(T, k=n, v0=None, return_eigenvectors=False)
    vals = np.copy(vals[::-1])
    vals = (1.0 / np.abs(vals[0])) * vals
    L = torch.zeros((n, 2), dtype=torch.float64, device=state.device)
    L[:, 0] = torch.as_tensor(np.real(vals))
    L[:, 1] = torch.as_tensor(np.imag(vals))
    return L


def get_EH_spec_Ttensor(n, L, state, env_c4v, verbosity=0):
    """Leading n eigenvalues of exp(EH) of an L-leg cylinder approximated by the periodic MPO of L half-row tensors T (:119-221):
    W[o_0 .. o_{L-1}] = sum  prod_k T[c_k, c_{k+1}, o_k, i_k]  V[i_0 .. i_{L-1}],  c_L = c_0.  One native contraction per product."""
    assert L > 1, "L must be larger than 1"
    if L > 6:
        raise NotImplementedError("get_EH_spec_Ttensor: L <= 6 (the running intermediate has L + 2 legs, engine limit 8)")
    chi = env_c4v.chi
    ad = next(iter(state.sites.values())).size(4)
    dev, dt = env_c4v.device, env_c4v.dtype
    T = env_c4v.get_T().view(chi, chi, ad, ad).contiguous()
    cs, os_, is_ = "abcdef"[:L], "mnopqr"[:L], "stuvwx"[:L]
    terms = [cs[k] + cs[(k + 1) % L] + os_[k] + is_[k] for k in range(L)]
    expr = ",".join([terms[0], is_] + terms[1:]) + "->" + os_

    def _mv(v0):
        V = torch.as_tensor(np.ascontiguousarray(v0), device=dev).to(dt).view([ad] * L)
        return corrf_c4v.einsum(expr, T, V, *([T] * (L - 1))).reshape(-1).detach().cpu().numpy()

    expEH = LinearOperator((ad ** L, ad ** L), matvec=_mv, dtype="complex128" if dt.is_complex else "float64")
    vals = eigs(expEH, k=n, v0=None, return_eigenvectors=False)
    vals = np.copy(vals[::-1])
    vals = (1.0 / np.abs(vals[0])) * vals
    S = torch.zeros((n, 2), dtype=torch.float64, device=state.device)
    S[:, 0] = torch.as_tensor(np.real(vals))
    S[:, 1] = torch.as_tensor(np.imag(vals))
    return S
