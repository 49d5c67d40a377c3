"""Spectrum of the width-0 transfer operator (reference ctm/generic/transferops.py:119-207).

ARPACK (scipy, host) drives the iteration; every matrix-vector product is lX (or lY) native transfer steps
(`corrf.apply_TM_1sO` on the engine), the vector travels host <-> HBM once per product (chi^2 D^2 numbers)."""
import numpy as np
import torch
from scipy.sparse.linalg import LinearOperator, eigs
from ctm.generic import corrf

_SITE_LEG = {(0, -1): 1, (-1, 0): 2, (0, 1): 3, (1, 0): 4}


def get_Top_spec(n, coord, direction, state, env, eigenvectors=False, verbosity=0):
    """Leading n eigenvalues (by modulus, normalised to |lambda_0| = 1) as an n x 2 tensor (real, imaginary parts)."""
    if direction not in _SITE_LEG:
        raise ValueError("Invalid direction: " + str(direction))
    c = state.vertexToSite(coord)
    chi = env.chi
    ad = state.site(c).size(_SITE_LEG[(-direction[0], -direction[1])])       # bond facing the incoming edge
    N = state.lX if direction in [(1, 0), (-1, 0)] else state.lY
    cplx = env.dtype.is_complex

    def _mv(v):
        c0 = coord
        V = torch.as_tensor(np.ascontiguousarray(v), device=state.device).to(env.dtype).view(chi, ad * ad, chi)
        for _ in range(N):
            V = corrf.apply_TM_1sO(c0, direction, state, env, V)
            c0 = (c0[0] + direction[0], c0[1] + direction[1])
        return V.reshape(-1).cpu().numpy()

    dim = chi * ad * ad * chi
    T = LinearOperator((dim, dim), matvec=_mv, dtype="complex128" if cplx else "float64")
    if eigenvectors:
        vals, vecs = eigs(T, k=n, v0=None, return_eigenvectors=True)
    else:
        vals = eigs(T, k=n, v0=None, return_eigenvectors=False)
    order = np.argsort(np.abs(vals))[::-1]
    vals = vals[order]
    vals = (1.0 / np.abs(vals[0])) * vals
    L = torch.zeros((n, 2), dtype=torch.float64, device=state.device)
    L[:, 0] = torch.as_tensor(np.real(vals))
    L[:, 1] = torch.as_tensor(np.imag(vals))
    if eigenvectors:
        return L, torch.as_tensor(vecs[:, order], device=state.device)
    return L
