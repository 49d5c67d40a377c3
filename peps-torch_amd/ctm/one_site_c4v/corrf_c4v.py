"""Transfer-matrix correlators of the one-site C4v network (reference ctm/one_site_c4v/corrf_c4v.py:5-41, 85-143, 178-271, 593-664).

An edge is a chi x D^2 x chi tensor; one transfer step contracts it with T, the site (layer by layer, an operator optionally
inserted between the layers) and T again -- one native contraction (`ctm_einsum`) per step, nothing of size D^8 is formed."""
import torch
from linalg.native_einsum import einsum


def _parts(state, env):
    return next(iter(state.sites.values())), env.C[env.keyC], env.T[env.keyT]


def get_edge(state, env, verbosity=0):
    """E = C - T - C (:5-41): C[x,a] T[x,b,s] C[b,c] -> [a, s, c]."""
    a, C, T = _parts(state, env)
    return einsum('xa,xbs,bc->asc', C, T, C)


def apply_edge(state, env, vec, verbosity=0):
    """Scalar vec . (C - T - C) (:85-143)."""
    a, C, T = _parts(state, env)
    r = einsum('asc,ad,eds->ce', vec, C, T)                     # then the closing corner: sum_ce r[c,e] C[e,c]
    return (r * C.t()).sum()


def apply_TM_1sO(state, env, edge, op=None, verbosity=0):
    """One transfer step T - (a^+ op a) - T applied to `edge` (:178-271); same index structure out.  `op[m, n]` sits between the
    ket index m and the bra index n exactly as in the reference's 'mefgh,mn,nabcd->eafbgchd'."""
    a, C, T = _parts(state, env)
    chi, D = T.shape[0], a.shape[1]
    e4 = edge.reshape(chi, D, D, chi)
    Tv = T.reshape(chi, chi, D, D)
    if op is None:
        r = einsum('alLc,xauU,suldr,sULDR,cydD->xrRy', e4, Tv, a, a, Tv, conj=(3,))
    else:
        if op.dim() != 2:
            raise ValueError("apply_TM_1sO: op must be a matrix")
        r = einsum('alLc,xauU,muldr,mn,nULDR,cydD->xrRy', e4, Tv, a, op.to(dtype=a.dtype, device=a.device), a, Tv, conj=(4,))
    return r.reshape(chi, D * D, chi)


def corrf_1sO1sO(state, env, op1, get_op2, dist, rl_0=None, verbosity=0):
    """<op1(0) op2(r)> / <1> for r = 0 .. dist along a row (:593-664); `rl_0` = (right, left) leading eigenvectors of the
    width-1 transfer matrix replaces the C - T - C boundaries."""
    E0 = get_edge(state, env) if rl_0 is None else rl_0[0]
    E1 = apply_TM_1sO(state, env, E0, op=op1)
    E0 = apply_TM_1sO(state, env, E0)
    corrf = torch.empty(dist + 1, dtype=E0.dtype, device=E0.device)
    for r in range(dist + 1):
        E12 = apply_TM_1sO(state, env, E1, op=get_op2(r))
        E0 = apply_TM_1sO(state, env, E0)
        E1 = apply_TM_1sO(state, env, E1)
        if rl_0 is None:
            n12, n00 = apply_edge(state, env, E12), apply_edge(state, env, E0)
        else:
            n12, n00 = (E12 * rl_0[1]).sum(), (E0 * rl_0[1]).sum()
        corrf[r] = n12 / n00
        m = E0.abs().max()
        E0, E1 = E0 / m, E1 / m
    return corrf
