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


def apply_TM_2sO(state, env, edge, op=None, verbosity=0):
    """Two transfer steps with a two-site operator op[s1, s2, s1', s2'] inserted (:435-592): op is split by an SVD of its
    (s1 s1') x (s2 s2') matrix into op_l[m, n, q] and op_r[m, n, q] (tiny, on the host), the bond q travels with the edge between
    the two steps."""
    a, C, T = _parts(state, env)
    chi, D, p = T.shape[0], a.shape[1], a.shape[0]
    if op is not None:
        if op.dim() != 4:
            raise ValueError("apply_TM_2sO: op must have four physical legs")
        m = op.detach().cpu().permute(0, 2, 1, 3).contiguous().reshape(p * p, p * p)
        U, S, Vh = torch.linalg.svd(m)
        op_l = U.reshape(p, p, S.shape[0])
        op_r = (S[:, None] * Vh).reshape(S.shape[0], p, p).permute(1, 2, 0).contiguous()
    else:
        op_l = op_r = torch.eye(p, dtype=a.dtype)[:, :, None]
    op_l = op_l.to(dtype=a.dtype, device=a.device).contiguous()
    op_r = op_r.to(dtype=a.dtype, device=a.device).contiguous()
    e4 = edge.reshape(chi, D, D, chi)
    Tv = T.reshape(chi, chi, D, D)
    e5 = einsum('alLc,xauU,muldr,mnq,nULDR,cydD->xrRqy', e4, Tv, a, op_l, a, Tv, conj=(4,))
    r = einsum('alLqc,xauU,muldr,mnq,nULDR,cydD->xrRy', e5, Tv, a, op_r, a, Tv, conj=(4,))
    return r.reshape(chi, D * D, chi)


def corrf_2sOH2sOH_E1(state, env, op1, get_op2, dist, verbosity=0):
    """<op1(0,1) op2(r+2, r+3)> / <1> of two horizontal two-site operators along a row, r = 0 .. dist (:666-737)."""
    E0 = get_edge(state, env)
    E1 = apply_TM_2sO(state, env, E0, op=op1)
    E0 = apply_TM_2sO(state, env, E0)
    corrf = torch.empty(dist + 1, dtype=E0.dtype, device=E0.device)
    for r in range(dist + 1):
        E12 = apply_TM_2sO(state, env, E1, op=get_op2(r))
        E0 = apply_TM_1sO(state, env, E0)
        E1 = apply_TM_1sO(state, env, E1)
        n12 = apply_edge(state, env, E12)
        n00 = apply_edge(state, env, apply_TM_1sO(state, env, E0))
        corrf[r] = n12 / n00
        m = E0.abs().max()
        E0, E1 = E0 / m, E1 / m
    return corrf


def get_edge_L(state, env, l=1, verbosity=0):
    """Edge of length l: C - T - ... - T - C with l open D^2 legs (:43-83)."""
    a, C, T = _parts(state, env)
    if l == 1:
        return einsum('ab,xbs,xe->ase', C, T, C)
    if l == 2:
        return einsum('ab,xbs,zxt,ze->aste', C, T, T, C)
    raise NotImplementedError("get_edge_L: l = 1, 2")


def apply_edge_L(state, env, vec, verbosity=0):
    """Scalar vec . edge_L with the legs of the edge in reverse order (:145-176)."""
    l = vec.dim() - 2
    E = get_edge_L(state, env, l=l)
    return (vec * E.permute(*range(l + 1, -1, -1))).sum()


def _split_two_site_op(op, p, dtype, device):
    """op[s1, s2, s1', s2'] -> op_1[m, n, q], op_2[m, n, q] by an SVD of the (s1 s1') x (s2 s2') matrix (tiny, host)."""
    if op is None:
        e = torch.eye(p, dtype=dtype, device=device)[:, :, None].contiguous()
        return e, e
    if op.dim() != 4:
        raise ValueError(f"Invalid op: rank {op.dim()}")
    m = op.detach().cpu().permute(0, 2, 1, 3).contiguous().reshape(p * p, p * p)
    U, S, Vh = torch.linalg.svd(m)
    o1 = U.reshape(p, p, S.shape[0])
    o2 = (S[:, None] * Vh).reshape(S.shape[0], p, p).permute(1, 2, 0).contiguous()
    return o1.to(dtype=dtype, device=device).contiguous(), o2.to(dtype=dtype, device=device).contiguous()


def apply_TM_1sO_2(state, env, edge, op=None, verbosity=0):
    """One step of the width-2 transfer matrix T - (a^+ o1 a) - (a^+ o2 a) - T applied to an edge chi x D^2 x D^2 x chi (:273-433);
    a two-site operator op[s1, s2, s1', s2'] (upper site first) is split into o1 - o2.  Three native contractions, the operator folded
    into the conjugated layer first so that no intermediate exceeds eight legs."""
    a, C, T = _parts(state, env)
    chi, D, p = T.shape[0], a.shape[1], a.shape[0]
    o1, o2 = _split_two_site_op(op, p, a.dtype, a.device)
    b1 = einsum('mnq,nULDR->mULDRq', o1, a, conj=(1,))            # bra layer of the upper site with o1 (q: the bond of the split)
    b2 = einsum('vwq,wDKEF->vDKEFq', o2, a, conj=(1,))
    Tv = T.reshape(chi, chi, D, D)
    e5 = edge.reshape(chi, D, D, D * D, chi)                      # [a, l, L, z = (k K), c]
    s1 = einsum('xauU,alLzc,muldr,mULDRq->xrRcqdDz', Tv, e5, a, b1)
    s1 = s1.reshape(chi * D * D, chi, b1.shape[-1], D, D, D, D)   # [(x r R), c, q, d, D, k, K]
    s2 = einsum('XcqdDkK,vdkef,vDKEFq,cyeE->XfFy', s1, a, b2, Tv)
    return s2.reshape(chi, D * D, D * D, chi)


def corrf_2sOV2sOV_E2(state, env, op1, get_op2, dist, verbosity=0):
    """<op1(0) op2(r)> / <1> of two VERTICAL two-site operators r columns apart, width-2 channel, r = 0 .. dist (:739-807)."""
    E0 = get_edge_L(state, env, l=2)
    E1 = apply_TM_1sO_2(state, env, E0, op=op1)
    E0 = apply_TM_1sO_2(state, env, E0)
    corrf = torch.empty(dist + 1, dtype=E0.dtype, device=E0.device)
    for r in range(dist + 1):
        E12 = apply_TM_1sO_2(state, env, E1, op=get_op2(r))
        E0 = apply_TM_1sO_2(state, env, E0)
        E1 = apply_TM_1sO_2(state, env, E1)
        corrf[r] = apply_edge_L(state, env, E12) / apply_edge_L(state, env, E0)
        m = E0.abs().max()
        E0, E1 = E0 / m, E1 / m
    return corrf
