"""Transfer-matrix correlation functions (reference ctm/generic/corrf.py:10-103,234-276,364-670,980-1067).

An edge is a (chi, D^2, chi) tensor with its legs ordered left-to-right / top-to-bottom (reference corrf.py:29-41);
`apply_TM_1sO` pushes it one site along `direction` through T . (a^+ op a) . T.  Every contraction runs on the native
executor (`Engine.einsum`), layer by layer -- the double-layer site tensor of the reference is never formed."""
import torch
import config as cfg
from backend import get_engine
from linalg.native_einsum import einsum as _einsum, needs_grad as _needs_grad

_UP, _LEFT, _DOWN, _RIGHT = (0, -1), (-1, 0), (0, 1), (1, 0)

# get_edge: E[a,b,c] = C_first . T . C_second in reading order (corrf.py:43-101)
_EDGE = {
    _UP: (((-1, -1), _UP, (1, -1)), "ax,xby,yc->abc"),
    _LEFT: (((-1, -1), _LEFT, (-1, 1)), "xa,xyb,yc->abc"),
    _DOWN: (((-1, 1), _DOWN, (1, 1)), "ax,bxy,cy->abc"),
    _RIGHT: (((1, -1), _RIGHT, (1, 1)), "ax,xby,yc->abc"),
}
# apply_TM_1sO: (rel. vectors of T1, T2), network over (T1, edge, a.op, conj(a), T2) with every D^2 leg split (ket,bra);
# site a[s,u,l,d,r]; capital letters = bra layer (corrf.py:449-667)
_TM = {
    _UP: ((_LEFT, _RIGHT), "axlL,xdDy,suldr,sULDR,crRy->auUc"),
    _LEFT: ((_UP, _DOWN), "auUx,xrRy,suldr,sULDR,dDcy->alLc"),
    _DOWN: ((_LEFT, _RIGHT), "xalL,xuUy,suldr,sULDR,yrRc->adDc"),
    _RIGHT: ((_UP, _DOWN), "xuUa,xlLy,suldr,sULDR,dDyc->arRc"),
}
# position of the D^2 axis of each T tensor (env.py:57-76) and the site leg it attaches to
_TAXIS = {_UP: (1, 1), _LEFT: (2, 2), _DOWN: (0, 3), _RIGHT: (1, 4)}


def _split(t, axis, D):
    s = list(t.shape)
    return t.reshape(s[:axis] + [D, D] + s[axis + 1:])


def _dot3(v, E):
    """sum_abc v_abc E_abc: one native contraction; with tensors that require grad the (chi D^2 chi-element) product in torch, whose
    adjoint autograd knows (a contraction node with a scalar output has no adjoint network)."""
    if _needs_grad(v, E):
        return (v * E).sum()
    return get_engine().einsum("abc,abc->", v.contiguous(), E.contiguous())


def get_edge(coord, direction, state, env, verbosity=0):
    if direction not in _EDGE:
        raise ValueError("Invalid direction: " + str(direction))
    c = state.vertexToSite(coord)
    (c1, t, c2), expr = _EDGE[direction]
    return _einsum(expr, env.C[(c, c1)], env.T[(c, t)], env.C[(c, c2)])


def apply_edge(coord, direction, state, env, vec, verbosity=0):
    if vec.dim() != 3:
        raise ValueError("Unsupported edge: " + str(tuple(vec.shape)))
    E = get_edge(coord, direction, state, env)
    return _dot3(vec, E)


def apply_TM_1sO(coord, direction, state, env, edge, op=None, verbosity=0):
    """edge -> edge . T1 . (a^+ op a) . T2 of site `coord` (op = None: identity)."""
    if direction not in _TM:
        raise ValueError("Invalid direction: " + str(direction))
    if edge.dim() != 3 or (op is not None and op.dim() != 2):
        raise NotImplementedError("apply_TM_1sO: MPO-carrying edges / operators are not on the native path")
    c = state.vertexToSite(coord)
    a = state.site(c)
    (v1, v2), expr = _TM[direction]
    ket = a if op is None else torch.einsum('mefgh,mn->nefgh', a, op.to(dtype=a.dtype, device=a.device)).contiguous()
    ax1, leg1 = _TAXIS[v1]
    ax2, leg2 = _TAXIS[v2]
    T1 = _split(env.T[(c, v1)], ax1, a.shape[leg1])
    T2 = _split(env.T[(c, v2)], ax2, a.shape[leg2])
    in_leg = {_UP: 3, _LEFT: 4, _DOWN: 1, _RIGHT: 2}[direction]          # site leg facing the incoming edge
    out_leg = {_UP: 1, _LEFT: 2, _DOWN: 3, _RIGHT: 4}[direction]
    E = _split(edge.contiguous(), 1, a.shape[in_leg])
    out = _einsum(expr, T1, E, ket, a, T2, conj=(3,))
    return out.reshape(out.shape[0], a.shape[out_leg] ** 2, out.shape[3])


def corrf_1sO1sO(coord, direction, state, env, op1, get_op2, dist, rl_0=None, verbosity=0):
    """<O1(0) O2(r)> for r = 1..dist+1 along `direction` (reference corrf.py:980-1067, same normalisation steps).
    rl_0 = (right(c), left(c)): optional functions returning the rank-3 boundary edges at coordinate c (typically leading
    eigenvectors of the transfer operator) used instead of the corner-T-corner edges."""
    eng = get_engine()
    c0 = coord
    rev = (-direction[0], -direction[1])
    E0 = get_edge(c0, rev, state, env) if rl_0 is None else rl_0[0](c0).contiguous()
    close = (lambda c, v: apply_edge(c, direction, state, env, v)) if rl_0 is None else \
        (lambda c, v: _dot3(v, rl_0[1](c)))
    E1 = apply_TM_1sO(c0, direction, state, env, E0, op=op1)
    E0 = apply_TM_1sO(c0, direction, state, env, E0)
    corrf = torch.empty(dist + 1, dtype=state.dtype, device=state.device)
    for r in range(dist + 1):
        c0 = (c0[0] + direction[0], c0[1] + direction[1])
        E12 = apply_TM_1sO(c0, direction, state, env, E1, op=get_op2(r))
        E0 = apply_TM_1sO(c0, direction, state, env, E0)
        E1 = apply_TM_1sO(c0, direction, state, env, E1)
        corrf[r] = close(c0, E12) / close(c0, E0)
        m = E0.abs().max()
        E0 = E0 / m
        E1 = E1 / m
    return corrf
