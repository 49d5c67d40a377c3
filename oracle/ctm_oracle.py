"""numpy restatement of the reference's *generic* directional CTMRG move and RDMs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function cites the reference
file:line (relative to /root/reference) whose behaviour it restates.  The restatement is
table driven (one einsum spec per corner / direction) instead of the reference's four
hand-written copies; it is pinned against the reference by oracle/gen_golden.py.

Conventions (ctm/generic/env.py:57-76, ipeps/ipeps.py:117-124):
  site   a[s,u,l,d,r]
  C(-1,-1)=(down,right) C(1,-1)=(left,down) C(1,1)=(up,left) C(-1,1)=(up,right)
  T(0,-1)=(left,down D^2,right) T(-1,0)=(up,down,right D^2)
  T(0,1)=(up D^2,left,right)    T(1,0)=(up,left D^2,down)
  fused double-layer leg = (ket, bra), ket first.
"""
import numpy as np

UP, LEFT, DOWN, RIGHT = (0, -1), (-1, 0), (0, 1), (1, 0)
DIRECTIONS = [UP, LEFT, DOWN, RIGHT]          # config.py:392 ctm_move_sequence
LU, RU, RD, LD = 0, 1, 2, 3

# ----------------------------------------------------------------------------------
# state / env containers (plain dicts of numpy arrays)
# ----------------------------------------------------------------------------------

class State:
    """Minimal IPEPS stand-in (ipeps/ipeps.py:89-238): sites + PBC vertexToSite."""

    def __init__(self, sites, lX=None, lY=None, vertexToSite=None):
        self.sites = dict(sites)
        xs = [c[0] for c in sites]
        ys = [c[1] for c in sites]
        self.lX = lX if lX else max(xs) - min(xs) + 1
        self.lY = lY if lY else max(ys) - min(ys) + 1
        if vertexToSite is None:
            def vertexToSite(c, lX=self.lX, lY=self.lY):      # ipeps/ipeps.py:233-238
                return ((c[0] + abs(c[0]) * lX) % lX, (c[1] + abs(c[1]) * lY) % lY)
        self.vertexToSite = vertexToSite

    def site(self, c):
        return self.sites[self.vertexToSite(c)]


class Env:
    """ENV stand-in (ctm/generic/env.py:14-108)."""

    def __init__(self, chi):
        self.chi = chi
        self.C = {}
        self.T = {}

    def clone(self):
        e = Env(self.chi)
        e.C = {k: v.copy() for k, v in self.C.items()}
        e.T = {k: v.copy() for k, v in self.T.items()}
        return e


def env_extend(env, new_chi):
    """ENV.extend (env.py:164-202): zero-padded copy with environment dimension new_chi (leading min(chi, new_chi) block kept)."""
    e = Env(new_chi)
    x = min(env.chi, new_chi)
    for k, c in env.C.items():
        e.C[k] = np.zeros((new_chi, new_chi), dtype=c.dtype); e.C[k][:x, :x] = c[:x, :x]
    for k, t in env.T.items():
        if k[1] in ((0, -1), (1, 0)):
            e.T[k] = np.zeros((new_chi, t.shape[1], new_chi), dtype=t.dtype); e.T[k][:x, :, :x] = t[:x, :, :x]
        elif k[1] == (-1, 0):
            e.T[k] = np.zeros((new_chi, new_chi, t.shape[2]), dtype=t.dtype); e.T[k][:x, :x, :] = t[:x, :x, :]
        elif k[1] == (0, 1):
            e.T[k] = np.zeros((t.shape[0], new_chi, new_chi), dtype=t.dtype); e.T[k][:, :x, :x] = t[:, :x, :x]
        else:
            raise ValueError(f"Unexpected direction {k[1]}")
    return e


def _nrm(a):
    return a / np.abs(a).max()


def seq_einsum(expr, *ops):
    """einsum evaluated strictly left to right as pairwise contractions (each one a BLAS
    tensordot) -- the same contraction ORDER the reference uses (torch contracts its
    einsums left to right when opt_einsum is absent)."""
    lhs, out = expr.split('->')
    ins = lhs.split(',')
    cur, cidx = ops[0], ins[0]
    for k in range(1, len(ins)):
        later = set(out).union(*[set(x) for x in ins[k + 1:]]) if k + 1 < len(ins) else set(out)
        nidx = ''.join(dict.fromkeys([c for c in cidx + ins[k] if c in later]))
        cur = np.einsum(f"{cidx},{ins[k]}->{nidx}", cur, ops[k], optimize=True)
        cidx = nidx
    if cidx != out:
        cur = np.einsum(f"{cidx}->{out}", cur)
    return cur


def init_env_ctmrg(state, chi):
    """init_from_ipeps_pbc (ctm/generic/env.py:367-536): every env tensor of `coord` is the
    partial trace of the double layer of the NEIGHBOUR in direction vec, divided by its
    max-abs and zero padded to chi."""
    env = Env(chi)
    cspec = {(-1, -1): ('mijef,mijab->eafb', 3, 4), (1, -1): ('miefj,miabj->eafb', 2, 3),
             (1, 1): ('mefij,mabij->eafb', 1, 2), (-1, 1): ('meijf,maijb->eafb', 1, 4)}
    tspec = {(0, -1): ('miefg,miabc->eafbgc', (2, 3, 4)), (-1, 0): ('meifg,maibc->eafbgc', (1, 3, 4)),
             (0, 1): ('mefig,mabic->eafbgc', (1, 2, 4)), (1, 0): ('mefgi,mabci->eafbgc', (1, 2, 3))}
    for coord in state.sites:
        for vec, (expr, i0, i1) in cspec.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            d = A.shape
            a = _nrm(np.einsum(expr, A, A.conj()).reshape(d[i0] ** 2, d[i1] ** 2))
            C = np.zeros((chi, chi), dtype=A.dtype)
            m0, m1 = min(chi, d[i0] ** 2), min(chi, d[i1] ** 2)
            C[:m0, :m1] = a[:m0, :m1]
            env.C[(coord, vec)] = C
        for vec, (expr, ii) in tspec.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            d = A.shape
            a = _nrm(np.einsum(expr, A, A.conj()).reshape(d[ii[0]] ** 2, d[ii[1]] ** 2, d[ii[2]] ** 2))
            if vec == (0, -1):
                T = np.zeros((chi, d[3] ** 2, chi), dtype=A.dtype)
                m0, m2 = min(chi, d[2] ** 2), min(chi, d[4] ** 2)
                T[:m0, :, :m2] = a[:m0, :, :m2]
            elif vec == (-1, 0):
                T = np.zeros((chi, chi, d[4] ** 2), dtype=A.dtype)
                m0, m1 = min(chi, d[1] ** 2), min(chi, d[3] ** 2)
                T[:m0, :m1, :] = a[:m0, :m1, :]
            elif vec == (0, 1):
                T = np.zeros((d[1] ** 2, chi, chi), dtype=A.dtype)
                m1, m2 = min(chi, d[2] ** 2), min(chi, d[4] ** 2)
                T[:, :m1, :m2] = a[:, :m1, :m2]
            else:
                T = np.zeros((chi, d[2] ** 2, chi), dtype=A.dtype)
                m0, m2 = min(chi, d[1] ** 2), min(chi, d[3] ** 2)
                T[:m0, :, :m2] = a[:m0, :, :m2]
            env.T[(coord, vec)] = T
    return env


def init_env_prod(state, chi):
    """init_prod (ctm/generic/env.py:274-365), 5-leg sites: C = e_00; T of direction vec = partial trace of the neighbour's double
    layer over all legs but the one pointing back, placed at T[0,:,0] / T[0,0,:] / T[:,0,0]; the (0,-1) one is NOT normalised."""
    env = Env(chi)
    spec = {(0, -1): ('miefg,miebg->fb', lambda n: (chi, n, chi), (0, slice(None), 0), False),
            (-1, 0): ('meifg,meifc->gc', lambda n: (chi, chi, n), (0, 0, slice(None)), True),
            (0, 1): ('mefig,mafig->ea', lambda n: (n, chi, chi), (slice(None), 0, 0), True),
            (1, 0): ('mefgi,mebgi->fb', lambda n: (chi, n, chi), (0, slice(None), 0), True)}
    for coord in state.sites:
        for vec in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            C = np.zeros((chi, chi), dtype=state.site(coord).dtype); C[0, 0] = 1.0
            env.C[(coord, vec)] = C
        for vec, (expr, shape, where, normalise) in spec.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            a = np.einsum(expr, A, A.conj()).reshape(-1)
            if normalise:
                a = a / np.abs(a).max()
            T = np.zeros(shape(a.size), dtype=A.dtype)
            T[where] = a
            env.T[(coord, vec)] = T
    return env


def init_env_obc(state, chi):
    """init_from_ipeps_obc (ctm/generic/env.py:538-716), 5-leg sites: the outward legs of each layer summed separately, BOTH layers
    un-conjugated (`einsum('mijef,mklab->eafb', A, A)`), divided by the max-abs, zero padded to chi."""
    env = Env(chi)
    cspec = {(-1, -1): 'mijef,mklab->eafb', (1, -1): 'miefj,mkabl->eafb', (1, 1): 'mefij,mabkl->eafb', (-1, 1): 'meijf,maklb->eafb'}
    tspec = {(0, -1): 'miefg,mkabc->eafbgc', (-1, 0): 'meifg,makbc->eafbgc', (0, 1): 'mefig,mabkc->eafbgc', (1, 0): 'mefgi,mabck->eafbgc'}
    for coord in state.sites:
        for vec, expr in cspec.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            r = np.einsum(expr, A, A)
            a = _nrm(r.reshape(r.shape[0] * r.shape[1], r.shape[2] * r.shape[3]))
            C = np.zeros((chi, chi), dtype=A.dtype)
            m0, m1 = min(chi, a.shape[0]), min(chi, a.shape[1])
            C[:m0, :m1] = a[:m0, :m1]
            env.C[(coord, vec)] = C
        for vec, expr in tspec.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            r = np.einsum(expr, A, A)
            a = _nrm(r.reshape(r.shape[0] * r.shape[1], r.shape[2] * r.shape[3], r.shape[4] * r.shape[5]))
            if vec in ((0, -1), (1, 0)):
                T = np.zeros((chi, a.shape[1], chi), dtype=A.dtype); m0, m2 = min(chi, a.shape[0]), min(chi, a.shape[2])
                T[:m0, :, :m2] = a[:m0, :, :m2]
            elif vec == (-1, 0):
                T = np.zeros((chi, chi, a.shape[2]), dtype=A.dtype); m0, m1 = min(chi, a.shape[0]), min(chi, a.shape[1])
                T[:m0, :m1, :] = a[:m0, :m1, :]
            else:
                T = np.zeros((a.shape[0], chi, chi), dtype=A.dtype); m1, m2 = min(chi, a.shape[1]), min(chi, a.shape[2])
                T[:, :m1, :m2] = a[:, :m1, :m2]
            env.T[(coord, vec)] = T
    return env


# ----------------------------------------------------------------------------------
# enlarged corners  (ctm/generic/ctm_components.py:372-434, 532-586, 683-733, 832-884)
# ----------------------------------------------------------------------------------
# per corner: (C rel-vector, T1 rel-vector, T2 rel-vector, T1 split, T2 split, einsum)
#   T1/T2 "split": which axis of the T tensor is the fused D^2 leg and which site leg
#   (index into a[s,u,l,d,r]) it attaches to.
_CORNER = {
    LU: dict(C=(-1, -1), T1=(0, -1), T2=(-1, 0), s1=(1, 1), s2=(2, 2),
             closed='ab,bUVx,ayLM,sULdr,sVMDR->ydDxrR', open='ab,bUVx,ayLM,sULdr,tVMDR->ydDxrRst'),
    RU: dict(C=(1, -1), T1=(1, 0), T2=(0, -1), s1=(1, 4), s2=(1, 1),
             closed='ab,brRx,yuUa,suldr,sULDR->ylLxdD', open='ab,brRx,yuUa,suldr,tULDR->ylLxdDst'),
    RD: dict(C=(1, 1), T1=(0, 1), T2=(1, 0), s1=(0, 3), s2=(1, 4),
             closed='ab,dDxb,yrRa,suldr,sULDR->yuUxlL', open='ab,dDxb,yrRa,suldr,tULDR->yuUxlLst'),
    LD: dict(C=(-1, 1), T1=(-1, 0), T2=(0, 1), s1=(2, 2), s2=(0, 3),
             closed='ab,xalL,dDby,suldr,sULDR->xuUyrR', open='ab,xalL,dDby,suldr,tULDR->xuUyrRst'),
}


def _split(T, axis, D):
    sh = list(T.shape)
    return T.reshape(sh[:axis] + [D, D] + sh[axis + 1:])


def c2x2_tensors(corner, coord, state, env):
    """c2x2_*_t (ctm_components.py:314-319, 480-485, 632-637, 779-784)."""
    sp = _CORNER[corner]
    s = state.vertexToSite(coord)
    return env.C[(s, sp['C'])], env.T[(s, sp['T1'])], env.T[(s, sp['T2'])], state.site(coord)


def c2x2_sl(corner, C, T1, T2, a, open_=False):
    """c2x2_{LU,RU,RD,LD}_sl_c: C.T1.T2.a.conj(a) -> (chi D^2) x (chi D^2) [x p x p if open]."""
    sp = _CORNER[corner]
    T1v = _split(T1, sp['s1'][0], a.shape[sp['s1'][1]])
    T2v = _split(T2, sp['s2'][0], a.shape[sp['s2'][1]])
    r = seq_einsum(sp['open' if open_ else 'closed'], C, T1v, T2v, a, a.conj())
    sh = r.shape
    n0, n1 = sh[0] * sh[1] * sh[2], sh[3] * sh[4] * sh[5]
    return r.reshape((n0, n1) + tuple(sh[6:]))


def c2x2(corner, coord, state, env, open_=False):
    return c2x2_sl(corner, *c2x2_tensors(corner, coord, state, env), open_=open_)


# ----------------------------------------------------------------------------------
# halves (ctm_components.py:10-265).  per direction: (cornerA, shiftA, cornerB, shiftB, opA, opB)
# R = opA(cA) @ opB(cB) where op in {'N','T'}; index 0 of R is the truncated bond.
# ----------------------------------------------------------------------------------
_HALVES = {
    UP:    dict(R=(RU, (0, 0), RD, (0, 1), 'N', 'N'), Rt=(LU, (-1, 0), LD, (-1, 1), 'T', 'N')),   # :71-72
    LEFT:  dict(R=(LU, (0, 0), RU, (1, 0), 'N', 'N'), Rt=(LD, (0, 1), RD, (1, 1), 'N', 'T')),     # :135-136
    DOWN:  dict(R=(LD, (0, 0), LU, (0, -1), 'T', 'N'), Rt=(RD, (1, 0), RU, (1, -1), 'T', 'T')),   # :197-198
    RIGHT: dict(R=(RD, (0, 0), LD, (-1, 0), 'N', 'T'), Rt=(RU, (0, -1), LU, (-1, -1), 'T', 'T')), # :261-262
}


def _op(m, o):
    return m if o == 'N' else m.T          # plain transpose, never conjugated


def halves(direction, coord, state, env):
    out = []
    for key in ('R', 'Rt'):
        cA, sA, cB, sB, oA, oB = _HALVES[direction][key]
        A = c2x2(cA, (coord[0] + sA[0], coord[1] + sA[1]), state, env)
        B = c2x2(cB, (coord[0] + sB[0], coord[1] + sB[1]), state, env)
        out.append(_op(A, oA) @ _op(B, oB))
    return out[0], out[1]


# ----------------------------------------------------------------------------------
# truncated SVD (linalg/custom_svd.py:38-101, linalg/svd_gesdd.py:18-26,77-96)
# ----------------------------------------------------------------------------------

def fix_svd_signs(U, V):
    """svd_gesdd.py:18-26: per column divide U and V by the phase of the max-|U| entry
    (argmax on the int64-quantised |U|*2^40, first occurrence wins)."""
    amp = (np.abs(U) * (2.0 ** 40)).astype(np.int64)
    ii = np.argmax(amp, axis=0)
    ph = U[ii, np.arange(U.shape[1])]
    ph = ph / np.abs(ph)
    return U * ph.conj()[None, :], V * ph.conj()[None, :]


def multiplet_chi(S, chi, eps_multiplet, abs_tol):
    """custom_svd.py:70-86 (also custom_eig.py:39-54 on |D|): index of the last kept
    triplet (inclusive) so that no multiplet is cut."""
    S = np.abs(np.asarray(S))
    g = S[:chi + 1].copy()
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


def truncated_svd_gesdd(M, chi, abs_tol=1.0e-14, keep_multiplets=False, eps_multiplet=1.0e-12):
    U, S, Vh = np.linalg.svd(M)                      # LAPACK gesdd, full_matrices
    V = Vh.conj().T
    U, V = fix_svd_signs(U, V)
    if keep_multiplets and chi < S.shape[0]:
        chi_new = multiplet_chi(S, chi, eps_multiplet, abs_tol)
        St = S[:chi].copy(); St[chi_new + 1:] = 0.
        Ut = U[:, :chi].copy(); Ut[:, chi_new + 1:] = 0.
        Vt = V[:, :chi].copy(); Vt[:, chi_new + 1:] = 0.
        return Ut, St, Vt
    k = min(chi, S.shape[0])
    return U[:, :k], S[:k], V[:, :k]


def truncated_eig_sym(M, chi, abs_tol=1.0e-14, keep_multiplets=False, eps_multiplet=1.0e-12):
    """custom_eig.py:7-65 + eig_sym.py:25-34: eigh (lower triangle), order by |D| descending."""
    D, U = np.linalg.eigh(M)
    p = np.argsort(-np.abs(D), kind='stable')
    D, U = D[p], U[:, p]
    if keep_multiplets and chi < D.shape[0]:
        chi_new = multiplet_chi(D, chi, eps_multiplet, abs_tol)
        Dt = D[:chi].copy(); Dt[chi_new + 1:] = 0.
        Ut = U[:, :chi].copy(); Ut[:, chi_new + 1:] = 0.
        return Dt, Ut
    k = min(chi, D.shape[0])
    return D[:k], U[:, :k]


def truncated_svd_symeig(M, chi, abs_tol=1.0e-14, keep_multiplets=False, eps_multiplet=1.0e-12):
    """svd_symeig.py:12-34 (SVDSYMEIG.forward: eigh -> order by |D| descending -> S = |D|, V = U sign(D)) followed by the
    truncation of custom_svd.py:143-208.  torch.symeig (ascending, lower triangle) is restated with LAPACK eigh; the
    reference function itself cannot run on a current torch (torch.symeig was removed), so this restatement is pinned through
    its defining identities M = U S V^T, V = U sign(D) and through truncated_eig_sym (tests/test_oracle_golden.py)."""
    D, U = np.linalg.eigh(M)
    p = np.argsort(-np.abs(D), kind='stable')
    D, U = D[p], U[:, p]
    S, V = np.abs(D), U * np.sign(D)[None, :]
    if keep_multiplets and chi < S.shape[0]:
        chi_new = multiplet_chi(S, chi, eps_multiplet, abs_tol)
        St = S[:chi].copy(); St[chi_new + 1:] = 0.
        Ut = U[:, :chi].copy(); Ut[:, chi_new + 1:] = 0.
        Vt = V[:, :chi].copy(); Vt[:, chi_new + 1:] = 0.
        return Ut, St, Vt
    k = min(chi, S.shape[0])
    return U[:, :k], S[:k], V[:, :k]


def _safe_inverse(x, eps):
    return x / (x ** 2 + eps)


def svd_backward(U, S, V, gU, gS, gV, eps):
    """SVDGESDD.backward (linalg/svd_gesdd.py:209-328): gradient of A = U diag(S) V^H; thin factors get the (1 - U U^H),
    (1 - V V^H) terms, complex input the extra imaginary-diagonal term."""
    m, k = U.shape
    n = V.shape[0]
    Uh, Vh = U.conj().T, V.conj().T
    s0 = S[0]
    sinv = np.where(np.abs(S) < s0 * eps, 0.0, 1.0 / np.where(S == 0, 1.0, S))            # safe_inverse_2
    F = _safe_inverse(S[None, :] - S[:, None], s0 * eps); np.fill_diagonal(F, 0.0)
    G = _safe_inverse(S[None, :] + S[:, None], s0 * eps); np.fill_diagonal(G, 0.0)
    dA = np.zeros((m, n), dtype=U.dtype)
    if gS is not None:
        dA = dA + (U * gS[None, :]) @ Vh
    if gU is not None:
        t = U @ ((F + G) * (Uh @ gU - gU.conj().T @ U)) * 0.5
        if m > k:
            t = t + (np.eye(m) - U @ Uh) @ (gU * sinv[None, :])
        dA = dA + t @ Vh
    if gV is not None:
        t = ((F - G) * (Vh @ gV - gV.conj().T @ V)) @ Vh * 0.5
        if n > k:
            t = t + sinv[:, None] * (gV.conj().T @ (np.eye(n) - V @ Vh))
        dA = dA + U @ t
    if np.iscomplexobj(U) and gU is not None:
        L = 1j * np.imag(np.diag(Uh @ gU)) * sinv
        dA = dA + (U * L[None, :]) @ Vh
    return dA


def eigh_backward(D, U, gD, gU, reg):
    """SYMEIG.backward (linalg/eig_sym.py:57-75)."""
    F = _safe_inverse(D[None, :] - D[:, None], reg); np.fill_diagonal(F, 0.0)
    mid = np.zeros((U.shape[1], U.shape[1]), dtype=U.dtype)
    if gD is not None:
        mid = mid + np.diag(gD)
    if gU is not None:
        mid = mid + F * (U.conj().T @ gU)
    return U @ mid @ U.conj().T


# ----------------------------------------------------------------------------------
# projectors (ctm/generic/ctm_projectors.py:142-293)
# ----------------------------------------------------------------------------------

def projectors_from_matrices(R, Rt, chi, svd_reltol=1.0e-8, eps_multiplet=1.0e-8, multiplet_abstol=1.0e-14,
                             return_S=False):
    assert R.shape == Rt.shape and R.ndim == 2
    M = R.T @ Rt                                                       # :263 (no conj)
    U, S, V = truncated_svd_gesdd(M, chi, abs_tol=multiplet_abstol, keep_multiplets=True,
                                  eps_multiplet=eps_multiplet)
    nz = S / S[0] > svd_reltol                                         # :266
    S_sqrt = np.zeros_like(S)
    S_nz = S[nz]
    S_sqrt[:S_nz.shape[0]] = 1.0 / np.sqrt(S_nz)                       # :267-270
    P = (R @ U.conj()) * S_sqrt[None, :]                               # :283
    Pt = (Rt @ V) * S_sqrt[None, :]
    if return_S:
        return P, Pt, S
    return P, Pt


def get_projectors_4x4(direction, coord, state, env, **kw):
    R, Rt = halves(direction, coord, state, env)
    return projectors_from_matrices(R, Rt, env.chi, **kw)


def halves_4x2(direction, coord, state, env):
    """ctm_get_projectors_4x2 (ctm_projectors.py:66-136): R, Rt are the single enlarged corners next to the cut --
    the first corner of each half of the 4x4 network with the same transposes (:113-131)."""
    out = []
    for key in ('R', 'Rt'):
        cA, shA, _cB, _shB, opA, _opB = _HALVES[direction][key]
        m = c2x2(cA, (coord[0] + shA[0], coord[1] + shA[1]), state, env)
        out.append(m if opA == 'N' else m.T)
    return out[0], out[1]


def get_projectors_4x2(direction, coord, state, env, **kw):
    R, Rt = halves_4x2(direction, coord, state, env)
    return projectors_from_matrices(R, Rt, env.chi, **kw)


# ----------------------------------------------------------------------------------
# absorb + truncate (ctm/generic/ctmrg.py:324-804), 'sl' mode
# ----------------------------------------------------------------------------------
# per direction: relative vectors of (C1,T1,T,T2,C2), neighbour shift for (P1,Pt1), einsum
# specs of nC1, nC2, nT, site legs used to split T / Pt2 / P1, output fusion of nT.
_ABSORB = {
    # einsum operand order: nC1 <- (Pt1, C1, T1); nC2 <- (C2, T2, P2); nT <- (T, Pt2, A, conj(A), P1)
    UP: dict(C1=(1, -1), T1=(1, 0), T=(0, -1), T2=(-1, 0), C2=(-1, -1), shift=(1, 0),
             nC1='abk,ac,cbd->kd', nC2='ca,cdb,abk->dk',                            # :351-374
             nT='abcd,aije,mbifk,mcjgl,dklh->efgh', tsplit=(1, 1), pt2=2, p1=4, fuse=(1, 2)),    # :423
    LEFT: dict(C1=(-1, -1), T1=(0, -1), T=(-1, 0), T2=(0, 1), C2=(-1, 1), shift=(0, -1),
               nC1='abk,ac,cbd->kd', nC2='ac,bcd,abk->kd',                          # :465-484
               nT='abcd,bghm,iecgk,ifdhl,aefj->jmkl', tsplit=(2, 2), pt2=3, p1=1, fuse=(2, 3)),   # :541
    DOWN: dict(C1=(-1, 1), T1=(-1, 0), T=(0, 1), T2=(1, 0), C2=(1, 1), shift=(-1, 0),
               nC1='abk,ca,dcb->dk', nC2='ca,dbc,abk->dk',                          # :593-616
               nT='abcd,dklh,mfiak,mgjbl,cije->fgeh', tsplit=(0, 3), pt2=4, p1=2, fuse=(0, 1)),   # :665
    # RIGHT: T(1,0)'s D^2 leg attaches to the site's right leg (A index 4 <-> T index 1/2 in the
    # einsum at ctmrg.py:781; the reference's view() uses A.size(2), equal for uniform D).
    RIGHT: dict(C1=(1, 1), T1=(0, 1), T=(1, 0), T2=(0, -1), C2=(1, -1), shift=(0, 1),
                nC1='abk,ac,bdc->kd', nC2='ca,dbc,abk->dk',                         # :706-724
                nT='abcd,aefj,iekgb,iflhc,dghm->jklm', tsplit=(1, 4), pt2=1, p1=3, fuse=(1, 2)),  # :781
}


def _as3(P, chi):
    return P.reshape(chi, P.shape[0] // chi, P.shape[1])


def absorb_truncate(direction, coord, state, env, P, Pt):
    """absorb_truncate_CTM_MOVE_<DIR>(+_c): returns un-normalised nC1, nC2, nT."""
    sp = _ABSORB[direction]
    c = state.vertexToSite(coord)
    nb = state.vertexToSite((coord[0] + sp['shift'][0], coord[1] + sp['shift'][1]))
    C1, T1, T, T2, C2 = (env.C[(c, sp['C1'])], env.T[(c, sp['T1'])], env.T[(c, sp['T'])],
                         env.T[(c, sp['T2'])], env.C[(c, sp['C2'])])
    A = state.site(coord)
    chi = C1.shape[0]
    P2, Pt2, P1, Pt1 = _as3(P[c], chi), _as3(Pt[c], chi), _as3(P[nb], chi), _as3(Pt[nb], chi)
    nC1 = seq_einsum(sp['nC1'], Pt1, C1, T1)
    nC2 = seq_einsum(sp['nC2'], C2, T2, P2)
    Tv = _split(T, sp['tsplit'][0], A.shape[sp['tsplit'][1]])
    Pt2v = _split(Pt2, 1, A.shape[sp['pt2']])
    P1v = _split(P1, 1, A.shape[sp['p1']])
    nT = seq_einsum(sp['nT'], Tv, Pt2v, A, A.conj(), P1v)
    f0, f1 = sp['fuse']
    sh = list(nT.shape)
    nT = nT.reshape(sh[:f0] + [sh[f0] * sh[f1]] + sh[f1 + 1:])
    return nC1, nC2, nT


_REL = {UP: ((1, -1), (-1, -1)), LEFT: ((-1, -1), (-1, 1)), DOWN: ((-1, 1), (1, 1)), RIGHT: ((1, 1), (1, -1))}


def ctm_move(direction, state, env, norm_type='inf', projector_method='4X4', **kw):
    """ctm_MOVE (ctmrg.py:179-319): projectors for all sites from the OLD env, absorb for all
    sites, normalise each new tensor by its own max-abs ('inf') or vector 2-norm (any other
    ctm_absorb_normalization, ctmrg.py:210-230), scatter to coord - direction."""
    nrm = _nrm if norm_type == 'inf' else (lambda a: a / np.linalg.norm(a.ravel()))
    P, Pt = {}, {}
    for coord in state.sites:
        P[coord], Pt[coord] = (get_projectors_4x4 if projector_method == '4X4' else get_projectors_4x2)(direction, coord, state, env, **kw)
    new = {}
    for coord in state.sites:
        nC1, nC2, nT = absorb_truncate(direction, coord, state, env, P, Pt)
        new[coord] = (nrm(nC1), nrm(nC2), nrm(nT))                     # :210-230
    r1, r2 = _REL[direction]
    for coord in state.sites:
        nc = state.vertexToSite((coord[0] - direction[0], coord[1] - direction[1]))
        env.C[(nc, r1)], env.C[(nc, r2)], env.T[(nc, direction)] = new[coord]   # :313-319
    return P, Pt


def ctm_sweep(state, env, **kw):
    """_ctmrg_iter (ctmrg.py:63-69)."""
    for d in DIRECTIONS:
        reps = state.lX if d in (LEFT, RIGHT) else state.lY
        for _ in range(reps):
            ctm_move(d, state, env, **kw)


def corner_spectra(env):
    """ENV.get_spectra (env.py:204-209)."""
    spec = {}
    for k, c in env.C.items():
        s = np.linalg.svd(c, compute_uv=False)
        spec[k] = s / s[0]
    return spec


def conv_specC(env, history, tol=1.0e-8, max_iter=50):
    """ctmrg_conv_specC (env.py:816-875), p='inf'."""
    if not history:
        history = {'spec': [], 'diffs': [], 'conv_crit': []}
    spec = {k: np.sort(v)[::-1] for k, v in corner_spectra(env).items()}
    crit, diffs = float('inf'), None
    if history['spec']:
        old = history['spec'][-1]
        diffs = [float(np.sum((spec[k] - old[k]) ** 2)) for k in spec]
        crit = float(np.sqrt(max(diffs)))
    history['spec'].append(spec); history['diffs'].append(diffs); history['conv_crit'].append(crit)
    done = (len(history['diffs']) > 1 and crit < tol) or len(history['diffs']) >= max_iter
    return done, history


# ----------------------------------------------------------------------------------
# RDMs (ctm/generic/rdm.py)
# ----------------------------------------------------------------------------------

def sym_pos_def_rdm(rdm, sym_pos_def=False):
    """_sym_pos_def_rdm/_matrix (rdm.py:38-68)."""
    sh = rdm.shape
    n = int(np.prod(sh[:len(sh) // 2]))
    m = rdm.reshape(n, n)
    m = 0.5 * (m + m.conj().T)
    if sym_pos_def:
        D, U = np.linalg.eigh(m)
        if D.min() < 0:
            D = np.clip(D, 0, None)
            m = (U * D[None, :]) @ U.conj().T
    m = m / np.real(np.trace(m))
    return m.reshape(sh)


def rdm2x2(coord, state, env, sym_pos_def=False):
    """rdm2x2_legacy (rdm.py:1362-1592): 4 open corners -> 2 halves -> trace.
    index order s0 s1 s2 s3 ; s0' s1' s2' s3' with s0=coord, s1=+x, s2=+y, s3=+x+y."""
    x, y = coord
    cLU = c2x2(LU, (x, y), state, env, open_=True)
    cRU = c2x2(RU, (x + 1, y), state, env, open_=True)
    cRD = c2x2(RD, (x + 1, y + 1), state, env, open_=True)
    cLD = c2x2(LD, (x, y + 1), state, env, open_=True)
    up = np.einsum('akst,kbuv->abstuv', cLU, cRU, optimize=True)
    lo = np.einsum('akst,bkuv->abstuv', cLD, cRD, optimize=True)
    r = np.einsum('abstuv,abwxyz->stuvwxyz', up, lo, optimize=True)
    r = r.transpose(0, 2, 4, 6, 1, 3, 5, 7)
    return sym_pos_def_rdm(r, sym_pos_def)


def rdm1x1(coord, state, env, sym_pos_def=False):
    """rdm1x1 (rdm.py:71-302, 'dl' route): full 1-site environment traced over the aux legs."""
    c = state.vertexToSite(coord)
    a = state.site(coord)
    D = a.shape
    C1, C2, C3, C4 = env.C[(c, (-1, -1))], env.C[(c, (1, -1))], env.C[(c, (1, 1))], env.C[(c, (-1, 1))]
    T1 = _split(env.T[(c, (0, -1))], 1, D[1])      # (l, uk, ub, r)
    T4 = _split(env.T[(c, (-1, 0))], 2, D[2])      # (u, d, lk, lb)
    T3 = _split(env.T[(c, (0, 1))], 0, D[3])       # (dk, db, l, r)
    T2 = _split(env.T[(c, (1, 0))], 1, D[4])       # (u, rk, rb, d)
    # C1(down a, right b) T1(b,U,V,c) C2(left c, down e) T2(e,R,Q,f) C3(up f, left g)
    # T3(X,Y,h,g) C4(up i, right h) T4(a,i,L,M)
    r = np.einsum('ab,bUVc,ce,eRQf,fg,XYhg,ih,aiLM,sULXR,tVMYQ->st',
                  C1, T1, C2, T2, C3, T3, C4, T4, a, a.conj(), optimize=True)
    return sym_pos_def_rdm(r, sym_pos_def)


def rdm2x1(coord, state, env, sym_pos_def=False):
    """rdm2x1 (rdm.py:304-500, 'dl' route): horizontal pair coord, coord+(1,0); order s0 s1; s0' s1'."""
    x, y = coord
    c0 = state.vertexToSite((x, y)); c1 = state.vertexToSite((x + 1, y))
    a0, a1 = state.site((x, y)), state.site((x + 1, y))
    D0, D1 = a0.shape, a1.shape
    C1, C4 = env.C[(c0, (-1, -1))], env.C[(c0, (-1, 1))]
    T1a = _split(env.T[(c0, (0, -1))], 1, D0[1]); T4 = _split(env.T[(c0, (-1, 0))], 2, D0[2])
    T3a = _split(env.T[(c0, (0, 1))], 0, D0[3])
    C2, C3 = env.C[(c1, (1, -1))], env.C[(c1, (1, 1))]
    T1b = _split(env.T[(c1, (0, -1))], 1, D1[1]); T2 = _split(env.T[(c1, (1, 0))], 1, D1[4])
    T3b = _split(env.T[(c1, (0, 1))], 0, D1[3])
    left = np.einsum('ab,bUVc,ih,aiLM,XYhg,sULXR,tVMYQ->cRQgst', C1, T1a, C4, T4, T3a, a0, a0.conj(),
                     optimize=True)
    right = np.einsum('ce,eRQf,fg,jUVc,XYhg,sULXR,tVMYQ->jLMhst', C2, T2, C3, T1b, T3b, a1, a1.conj(),
                      optimize=True)
    r = np.einsum('cRQgst,cRQguv->sutv', left, right, optimize=True)
    return sym_pos_def_rdm(r, sym_pos_def)


def rdm1x2(coord, state, env, sym_pos_def=False):
    """rdm1x2 (rdm.py:622-826, 'dl' route): vertical pair coord, coord+(0,1); order s0 s1; s0' s1'."""
    x, y = coord
    c0 = state.vertexToSite((x, y)); c1 = state.vertexToSite((x, y + 1))
    a0, a1 = state.site((x, y)), state.site((x, y + 1))
    D0, D1 = a0.shape, a1.shape
    C1, C2 = env.C[(c0, (-1, -1))], env.C[(c0, (1, -1))]
    T1 = _split(env.T[(c0, (0, -1))], 1, D0[1]); T4a = _split(env.T[(c0, (-1, 0))], 2, D0[2])
    T2a = _split(env.T[(c0, (1, 0))], 1, D0[4])
    C4, C3 = env.C[(c1, (-1, 1))], env.C[(c1, (1, 1))]
    T3 = _split(env.T[(c1, (0, 1))], 0, D1[3]); T4b = _split(env.T[(c1, (-1, 0))], 2, D1[2])
    T2b = _split(env.T[(c1, (1, 0))], 1, D1[4])
    up = np.einsum('ab,bUVc,ce,aiLM,eRQf,sULXR,tVMYQ->iXYfst', C1, T1, C2, T4a, T2a, a0, a0.conj(),
                   optimize=True)
    lo = np.einsum('jh,XYhg,fg,ijLM,eRQf,sULXR,tVMYQ->iUVest', C4, T3, C3, T4b, T2b, a1, a1.conj(),
                   optimize=True)
    r = np.einsum('iXYfst,iXYfuv->sutv', up, lo, optimize=True)
    return sym_pos_def_rdm(r, sym_pos_def)


# ----------------------------------------------------------------------------------
# transfer-matrix correlators (ctm/generic/corrf.py:10-103,234-276,364-670,980-1067)
# ----------------------------------------------------------------------------------
_EDGE_SPEC = {
    UP: (((-1, -1), UP, (1, -1)), "ax,xby,yc->abc"),          # corrf.py:43-56
    LEFT: (((-1, -1), LEFT, (-1, 1)), "xa,xyb,yc->abc"),      # :57-73
    DOWN: (((-1, 1), DOWN, (1, 1)), "ax,bxy,cy->abc"),        # :74-84
    RIGHT: (((1, -1), RIGHT, (1, 1)), "ax,xby,yc->abc"),      # :85-101
}
# (T1, T2) and the network over (T1, edge, A, T2) with the double-layer site A[u,l,d,r] (corrf.py:449-667)
_TM_SPEC = {
    UP: ((LEFT, RIGHT), "axl,xdy,bldr,cry->abc"),
    LEFT: ((UP, DOWN), "aux,xry,ubdr,dcy->abc"),
    DOWN: ((LEFT, RIGHT), "xal,xuy,ulbr,yrc->abc"),
    RIGHT: ((UP, DOWN), "xua,xly,uldb,dyc->abc"),
}


def get_edge(coord, direction, state, env):
    c = state.vertexToSite(coord)
    (c1, t, c2), expr = _EDGE_SPEC[direction]
    return seq_einsum(expr, env.C[(c, c1)], env.T[(c, t)], env.C[(c, c2)])


def apply_edge(coord, direction, state, env, vec):
    return np.tensordot(vec, get_edge(coord, direction, state, env), ([0, 1, 2], [0, 1, 2]))


def apply_TM_1sO(coord, direction, state, env, edge, op=None):
    """corrf.py:364-670 (no MPO index): A = a^+ op a with the ket layer carrying op (get_aXa, :408-421)."""
    c = state.vertexToSite(coord)
    a = state.site(c)
    ket = a if op is None else np.einsum('mefgh,mn->nefgh', a, op)
    d = a.shape
    A = np.einsum('nefgh,nabcd->eafbgchd', ket, a.conj()).reshape(d[1] ** 2, d[2] ** 2, d[3] ** 2, d[4] ** 2)
    (v1, v2), expr = _TM_SPEC[direction]
    return seq_einsum(expr, env.T[(c, v1)], edge, A, env.T[(c, v2)])


def corrf_1sO1sO(coord, direction, state, env, op1, get_op2, dist):
    """corrf.py:980-1067: <O1(0) O2(r)>, r = 1..dist+1, with the reference's running normalisation."""
    c0 = coord
    rev = (-direction[0], -direction[1])
    E0 = get_edge(c0, rev, state, env)
    E1 = apply_TM_1sO(c0, direction, state, env, E0, op=op1)
    E0 = apply_TM_1sO(c0, direction, state, env, E0)
    out = np.empty(dist + 1, dtype=E0.dtype)
    for r in range(dist + 1):
        c0 = (c0[0] + direction[0], c0[1] + direction[1])
        E12 = apply_TM_1sO(c0, direction, state, env, E1, op=get_op2(r))
        E0 = apply_TM_1sO(c0, direction, state, env, E0)
        E1 = apply_TM_1sO(c0, direction, state, env, E1)
        out[r] = apply_edge(c0, direction, state, env, E12) / apply_edge(c0, direction, state, env, E0)
        m = np.abs(E0).max()
        E0 = E0 / m; E1 = E1 / m
    return out


def get_Top_spec(n, coord, direction, state, env):
    """transferops.py:119-207: leading n eigenvalues (modulus-descending, |lambda_0| = 1) of the width-0 transfer operator,
    ARPACK on a matrix-free operator built from lX (lY) transfer steps."""
    from scipy.sparse.linalg import LinearOperator, eigs
    c = state.vertexToSite(coord)
    leg = {(0, -1): 1, (-1, 0): 2, (0, 1): 3, (1, 0): 4}[(-direction[0], -direction[1])]
    ad = state.site(c).shape[leg]
    chi = env.chi
    N = state.lX if direction in (LEFT, RIGHT) else state.lY
    dt = next(iter(env.T.values())).dtype

    def mv(v):
        V = np.asarray(v, dtype=dt).reshape(chi, ad * ad, chi)
        c0 = coord
        for _ in range(N):
            V = apply_TM_1sO(c0, direction, state, env, V)
            c0 = (c0[0] + direction[0], c0[1] + direction[1])
        return V.reshape(-1)

    dim = chi * ad * ad * chi
    vals = eigs(LinearOperator((dim, dim), matvec=mv, dtype=dt), k=n, return_eigenvectors=False)
    vals = vals[np.argsort(np.abs(vals))[::-1]]
    return vals / np.abs(vals[0])
