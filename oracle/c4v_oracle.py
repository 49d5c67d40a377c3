"""numpy restatement of the reference's one-site C4v CTMRG move and RDMs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Conventions (ctm/one_site_c4v/env_c4v.py:38-51):
one corner C (chi x chi, Hermitian), one half-row tensor T (chi, chi, D^2), T_ija = conj(T_jia);
site a[s,u,l,d,r] with all four aux dims equal.
"""
import numpy as np
from .ctm_oracle import truncated_eig_sym, sym_pos_def_rdm, seq_einsum


def c2x2_sl(a, C, T, open_=False):
    """c2x2_sl (ctm_components_c4v.py:52-130) / _get_open_C2x2_LU_sl (rdm_c4v.py:13-93).
    C[x,y] T[.,y,(u U)] T[x,.,(l L)] a[s,u,l,d,r] conj(a)[s,U,L,D,R] -> (e d D),(c r R)."""
    chi, D = C.shape[0], a.shape[1]
    Tv = T.reshape(chi, chi, D, D)
    if open_:
        r = seq_einsum('xy,cyuU,xelL,suldr,tULDR->edDcrRst', C, Tv, Tv, a, a.conj())
        return r.reshape(chi * D * D, chi * D * D, a.shape[0], a.shape[0])
    r = seq_einsum('xy,cyuU,xelL,suldr,sULDR->edDcrR', C, Tv, Tv, a, a.conj())
    return r.reshape(chi * D * D, chi * D * D)


def init_env_ctmrg(a, chi):
    """_init_from_ipeps_pbc (env_c4v.py:262-311)."""
    D = a.shape[1]
    c = np.einsum('mijef,mijab->eafb', a, a.conj()).reshape(D * D, D * D)
    c = c / np.abs(c).max()
    assert np.linalg.norm(c.conj().T - c) / np.abs(c).max() < 1.0e-8
    Dv, U = truncated_eig_sym(c, c.shape[0])
    C = np.zeros((chi, chi), dtype=a.dtype)
    m = min(chi, D * D)
    C[:m, :m] = np.diag(Dv)[:m, :m]
    t = np.einsum('meifg,maibc->eafbgc', a, a.conj()).reshape(D * D, D * D, D * D)
    t = t / np.abs(t).max()
    t = np.einsum('ai,abs,bj->ijs', U, t, U.conj())
    T = np.zeros((chi, chi, D * D), dtype=a.dtype)
    T[:m, :m, :] = t[:m, :m, :]
    return C, T


def init_env_prod(a, chi):
    """init_prod (env_c4v.py:215-246): C = e_00, T[0,0,:] = leading eigenvector (by |eigenvalue|) of 'meifj,maibj->eafb' / max-abs."""
    D = a.shape[1]
    t = np.einsum('meifj,maibj->eafb', a, a.conj()).reshape(D * D, D * D)
    t = t / np.abs(t).max()
    Dv, U = truncated_eig_sym(t, 2)
    C = np.zeros((chi, chi), dtype=a.dtype); C[0, 0] = 1.0
    T = np.zeros((chi, chi, D * D), dtype=a.dtype); T[0, 0, :] = U[:, 0]
    return C, T


def init_env_obc(a, chi):
    """init_from_ipeps_obc (env_c4v.py:315-355): outward legs summed layer by layer (ket and conjugated bra), / max-abs, zero padded."""
    D = a.shape[1]
    c = np.einsum('mijef,mklab->eafb', a, a.conj()).reshape(D * D, D * D)
    c = c / np.abs(c).max()
    C = np.zeros((chi, chi), dtype=a.dtype)
    m = min(chi, D * D)
    C[:m, :m] = c[:m, :m]
    t = np.einsum('meifg,makbc->eafbgc', a, a.conj()).reshape(D * D, D * D, D * D)
    t = t / np.abs(t).max()
    T = np.zeros((chi, chi, D * D), dtype=a.dtype)
    T[:m, :m, :] = t[:m, :m, :]
    return C, T


def ctm_move_sl(a, C, T, chi=None, eps_multiplet=1.0e-12, abs_tol=1.0e-14, return_P=False, norm_type='inf'):
    """ctm_MOVE_sl (ctmrg_c4v.py:325-463) with truncated_eig_sym(keep_multiplets=True)
    (ctmrg_c4v.py:49-52: eps_multiplet/abs_tol at their custom_eig.py defaults)."""
    chi = C.shape[0] if chi is None else chi
    D = a.shape[1]
    C2X2 = c2x2_sl(a, C, T)
    Dv, P = truncated_eig_sym(C2X2, chi, abs_tol=abs_tol, keep_multiplets=True, eps_multiplet=eps_multiplet)
    nC = np.diag(Dv).astype(a.dtype)                                   # :374
    P3 = P.reshape(chi, D * D, chi)
    Pv = P3.reshape(chi, D, D, chi)
    Tv = T.reshape(chi, chi, D, D)
    # nT = P . T . a . a* . P*  (:383-443)
    nT = seq_einsum('xuUi,xelL,suldr,sULDR,edDj->ijrR', Pv, Tv, a, a.conj(), Pv.conj())
    nT = nT.reshape(chi, chi, D * D)
    nT = 0.5 * (nT + nT.conj().transpose(1, 0, 2))                     # :446
    nC = nC / np.abs(nC[0, 0])                                         # :182-197
    nT = nT / (np.abs(nT).max() if norm_type == 'inf' else np.linalg.norm(nT.ravel()))   # vector_norm(ord=inf | 2)
    if return_P:
        return nC, nT, Dv, P
    return nC, nT


def rdm2x1_sl(a, C, T, sym_pos_def=False):
    """rdm2x1_sl (rdm_c4v.py:530-665)."""
    chi, D = C.shape[0], a.shape[1]
    C2x2 = c2x2_sl(a, C, T, open_=True).reshape(chi, D * D, chi, D * D, a.shape[0], a.shape[0])
    C2x1 = np.tensordot(C, T, ([1], [0]))                               # [C0, T1, T2]
    left = np.tensordot(C2x1, C2x2, ([0, 2], [0, 1]))                   # [T1, chi, D^2, s, t]
    r = np.tensordot(left, left, ([0, 1, 2], [1, 0, 2]))                # [s,t,s',t']
    r = r.transpose(0, 2, 1, 3)
    return sym_pos_def_rdm(r, sym_pos_def)


def rdm1x1(a, C, T, sym_pos_def=False):
    """rdm1x1 (rdm_c4v.py:168-262)."""
    chi, D, p = C.shape[0], a.shape[1], a.shape[0]
    CTC = np.tensordot(np.tensordot(C, T, ([0], [0])), C, ([1], [0]))
    r = np.tensordot(CTC, T, ([2], [0]))
    aa = np.einsum('mefgh,nabcd->eafbgchdmn', a, a.conj()).reshape(D * D, D * D, D * D, D * D, p, p)
    r = np.tensordot(r, aa, ([1, 3], [1, 2]))
    r = np.tensordot(T, r, ([1, 2], [0, 2]))
    r = np.tensordot(r, CTC, ([0, 1, 2], [2, 0, 1]))
    return sym_pos_def_rdm(r, sym_pos_def)


def apply_TM_1sO(a, C, T, edge, op=None):
    """corrf_c4v.apply_TM_1sO (corrf_c4v.py:178-271)."""
    D, p = a.shape[1], a.shape[0]
    X = np.eye(p, dtype=a.dtype) if op is None else op
    A = np.einsum('mefgh,mn,nabcd->eafbgchd', a, X, a.conj()).reshape(D * D, D * D, D * D, D * D)
    E = np.tensordot(edge, T, ([0], [1]))
    E = np.tensordot(E, A, ([0, 3], [1, 0]))
    return np.tensordot(E, T, ([0, 2], [0, 2]))


def corrf_1sO1sO(a, C, T, op1, get_op2, dist):
    """corrf_c4v.corrf_1sO1sO (corrf_c4v.py:593-664) with the C - T - C boundaries."""
    def edge_scalar(v):
        S = np.tensordot(v, C, ([0], [0]))
        S = np.tensordot(S, T, ([0, 2], [2, 1]))
        return np.tensordot(S, C, ([0, 1], [1, 0]))
    E0 = np.tensordot(np.tensordot(C, T, ([0], [0])), C, ([1], [0]))
    E1 = apply_TM_1sO(a, C, T, E0, op1)
    E0 = apply_TM_1sO(a, C, T, E0)
    out = np.empty(dist + 1, dtype=a.dtype)
    for r in range(dist + 1):
        E12 = apply_TM_1sO(a, C, T, E1, get_op2(r))
        E0 = apply_TM_1sO(a, C, T, E0)
        E1 = apply_TM_1sO(a, C, T, E1)
        out[r] = edge_scalar(E12) / edge_scalar(E0)
        m = np.abs(E0).max()
        E0, E1 = E0 / m, E1 / m
    return out


def rdm3x1_sl(a, C, T, sym_pos_def=False):
    """rdm3x1_sl (rdm_c4v.py:829-994): end sites of a 3x1 strip, middle site traced; s0 s1 ; s0' s1'."""
    chi, D, p = C.shape[0], a.shape[1], a.shape[0]
    c6 = c2x2_sl(a, C, T, open_=True).reshape(chi, D * D, chi, D * D, p, p)
    C2x1 = np.tensordot(C, T, ([1], [0]))
    left = np.tensordot(C2x1, c6, ([0, 2], [0, 1])).reshape(chi, chi, D, D, p, p)
    Tv = T.reshape(chi, chi, D, D)
    mid = seq_einsum('xpgG,xylLst,SULGR,Sulgr,PyuU->PrRpst', Tv, left, a.conj(), a, Tv)
    r = np.einsum('PrRpst,PprRuv->stuv', mid, left, optimize=True).transpose(0, 2, 1, 3)
    return sym_pos_def_rdm(r, sym_pos_def)


def _nn_pieces(a, C, T):
    n = C.shape[0] * a.shape[1] ** 2
    C2x2 = c2x2_sl(a, C, T, open_=True)
    C2x2c = np.einsum('abii->ab', C2x2)
    return C2x2.reshape(n, n, a.shape[0] ** 2), C2x2c


def rdm2x2_NN_lowmem_sl(a, C, T, sym_pos_def=False):
    """_rdm2x2_NN_lowmem (rdm_c4v.py:1204-1284)."""
    p = a.shape[0]
    C2x2, C2x2c = _nn_pieces(a, C, T)
    r = np.tensordot(C2x2c, C2x2, ([1], [0]))
    r = np.tensordot(C2x2c, r, ([1], [0]))
    r = np.tensordot(C2x2, r, ([0, 1], [1, 0]))
    r = r.reshape(p, p, p, p).transpose(0, 2, 1, 3)
    return sym_pos_def_rdm(r, sym_pos_def)


def rdm2x2_NNN_lowmem_sl(a, C, T, sym_pos_def=False):
    """_rdm2x2_NNN_lowmem (rdm_c4v.py:1373-1443)."""
    p = a.shape[0]
    C2x2, C2x2c = _nn_pieces(a, C, T)
    h = np.tensordot(C2x2c, C2x2, ([1], [0]))
    r = np.tensordot(h, h, ([0, 1], [1, 0]))
    r = r.reshape(p, p, p, p).transpose(0, 2, 1, 3)
    return sym_pos_def_rdm(r, sym_pos_def)


def rdm2x2(a, C, T, sym_pos_def=False):
    """rdm2x2 (rdm_c4v.py:1446-1545): 4 identical open corners around the plaquette;
    index order s0 s1 s2 s3 ; s0' s1' s2' s3' with s0 s1 / s2 s3."""
    c = c2x2_sl(a, C, T, open_=True)           # (down, right, s, t)
    # LU at s0; by C4v symmetry the other three corners are rotations of the same tensor:
    # RU: (left,down) -> c[left=right-leg..]; contraction pattern follows rdm_c4v.py:1489-1533
    up = np.tensordot(c, c, ([1], [0]))         # [d0, s0,t0, r1(=down of RU), s1,t1]
    up = up.transpose(0, 3, 1, 2, 4, 5)         # [d0, d1, s0,t0,s1,t1]
    r = np.tensordot(up, up, ([0, 1], [1, 0]))  # [s0,t0,s1,t1, s3,t3,s2,t2]
    r = r.transpose(0, 2, 6, 4, 1, 3, 7, 5)     # s0 s1 s2 s3 ; t0 t1 t2 t3
    return sym_pos_def_rdm(r, sym_pos_def)
