"""RDMs of the one-site C4v network (reference ctm/one_site_c4v/rdm_c4v.py:13-136,530-665,1160-1284,
1373-1443,1446-1545)."""
import torch
from backend import get_engine
from linalg.native_einsum import einsum, needs_grad
from ctm.generic.rdm import _sym_pos_def_rdm


def _parts(state, env):
    return next(iter(state.sites.values())), env.C[env.keyC], env.T[env.keyT]


def _get_open_C2x2_LU_sl(C, T, a, verbosity=0):
    return get_engine().c2x2_c4v(a, C, T, open_=True)


_get_open_C2x2_LU_dl = _get_open_C2x2_LU_sl


def _open_c2x2_ad(a, C, T):
    """(e d D),(c r R), s, t: the enlarged corner with the physical legs of the two layers open (:13-93), differentiable."""
    chi, D = C.shape[0], a.shape[1]
    Tv = T.reshape(chi, chi, D, D)
    r = einsum('xy,cyuU,xelL,suldr,tULDR->edDcrRst', C, Tv, Tv, a, a, conj=(4,))
    return r.reshape(chi * D * D, chi * D * D, a.shape[0], a.shape[0])


def _rdm_ad(which, a, C, T):
    """The four RDM kinds as graphs of differentiable native contractions (same index conventions as the fused `ctm_rdm_c4v`):
    0 rdm2x1_sl (:530-665), 1 rdm2x2_NN_lowmem (:1204-1284), 2 rdm2x2_NNN_lowmem (:1373-1443), 3 rdm2x2 (:1446-1545)."""
    chi, D, p = C.shape[0], a.shape[1], a.shape[0]
    n = chi * D * D
    c = _open_c2x2_ad(a, C, T)                                   # [down, right, s, t]
    if which == 0:
        c6 = c.reshape(chi, D * D, chi, D * D, p, p)
        C2x1 = einsum('xy,yba->xba', C, T)                       # [C0, T1, T2]
        left = einsum('xba,xaydst->bydst', C2x1, c6)             # [T1, chi, D^2, s, t]
        r = einsum('bydst,ybduv->stuv', left, left)
        return r.permute(0, 2, 1, 3)
    if which in (1, 2):
        cc = torch.diagonal(c, dim1=2, dim2=3).sum(-1)           # physical legs traced: [down, right]
        c3 = c.reshape(n, n, p * p)
        if which == 1:
            r = einsum('ab,bcs->acs', cc, c3)
            r = einsum('ab,bcs->acs', cc, r)
            r = einsum('abs,bat->st', c3, r)
        else:
            h = einsum('ab,bcs->acs', cc, c3)
            r = einsum('abs,bat->st', h, h)
        return r.reshape(p, p, p, p).permute(0, 2, 1, 3)
    up = einsum('abst,bcuv->acstuv', c, c)                       # [d0, d1, s0,t0,s1,t1]
    r = einsum('abstuv,bawxyz->stuvwxyz', up, up)                # [s0,t0,s1,t1, s3,t3,s2,t2]
    return r.permute(0, 2, 6, 4, 1, 3, 7, 5)


def rdm1x1_sl(state, env, sym_pos_def=False, verbosity=0):
    """One-site reduced density matrix s ; s' (reference rdm_c4v.py:168-262): C - T - C edge, T, the site with its physical legs
    open, T, and the edge again; one native contraction."""
    a, C, T = _parts(state, env)
    chi, D = C.shape[0], a.shape[1]
    E = einsum('xa,xbs,bc->asc', C, T, C).reshape(chi, D, D, chi)
    Tv = T.reshape(chi, chi, D, D)
    r = einsum('alLc,cydD,muldr,nULDR,pauU,yrRp->mn', E, Tv, a, a, Tv, E, conj=(3,))
    return _sym_pos_def_rdm(r, sym_pos_def=sym_pos_def, verbosity=verbosity, who="rdm1x1")


rdm1x1 = rdm1x1_sl


def rdm3x1_sl(state, env, sym_pos_def=False, force_cpu=False, verbosity=0):
    """Reduced density matrix of the two END sites of a horizontal 3x1 strip, s0 s1 ; s0' s1' (the middle site is traced): the
    next-to-next-nearest-neighbour pair of the j3 term (reference rdm_c4v.py:829-994).  Left half = lower-left corner + enlarged
    upper-left corner with the physical legs of the first site open (as in rdm2x1_sl), one closed column T - a a* - T, and the left
    half again for the right end (C4v).  Three native contractions; differentiable like the other RDMs."""
    a, C, T = _parts(state, env)
    chi, D, p = C.shape[0], a.shape[1], a.shape[0]
    c6 = _open_c2x2_ad(a, C, T).reshape(chi, D * D, chi, D * D, p, p)
    C2x1 = einsum('xy,yba->xba', C, T)                                  # [C0, T1, T2]
    # [b, y, l, L, z]: the two physical legs of the end site fused (z = (s t)) so that no intermediate exceeds the engine's rank limit
    left = einsum('xba,xaydst->bydst', C2x1, c6).reshape(chi, chi, D, D, p * p)
    Tv = T.reshape(chi, chi, D, D)
    mid = einsum('xpgG,xylLz,SULGR,Sulgr,PyuU->PrRpz', Tv, left, a, a, Tv, conj=(2,))
    r = einsum('PrRpz,PprRw->zw', mid, left).reshape(p, p, p, p)
    return _sym_pos_def_rdm(r.permute(0, 2, 1, 3).contiguous(), sym_pos_def=sym_pos_def, verbosity=verbosity, who="rdm3x1_sl")


rdm3x1 = rdm3x1_sl


def _rdm(which, who, state, env, sym_pos_def, verbosity):
    a, C, T = _parts(state, env)
    raw = _rdm_ad(which, a, C, T) if needs_grad(a, C, T) else get_engine().rdm_c4v(which, a, C, T)
    return _sym_pos_def_rdm(raw, sym_pos_def=sym_pos_def, verbosity=verbosity, who=who)


def rdm2x1_sl(state, env, sym_pos_def=False, force_cpu=False, verbosity=0):
    return _rdm(0, "rdm2x1_sl", state, env, sym_pos_def, verbosity)


def rdm2x2_NN_lowmem_sl(state, env, sym_pos_def=False, force_cpu=False, verbosity=0):
    return _rdm(1, "rdm2x2_NN_lowmem", state, env, sym_pos_def, verbosity)


def rdm2x2_NNN_lowmem_sl(state, env, sym_pos_def=False, force_cpu=False, verbosity=0):
    return _rdm(2, "rdm2x2_NNN_lowmem", state, env, sym_pos_def, verbosity)


def rdm2x2(state, env, sym_pos_def=False, force_cpu=False, verbosity=0):
    return _rdm(3, "rdm2x2", state, env, sym_pos_def, verbosity)


rdm2x1 = rdm2x1_sl
rdm2x2_NN_lowmem = rdm2x2_NN_lowmem_sl
rdm2x2_NNN_lowmem = rdm2x2_NNN_lowmem_sl
