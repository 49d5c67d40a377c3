"""Enlarged corner of the one-site C4v network (reference ctm/one_site_c4v/ctm_components_c4v.py:9-130)."""
from backend import get_engine
from linalg.native_einsum import einsum, needs_grad


def c2x2_sl(a, C, T, verbosity=0):
    """C[x,y] T[c,y,(u U)] T[x,e,(l L)] a[s,u,l,d,r] conj(a)[s,U,L,D,R] -> (e d D),(c r R)   (:52-130).
    Forward only: the fused native corner.  With a tensor that requires grad: the same network as one differentiable native
    contraction (linalg/native_einsum.py), so that autograd sees what it sees in the reference."""
    if needs_grad(a, C, T):
        chi, D = C.shape[0], a.shape[1]
        Tv = T.reshape(chi, chi, D, D)
        r = einsum('xy,cyuU,xelL,suldr,sULDR->edDcrR', C, Tv, Tv, a, a, conj=(4,))
        return r.reshape(chi * D * D, chi * D * D)
    return get_engine().c2x2_c4v(a, C, T, open_=False)


c2x2_dl = c2x2_sl
