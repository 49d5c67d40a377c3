"""Backend shim with the reference's names (tn_interface.py:3-27).  `contract` and `mm` run on the
native FP64-MFMA GEMM; layout helpers are torch views (no arithmetic)."""
import torch
from backend import get_engine


def contract(t1, t2, *args):
    """tensordot(t1, t2, (axes1, axes2)) on the native engine: permute contracted legs together,
    one GEMM, result legs = free(t1) + free(t2)."""
    ax1, ax2 = args[0]
    ax1 = [a % t1.dim() for a in ax1]; ax2 = [a % t2.dim() for a in ax2]
    f1 = [i for i in range(t1.dim()) if i not in ax1]
    f2 = [i for i in range(t2.dim()) if i not in ax2]
    eng = get_engine()
    A = eng.permute(t1.contiguous(), f1 + ax1) if f1 + ax1 != list(range(t1.dim())) else t1.contiguous()
    B = eng.permute(t2.contiguous(), ax2 + f2) if ax2 + f2 != list(range(t2.dim())) else t2.contiguous()
    K = 1
    for a in ax1:
        K *= t1.shape[a]
    out = eng.gemm(A.reshape(-1, K), B.reshape(K, -1))
    return out.reshape([t1.shape[i] for i in f1] + [t2.shape[i] for i in f2])


def mm(m1, m2):
    return get_engine().gemm(m1, m2)


def einsum(op, *ts):
    """Explicit-output einsum ("ab,bc->ac") on the native contraction executor (left-to-right pairwise GEMMs)."""
    if "->" not in op:
        raise ValueError("einsum: explicit output indices required")
    return get_engine().einsum(op, *[t.contiguous() for t in ts])


def view(t, *args):
    return t.view(*args)


def permute(t, *args):
    return t.permute(*args)


def contiguous(t):
    return t.contiguous()


def transpose(t):
    return t.transpose(0, 1)


def conj(t):
    return t.conj()
