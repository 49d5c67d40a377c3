"""SVD with the reference's regularised backward (linalg/svd_gesdd.py:77-96 forward, :209-328 backward) as a torch.autograd.Function
whose forward AND backward run on the native engine: forward = full decomposition by the native Jacobi SVD + fix_svd_signs,
backward = ctm_svd_backward (GEMMs + one elementwise kernel).  This is the decomposition-level part of the backward pass
(SURVEY 8 f4); the contractions of a move are differentiable nodes of their own (linalg/native_einsum.py, ctm/generic/ctm_ad.py)."""
import torch
from backend import get_engine
from linalg.native_einsum import mark_stream


class SVDGESDD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, cutoff, diagnostics=None, basis=None):
        """basis: optional workspace (engine.warm_basis(n, n, dtype)) holding the left vectors of the previous call on a nearby
        matrix -- the Jacobi sweeps then start from rows that are already almost orthogonal; updated in place."""
        eng = get_engine()
        mark_stream(A)
        n = min(A.shape)
        kw = {"basis": basis} if basis is not None else {}
        U, S, V = eng.truncated_svd(A.detach(), n, eng.cfg(keep_multiplets=False), **kw)
        ctx.save_for_backward(U, S, V)
        ctx.cutoff = float(cutoff)
        return U, S, V

    @staticmethod
    def backward(ctx, gu, gsigma, gv):
        U, S, V = ctx.saved_tensors
        mark_stream(U, S, V, gu, gsigma, gv)
        return get_engine().svd_backward(U, S, V, gu, gsigma, gv, eps=ctx.cutoff), None, None, None
