"""SVD with the reference's regularised backward (linalg/svd_gesdd.py:77-96 forward, :209-328 backward) as a torch.autograd.Function
whose forward AND backward run on the native engine: forward = full decomposition by the native Jacobi SVD + fix_svd_signs,
backward = ctm_svd_backward (GEMMs + one elementwise kernel).  This is the decomposition-level part of the backward pass
(SURVEY 8 f4); the contractions of a move are differentiable nodes of their own (linalg/native_einsum.py, ctm/generic/ctm_ad.py)."""
import torch
from backend import get_engine


class SVDGESDD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, cutoff, diagnostics=None):
        eng = get_engine()
        n = min(A.shape)
        U, S, V = eng.truncated_svd(A.detach(), n, eng.cfg(keep_multiplets=False))
        ctx.save_for_backward(U, S, V)
        ctx.cutoff = float(cutoff)
        return U, S, V

    @staticmethod
    def backward(ctx, gu, gsigma, gv):
        U, S, V = ctx.saved_tensors
        return get_engine().svd_backward(U, S, V, gu, gsigma, gv, eps=ctx.cutoff), None, None
