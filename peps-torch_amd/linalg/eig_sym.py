"""Symmetric / Hermitian eigendecomposition with the reference's regularised backward (linalg/eig_sym.py:13-34 forward, :57-75
backward) as a torch.autograd.Function on the native engine (forward: native Jacobi eigensolver, ordered by |D| descending;
backward: ctm_eigh_backward)."""
import torch
from backend import get_engine
from linalg.native_einsum import mark_stream


class SYMEIG(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, ad_decomp_reg, basis=None):
        """basis: optional n x n workspace (engine.warm_basis_c4v(n, n, dtype)) holding the eigenvectors of the previous call on a
        nearby matrix -- the Jacobi sweeps then start from rows that are already almost orthogonal; updated in place."""
        eng = get_engine()
        mark_stream(A)
        kw = {"basis": basis} if basis is not None else {}
        D, U = eng.truncated_eigh(A.detach(), A.shape[0], eng.cfg(keep_multiplets=False), **kw)
        ctx.save_for_backward(D, U)
        ctx.reg = float(ad_decomp_reg)
        return D, U

    @staticmethod
    def backward(ctx, dD, dU):
        D, U = ctx.saved_tensors
        mark_stream(D, U, dD, dU)
        return get_engine().eigh_backward(D, U, dD, dU, reg=ctx.reg), None, None
