"""Differentiable tensor-network contraction on the native engine.

The reference gets the adjoint of a CTM move from torch's autograd over its `tensordot` / `einsum` / `contiguous().view()` calls
(ctm/one_site_c4v/ctm_components_c4v.py:52-130, ctmrg_c4v.py:383-446, ctm/generic/ctm_components.py:10-265).  Here a whole
network is ONE autograd node: forward = `ctm_einsum` (pairwise contractions on the FP64 MFMA GEMM kernels, permutations on the
tiled transpose kernels), backward = one more `ctm_einsum` per operand that needs a gradient,

    d out / d X_i  :   G_i[idx_i] = sum  gout[idx_out] * prod_{j != i} conj(f_j(X_j))[idx_j],     grad_i = G_i  (conj(G_i) if X_i enters conjugated)

with the operands ordered greedily by the size of the running intermediate.  Operand slots are independent inputs, so a tensor
used twice (a and conj(a)) gets the sum of its slots' gradients from autograd itself.
"""
import torch
from backend import get_engine


def _order(target, first, others, ext):
    """Greedy left-to-right order of `others` (list of (slot, idx)) after `first`: next comes the operand that leaves the smallest
    intermediate.  Returns the list of slots."""
    cur = set(first)
    rest = list(others)
    order = []
    while rest:
        best, best_size = None, None
        for cand in rest:
            later = set(target)
            for o in rest:
                if o is not cand:
                    later |= set(o[1])
            keep = (cur | set(cand[1])) & later
            size = 1
            for ch in keep:
                size *= ext[ch]
            shares = len(cur & set(cand[1])) > 0
            key = (0 if shares else 1, size)
            if best is None or key < best_size:
                best, best_size = cand, key
        if best_size[0] == 1:
            # nothing left shares an index with the running intermediate: the engine contracts pairwise without outer products
            raise NotImplementedError("einsum backward: the adjoint network is disconnected (needs an outer product)")
        later = set(target)
        for o in rest:
            if o is not best:
                later |= set(o[1])
        cur = (cur | set(best[1])) & later
        order.append(best[0])
        rest.remove(best)
    return order


def mark_stream(*tensors):
    """Tell torch's caching allocator that these tensors are read on the CURRENT stream.  The site units of a differentiable move
    run on worker streams (units.UnitPool) and autograd replays every node on the stream of its forward: a tensor allocated on one
    stream is then an operand on another, and without the mark its block could be handed out again (on its own stream) as soon as
    the last reference drops -- while the other stream's kernel that reads it is still queued."""
    if not torch.cuda.is_available():
        return
    cur = None
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            if cur is None:
                cur = torch.cuda.current_stream(t.device)
            t.record_stream(cur)


class EINSUM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, expr, conj, *tensors):
        eng = get_engine()
        mark_stream(*tensors)
        ctx.expr, ctx.conj = expr, tuple(conj)
        ctx.save_for_backward(*tensors)
        return eng.einsum(expr, *[t.detach() for t in tensors], conj=tuple(conj))

    @staticmethod
    def backward(ctx, gout):
        eng = get_engine()
        tensors = ctx.saved_tensors
        mark_stream(gout, *tensors)
        lhs, out = ctx.expr.split("->")
        ins = lhs.split(",")
        ext = {}
        for idx, t in zip(ins, tensors):
            for ch, n in zip(idx, t.shape):
                ext[ch] = n
        gout = gout.contiguous()
        grads = [None, None]
        for i, idx_i in enumerate(ins):
            if not ctx.needs_input_grad[2 + i]:
                grads.append(None)
                continue
            others = [(j, ins[j]) for j in range(len(ins)) if j != i]
            avail = set(out).union(*[set(x) for _, x in others]) if others else set(out)
            if len(set(idx_i)) != len(idx_i) or not set(idx_i) <= avail:
                raise NotImplementedError(f"einsum backward: operand '{idx_i}' of '{ctx.expr}' has an index that is traced out alone")
            order = _order(idx_i, out, others, ext)
            ops = [gout] + [tensors[j].detach() for j in order]
            e = ",".join([out] + [ins[j] for j in order]) + "->" + idx_i
            cj = tuple(1 + k for k, j in enumerate(order) if j not in ctx.conj)      # conj(f_j(X_j)): flip the flag of every other operand
            g = eng.einsum(e, *ops, conj=cj)
            if i in ctx.conj:
                g = g.conj().resolve_conj() if g.is_complex() else g
            grads.append(g)
        return tuple(grads)


def einsum(expr, *tensors, conj=()):
    """Contraction "i0,i1,...->o" on the native engine; differentiable.  conj: operand positions read conjugated."""
    tensors = tuple(t.resolve_conj() if t.is_complex() else t for t in tensors)     # the engine reads the stored values
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return EINSUM.apply(expr, tuple(conj), *tensors)
    return get_engine().einsum(expr, *tensors, conj=tuple(conj))


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)
