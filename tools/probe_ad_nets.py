"""Development probe: every contraction network of the differentiable generic path, native forward/backward vs torch.einsum autograd."""
import sys, os, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "peps-torch_amd"))
import _native
from linalg.native_einsum import einsum
from ctm.generic import ctm_ad as A
eng = _native.engine()
g = torch.Generator().manual_seed(1)
ext_default = 3
nets = []
for c, sp in A._CORNER.items():
    nets += [(sp['closed'], (4,)), (sp['open'], (4,))]
for d, sp in A._ABSORB.items():
    nets += [(sp['nC1'], ()), (sp['nC2'], ()), (sp['nT'], (3,))]
nets += [('ab,bc->ac', ()), ('ba,bc->ac', ()), ('ab,cb->ac', ()), ('ba,cb->ac', ()), ('ab,bk->ak', (1,)),
         ('akst,kbuv->abstuv', ()), ('akst,bkuv->abstuv', ()), ('abstuv,abwxyz->stuvwxyz', ())]
for cplx in (False, True):
    dt = torch.complex128 if cplx else torch.float64
    for expr, conj in nets:
        lhs, out = expr.split('->')
        ins = lhs.split(',')
        ext = {}
        for idx in ins:
            for ch in idx:
                ext.setdefault(ch, 2 + (ord(ch) % 3))
        ops = [torch.randn(*[ext[ch] for ch in idx], generator=g, dtype=dt).cuda().requires_grad_(True) for idx in ins]
        ref = [o.detach().clone().requires_grad_(True) for o in ops]
        mine = einsum(expr, *ops, conj=conj)
        tor = torch.einsum(expr, *[(o.conj() if i in conj else o) for i, o in enumerate(ref)])
        w = torch.randn(*mine.shape, generator=g, dtype=dt).cuda()
        (mine * w).sum().abs().backward(); (tor * w).sum().abs().backward()
        ef = float((mine - tor).abs().max() / tor.abs().max())
        eb = max(float((a.grad - b.grad).abs().max() / b.grad.abs().max()) for a, b in zip(ops, ref))
        print(f"{'c128' if cplx else 'f64 '} {expr:45s} fwd {ef:.1e} bwd {eb:.1e}" + ("   <-----" if eb > 1e-10 or ef > 1e-10 else ""), flush=True)
