"""Differentiable generic directional move and 2x2 RDM on the native engine.

The forward-only product path hands a whole unit (four corners -> implicit operator -> leading-chi triplets -> projectors, or the
absorb of a site) to one fused native call; none of that is a differentiable graph.  When a site or environment tensor requires
grad, the same mathematics is assembled here the way the reference's autograd sees it (ctm/generic/ctm_components.py:10-265,
372-884; ctm_projectors.py:142-293; ctmrg.py:179-319, 324-804; rdm.py:1362-1592) from three kinds of nodes, all executed by the
engine: contraction networks (linalg/native_einsum.py), the full SVD with the regularised backward (linalg/svd_gesdd.py) and
elementwise torch ops on chi-sized vectors.  This is the EXPLICIT path: R, R~ and M = R^T R~ are formed (three n^3 products per
projector pair) and decomposed in full, exactly what the reference differentiates.

The tables are the reference's geometry: which environment tensors surround an enlarged corner, which corners make the two
halves of a cut, which tensors an absorb reads (Appendix A of SURVEY.md)."""
import torch
import config as cfg
from backend import get_engine
from linalg.native_einsum import einsum, needs_grad

LU, RU, RD, LD = 0, 1, 2, 3
UP, LEFT, DOWN, RIGHT = (0, -1), (-1, 0), (0, 1), (1, 0)

# per corner: relative vectors of (C, T1, T2); which axis of T1 / T2 is the fused D^2 leg and the site leg (index of a[s,u,l,d,r]) it
# attaches to; the network C.T1.T2.a.conj(a) -> (chi D^2) x (chi D^2) [x p x p with the physical legs open]
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

# per direction: R = opA(cA) opB(cB), R~ likewise: (corner, shift, corner, shift, opA, opB); 'T' = plain transpose
_HALVES = {
    UP:    dict(R=(RU, (0, 0), RD, (0, 1), 'N', 'N'), Rt=(LU, (-1, 0), LD, (-1, 1), 'T', 'N')),
    LEFT:  dict(R=(LU, (0, 0), RU, (1, 0), 'N', 'N'), Rt=(LD, (0, 1), RD, (1, 1), 'N', 'T')),
    DOWN:  dict(R=(LD, (0, 0), LU, (0, -1), 'T', 'N'), Rt=(RD, (1, 0), RU, (1, -1), 'T', 'T')),
    RIGHT: dict(R=(RD, (0, 0), LD, (-1, 0), 'N', 'T'), Rt=(RU, (0, -1), LU, (-1, -1), 'T', 'T')),
}

# per direction: relative vectors of (C1, T1, T, T2, C2), neighbour whose (P1, P~1) enter, the three networks
#   nC1 <- (P~1, C1, T1);  nC2 <- (C2, T2, P2);  nT <- (T, P~2, a, conj(a), P1)
_ABSORB = {
    UP: dict(C1=(1, -1), T1=(1, 0), T=(0, -1), T2=(-1, 0), C2=(-1, -1), shift=(1, 0),
             nC1='abk,ac,cbd->kd', nC2='ca,cdb,abk->dk',
             nT='abcd,aije,mbifk,mcjgl,dklh->efgh', tsplit=(1, 1), pt2=2, p1=4, fuse=(1, 2)),
    LEFT: dict(C1=(-1, -1), T1=(0, -1), T=(-1, 0), T2=(0, 1), C2=(-1, 1), shift=(0, -1),
               nC1='abk,ac,cbd->kd', nC2='ac,bcd,abk->kd',
               nT='abcd,bghm,iecgk,ifdhl,aefj->jmkl', tsplit=(2, 2), pt2=3, p1=1, fuse=(2, 3)),
    DOWN: dict(C1=(-1, 1), T1=(-1, 0), T=(0, 1), T2=(1, 0), C2=(1, 1), shift=(-1, 0),
               nC1='abk,ca,dcb->dk', nC2='ca,dbc,abk->dk',
               nT='abcd,dklh,mfiak,mgjbl,cije->fgeh', tsplit=(0, 3), pt2=4, p1=2, fuse=(0, 1)),
    RIGHT: dict(C1=(1, 1), T1=(0, 1), T=(1, 0), T2=(0, -1), C2=(1, -1), shift=(0, 1),
                nC1='abk,ac,bdc->kd', nC2='ca,dbc,abk->dk',
                nT='abcd,aefj,iekgb,iflhc,dghm->jklm', tsplit=(1, 4), pt2=1, p1=3, fuse=(1, 2)),
}
_REL = {UP: ((1, -1), (-1, -1)), LEFT: ((-1, -1), (-1, 1)), DOWN: ((-1, 1), (1, 1)), RIGHT: ((1, 1), (1, -1))}


def _split(T, axis, D):
    sh = list(T.shape)
    return T.reshape(sh[:axis] + [D, D] + sh[axis + 1:])


def c2x2(corner, coord, state, env, open_=False):
    """c2x2_{LU,RU,RD,LD} (ctm_components.py:372-884), one contraction node."""
    sp = _CORNER[corner]
    s = state.vertexToSite(coord)
    C, T1, T2, a = env.C[(s, sp['C'])], env.T[(s, sp['T1'])], env.T[(s, sp['T2'])], state.site(coord)
    T1v = _split(T1, sp['s1'][0], a.shape[sp['s1'][1]])
    T2v = _split(T2, sp['s2'][0], a.shape[sp['s2'][1]])
    r = einsum(sp['open' if open_ else 'closed'], C, T1v, T2v, a, a, conj=(4,))
    sh = r.shape
    return r.reshape((sh[0] * sh[1] * sh[2], sh[3] * sh[4] * sh[5]) + tuple(sh[6:]))


def _mm(A, oA, B, oB):
    return einsum(('ab' if oA == 'N' else 'ba') + ',' + ('bc' if oB == 'N' else 'cb') + '->ac', A, B)


def halves(direction, coord, state, env, four_by_two=False):
    """halves_of_4x4_CTM_MOVE_* (ctm_components.py:10-265); 4X2: the corner next to the cut alone (ctm_projectors.py:66-136)."""
    out = []
    for key in ('R', 'Rt'):
        cA, sA, cB, sB, oA, oB = _HALVES[direction][key]
        A = c2x2(cA, (coord[0] + sA[0], coord[1] + sA[1]), state, env)
        if four_by_two:
            out.append(A if oA == 'N' else A.t())
            continue
        B = c2x2(cB, (coord[0] + sB[0], coord[1] + sB[1]), state, env)
        out.append(_mm(A, oA, B, oB))
    return out[0], out[1]


def projectors_from_matrices(R, Rt, chi, ctm_args=cfg.ctm_args, basis=None):
    """ctm_get_projectors_from_matrices (ctm_projectors.py:142-293) with the GESDD route: M = R^T R~ = U S V^H (full, regularised
    backward), S^-1/2 on the values above projector_svd_reltol, P = R conj(U) S^-1/2, P~ = R~ V S^-1/2."""
    from linalg.custom_svd import truncated_svd_gesdd
    M = einsum('ba,bc->ac', R, Rt)
    U, S, V = truncated_svd_gesdd(M, chi, keep_multiplets=True, abs_tol=ctm_args.projector_multiplet_abstol,
                                  eps_multiplet=ctm_args.projector_eps_multiplet, ad_decomp_reg=ctm_args.ad_decomp_reg, basis=basis)
    nz = int((S.detach() / S.detach()[0] > ctm_args.projector_svd_reltol).sum())
    S_sqrt = torch.cat([torch.rsqrt(S[:nz]), torch.zeros(S.shape[0] - nz, dtype=S.dtype, device=S.device)])
    P = einsum('ab,bk->ak', R, U, conj=(1,)) * S_sqrt.to(U.dtype)[None, :]
    Pt = einsum('ab,bk->ak', Rt, V) * S_sqrt.to(V.dtype)[None, :]
    return P, Pt


def absorb(direction, coord, state, env, P, Pt):
    """absorb_truncate_CTM_MOVE_<DIR> (ctmrg.py:324-804): un-normalised nC1, nC2, nT."""
    sp = _ABSORB[direction]
    c = state.vertexToSite(coord)
    nb = state.vertexToSite((coord[0] + sp['shift'][0], coord[1] + sp['shift'][1]))
    C1, T1, T, T2, C2 = (env.C[(c, sp['C1'])], env.T[(c, sp['T1'])], env.T[(c, sp['T'])], env.T[(c, sp['T2'])], env.C[(c, sp['C2'])])
    A = state.site(coord)
    chi = C1.shape[0]
    as3 = lambda X: X.reshape(chi, X.shape[0] // chi, X.shape[1])
    P2, Pt2, P1, Pt1 = as3(P[c]), as3(Pt[c]), as3(P[nb]), as3(Pt[nb])
    nC1 = einsum(sp['nC1'], Pt1, C1, T1)
    nC2 = einsum(sp['nC2'], C2, T2, P2)
    Tv = _split(T, sp['tsplit'][0], A.shape[sp['tsplit'][1]])
    Pt2v = _split(Pt2, 1, A.shape[sp['pt2']])
    P1v = _split(P1, 1, A.shape[sp['p1']])
    nT = einsum(sp['nT'], Tv, Pt2v, A, A, P1v, conj=(3,))
    f0, f1 = sp['fuse']
    sh = list(nT.shape)
    return nC1, nC2, nT.reshape(sh[:f0] + [sh[f0] * sh[f1]] + sh[f1 + 1:])


def _warm_ws(env, coord, R, ctm_args):
    """An optimisation evaluates the same sequence of moves again and again on slowly changing tensors: the decomposition of
    (move of the run, site) sees almost the matrix it saw in the previous evaluation.  The environment keeps the left singular
    vectors of each (set by ctmrg.run as env._move_index; handed on by ENV.detach()), and the next full decomposition starts from
    them.  None outside a run, off the GPU, or when the workspaces would take more than 5 % of the device memory."""
    idx = env.__dict__.get("_move_index")
    eng = get_engine()
    if idx is None or not getattr(ctm_args, "projector_warm_start", True) or not hasattr(eng, "warm_basis") or not R.is_cuda:
        return None
    n = R.shape[1]
    ws = env.__dict__.setdefault("_warm_ad", {})
    key = (idx, coord)
    b = ws.get(key)
    if b is None or tuple(b.shape) != ((2 if R.is_complex() else 1) * n + 1, n) or b.device != R.device:
        if (len(ws) + 1) * n * n * 8 * (2 if R.is_complex() else 1) > 0.05 * torch.cuda.get_device_properties(R.device).total_memory:
            return None
        b = ws[key] = eng.warm_basis(n, n, R.dtype)
    return b


def ctm_MOVE(direction, state, env, ctm_args=cfg.ctm_args):
    """ctm_MOVE (ctmrg.py:179-319): projectors of all sites from the old environment, absorb of all sites, each new tensor divided
    by its own norm taken without gradient (:210-230), scatter to coord - direction.  fwd_checkpoint_move recomputes in backward."""
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    keys_s, keys_C, keys_T = list(state.sites.keys()), list(env.C.keys()), list(env.T.keys())
    four_by_two = ctm_args.projector_method == '4X2'
    norm_inf = ctm_args.ctm_absorb_normalization == 'inf'

    def core(*tensors):
        st = IPEPS(dict(zip(keys_s, tensors[:len(keys_s)])), vertexToSite=state.vertexToSite, lX=state.lX, lY=state.lY)
        ev = ENV(env.chi)
        ev.C = dict(zip(keys_C, tensors[len(keys_s):len(keys_s) + len(keys_C)]))
        ev.T = dict(zip(keys_T, tensors[len(keys_s) + len(keys_C):]))
        # The sites of a move are independent (ctmrg.py:238-275) and a full decomposition at these sizes is latency bound (one
        # workgroup per panel pair): like the forward-only path, they are issued from worker threads, each with its own native
        # context and stream (units.UnitPool; autograd builds the per-site subgraphs from those threads, the phase barriers order
        # the streams).  Grad mode is thread local: the workers inherit the caller's.
        grad_on = torch.is_grad_enabled()

        def proj(coord):
            with torch.set_grad_enabled(grad_on):
                R, Rt = halves(direction, coord, st, ev, four_by_two)
                # (inside a checkpointed move the decomposition is run twice: no warm basis, which the first run would have advanced)
                basis = None if getattr(ctm_args, "fwd_checkpoint_move", False) else _warm_ws(env, coord, R, ctm_args)
                return projectors_from_matrices(R, Rt, env.chi, ctm_args, basis=basis)

        def absb(coord):
            with torch.set_grad_enabled(grad_on):
                new = absorb(direction, coord, st, ev, P, Pt)
                with torch.no_grad():
                    sc = [(t.abs().max() if norm_inf else torch.linalg.vector_norm(t)) for t in new]
                return tuple(t / s for t, s in zip(new, sc))

        pool = None
        a0 = tensors[0]
        # fwd_checkpoint_move: torch's non-reentrant checkpoint intercepts saved tensors through thread-local hooks, so nodes built
        # in worker threads would not be checkpointed at all: the units of a checkpointed move run in the calling thread
        if len(keys_s) > 1 and a0.is_cuda and getattr(ctm_args, "concurrent_units", True) and not getattr(ctm_args, "fwd_checkpoint_move", False):
            import units
            n = env.chi * max(a0.shape[1:]) ** 2
            pool = units.pool_for(get_engine(), len(keys_s), n, a0.is_complex(), est_bytes=40.0 * n * n * a0.element_size())
        import parallel
        if parallel.is_distributed():
            # decided on every rank alike, BEFORE any rank-dependent work: with more ranks than sites (or unequal bond dimensions) a rank
            # that owns no unit must not bail out while the owners go on into the exchange and block there
            if parallel.world()[1] > len(keys_s) or len({st.site(c).shape for c in keys_s}) != 1:
                raise NotImplementedError("distributed differentiable move: uniform bond dimensions and at least one site per rank "
                                          f"(world size {parallel.world()[1]}, {len(keys_s)} sites)")
        mine = parallel.my_units(keys_s)                       # all sites in a single process; this rank's share under torch.distributed
        P, Pt = {}, {}
        for coord, (p_, pt_) in zip(mine, pool.map(proj, mine) if pool is not None and len(mine) > 1 else [proj(c) for c in mine]):
            P[coord], Pt[coord] = p_, pt_
        if parallel.is_distributed():
            # the site-sharded move of ctmrg.ctm_MOVE with differentiable exchanges: every rank gets all projectors, absorbs for its
            # sites, gets all new tensors; cotangents return to the owners through parallel._ExchangeAD
            like = tensors[0]
            shp = {c: tuple(P[mine[0]].shape) for c in keys_s} if mine else None
            if shp is None or len({st.site(c).shape for c in keys_s}) != 1:
                raise NotImplementedError("distributed differentiable move: uniform bond dimensions and at least one site per rank")
            P = parallel.exchange_ad(P, keys_s, shp, like)
            Pt = parallel.exchange_ad(Pt, keys_s, shp, like)
        out_m = pool.map(absb, mine) if pool is not None and len(mine) > 1 else [absb(c) for c in mine]
        if parallel.is_distributed():
            shapes3 = [tuple(t.shape) for t in out_m[0]]
            packs = {c: torch.cat([t.reshape(-1) for t in trip]) for c, trip in zip(mine, out_m)}
            n3 = sum(int(torch.tensor(sh).prod()) for sh in shapes3)
            packs = parallel.exchange_ad(packs, keys_s, {c: (n3,) for c in keys_s}, tensors[0])
            out = []
            for c in keys_s:
                v, off, trip = packs[c], 0, []
                for sh in shapes3:
                    m_ = int(torch.tensor(sh).prod())
                    trip.append(v[off:off + m_].reshape(sh)); off += m_
                out.append(tuple(trip))
        else:
            out = out_m
        return tuple(x for trip in out for x in trip)

    tensors = tuple(state.sites[k] for k in keys_s) + tuple(env.C[k] for k in keys_C) + tuple(env.T[k] for k in keys_T)
    if getattr(ctm_args, "fwd_checkpoint_move", False):
        from torch.utils.checkpoint import checkpoint
        flat = checkpoint(core, *tensors, use_reentrant=False)
    else:
        flat = core(*tensors)
    r1, r2 = _REL[direction]
    for i, coord in enumerate(keys_s):
        nc = state.vertexToSite((coord[0] - direction[0], coord[1] - direction[1]))
        env.C[(nc, r1)], env.C[(nc, r2)], env.T[(nc, direction)] = flat[3 * i], flat[3 * i + 1], flat[3 * i + 2]


def rdm2x2(coord, state, env):
    """rdm2x2 (rdm.py:1362-1592): four open corners -> two halves -> trace; s0 s1 s2 s3 ; s0' s1' s2' s3' with s0 = coord,
    s1 = +x, s2 = +y, s3 = +x+y.  Raw (not symmetrised, not normalised)."""
    x, y = coord
    cLU = c2x2(LU, (x, y), state, env, open_=True)
    cRU = c2x2(RU, (x + 1, y), state, env, open_=True)
    cRD = c2x2(RD, (x + 1, y + 1), state, env, open_=True)
    cLD = c2x2(LD, (x, y + 1), state, env, open_=True)
    up = einsum('akst,kbuv->abstuv', cLU, cRU)
    lo = einsum('akst,bkuv->abstuv', cLD, cRD)
    r = einsum('abstuv,abwxyz->stuvwxyz', up, lo)
    return r.permute(0, 2, 4, 6, 1, 3, 5, 7)


def rdm1x1(coord, state, env):
    """rdm1x1 (rdm.py:71-302): the full one-site environment contracted over the auxiliary legs.  Raw."""
    c = state.vertexToSite(coord)
    a = state.site(coord)
    D = a.shape
    C1, C2, C3, C4 = env.C[(c, (-1, -1))], env.C[(c, (1, -1))], env.C[(c, (1, 1))], env.C[(c, (-1, 1))]
    T1 = _split(env.T[(c, (0, -1))], 1, D[1]); T4 = _split(env.T[(c, (-1, 0))], 2, D[2])
    T3 = _split(env.T[(c, (0, 1))], 0, D[3]); T2 = _split(env.T[(c, (1, 0))], 1, D[4])
    return einsum('ab,bUVc,ce,eRQf,fg,XYhg,ih,aiLM,sULXR,tVMYQ->st', C1, T1, C2, T2, C3, T3, C4, T4, a, a, conj=(9,))


def rdm2x1(coord, state, env):
    """rdm2x1 (rdm.py:304-500): horizontal pair coord, coord + (1,0); s0 s1 ; s0' s1'.  Raw."""
    x, y = coord
    c0, c1 = state.vertexToSite((x, y)), state.vertexToSite((x + 1, y))
    a0, a1 = state.site((x, y)), state.site((x + 1, y))
    D0, D1 = a0.shape, a1.shape
    C1, C4 = env.C[(c0, (-1, -1))], env.C[(c0, (-1, 1))]
    T1a = _split(env.T[(c0, (0, -1))], 1, D0[1]); T4 = _split(env.T[(c0, (-1, 0))], 2, D0[2]); T3a = _split(env.T[(c0, (0, 1))], 0, D0[3])
    C2, C3 = env.C[(c1, (1, -1))], env.C[(c1, (1, 1))]
    T1b = _split(env.T[(c1, (0, -1))], 1, D1[1]); T2 = _split(env.T[(c1, (1, 0))], 1, D1[4]); T3b = _split(env.T[(c1, (0, 1))], 0, D1[3])
    # operand order: every operand shares an index with the running intermediate (the engine contracts left to right, no outer products)
    left = einsum('ab,bUVc,aiLM,ih,XYhg,sULXR,tVMYQ->cRQgst', C1, T1a, T4, C4, T3a, a0, a0, conj=(6,))
    right = einsum('ce,eRQf,fg,jUVc,XYhg,sULXR,tVMYQ->jLMhst', C2, T2, C3, T1b, T3b, a1, a1, conj=(6,))
    return einsum('cRQgst,cRQguv->sutv', left, right)


def rdm1x2(coord, state, env):
    """rdm1x2 (rdm.py:622-826): vertical pair coord, coord + (0,1); s0 s1 ; s0' s1'.  Raw."""
    x, y = coord
    c0, c1 = state.vertexToSite((x, y)), state.vertexToSite((x, y + 1))
    a0, a1 = state.site((x, y)), state.site((x, y + 1))
    D0, D1 = a0.shape, a1.shape
    C1, C2 = env.C[(c0, (-1, -1))], env.C[(c0, (1, -1))]
    T1 = _split(env.T[(c0, (0, -1))], 1, D0[1]); T4a = _split(env.T[(c0, (-1, 0))], 2, D0[2]); T2a = _split(env.T[(c0, (1, 0))], 1, D0[4])
    C4, C3 = env.C[(c1, (-1, 1))], env.C[(c1, (1, 1))]
    T3 = _split(env.T[(c1, (0, 1))], 0, D1[3]); T4b = _split(env.T[(c1, (-1, 0))], 2, D1[2]); T2b = _split(env.T[(c1, (1, 0))], 1, D1[4])
    up = einsum('ab,bUVc,ce,aiLM,eRQf,sULXR,tVMYQ->iXYfst', C1, T1, C2, T4a, T2a, a0, a0, conj=(6,))
    lo = einsum('jh,XYhg,fg,ijLM,eRQf,sULXR,tVMYQ->iUVest', C4, T3, C3, T4b, T2b, a1, a1, conj=(6,))
    return einsum('iXYfst,iXYfuv->sutv', up, lo)


def wants_grad(state, env):
    return needs_grad(*state.sites.values(), *env.C.values(), *env.T.values())
