"""Directional CTMRG for a generic unit cell (reference ctm/generic/ctmrg.py:18-110,179-319,324-804).

`run` and `ctm_MOVE` keep the reference's signatures.  One move = projectors for every site from the
OLD environment, then absorb+truncate for every site, then each new tensor divided by its own max-abs
and written to the site at `coord - direction`.  With torch.distributed initialised the per-site units
of both phases are sharded over the ranks (one MI355X each) with one all-gather after each phase
(parallel.py); without it everything runs on the current device.
"""
import os
import time
import copy
import logging
from math import ceil
import torch
import config as cfg
from backend import get_engine
from _native import NativeError, CTM_ERR_NOMEM
import parallel
from ctm.generic.ctm_projectors import ctm_get_projectors_4x4, ctm_get_projectors_4x2, _trunc_cfg, _unit_inputs, _sync_warm_tol, SVD_METHODS, _note_method, \
    rectangular_unit
from ctm.generic.ctm_components import _halves_t

log = logging.getLogger(__name__)

# per direction: relative vectors of (C1, T1, T, T2, C2), shift to the neighbour whose projectors are P1, Pt1
_ABS = {
    (0, -1): (((1, -1), (1, 0), (0, -1), (-1, 0), (-1, -1)), (1, 0)),
    (-1, 0): (((-1, -1), (0, -1), (-1, 0), (0, 1), (-1, 1)), (0, -1)),
    (0, 1): (((-1, 1), (-1, 0), (0, 1), (1, 0), (1, 1)), (-1, 0)),
    (1, 0): (((1, 1), (0, 1), (1, 0), (0, -1), (1, -1)), (0, 1)),
}
# Ownership of the projector units when sharded over ranks: window `c` of a move goes to the round-robin owner of the site
# c + shift.  A move leaves the two enlarged corners on its far side unchanged; with these shifts (default move sequence
# UP, LEFT, DOWN, RIGHT) the window that needs them in the NEXT move is computed by the rank that built them, so the corner
# cache hits across ranks as it does in one process (8 of 16 corner contractions per move).
_OWNER_SHIFT = {(0, -1): (0, 0), (-1, 0): (1, 0), (0, 1): (1, -1), (1, 0): (0, -1)}
# where the new tensors go (ctmrg.py:302-309)
_REL = {(0, -1): ((1, -1), (-1, -1)), (-1, 0): ((-1, -1), (-1, 1)), (0, 1): ((-1, 1), (1, 1)), (1, 0): ((1, 1), (1, -1))}


def run(state, env, conv_check=None, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
    """Sweep until `conv_check(state, env, history, ctm_args=)` says so or ctm_max_iter sweeps.
    Returns (env, history, t_ctm, t_obs).  t_ctm is measured on this rank with device syncs around
    each sweep (and, unlike the reference's quirk at ctmrg.py:101-108, includes the last sweep)."""
    # ctm_force_dl (pre-fused double-layer tensors, ctmrg.py:51-61) is a memory/speed trade-off of the reference with
    # identical mathematics; the engine always contracts layer by layer, so the flag is accepted and has no effect.
    eng = get_engine()

    def _ctmrg_iter(i, loc_ctm_args=ctm_args):
        for direction in loc_ctm_args.ctm_move_sequence:
            diagnostics = {"ctm_i": i, "ctm_d": direction} if loc_ctm_args.verbosity_projectors > 0 else None
            n = state.lX if direction in [(-1, 0), (1, 0)] else state.lY
            for rep in range(n):
                env.__dict__["_move_index"] = (i, direction, rep)       # the differentiable route keeps one basis per move of a run
                ctm_MOVE(direction, state, env, ctm_args=loc_ctm_args, global_args=global_args,
                         verbosity=loc_ctm_args.verbosity_ctm_move, diagnostics=diagnostics)

    t_obs = t_ctm = 0.
    history = None
    if ctm_args.ctm_warmup_iter >= 0:
        wargs = copy.deepcopy(ctm_args)
        wargs.projector_svd_method = ctm_args.warmup_projector_svd_method
        maxD = max(state.get_aux_bond_dims())
        with torch.no_grad():                                   # the warm-up carries no gradient (ctmrg.py:80)
            for i in range(max(ctm_args.ctm_warmup_iter, ceil(env.chi / maxD ** 2))):
                t0 = time.perf_counter(); _ctmrg_iter(i, loc_ctm_args=wargs); eng.sync(); t_ctm += time.perf_counter() - t0
    for i in range(ctm_args.ctm_max_iter):
        t0 = time.perf_counter()
        _ctmrg_iter(i)
        eng.sync()
        t1 = time.perf_counter()
        t_ctm += t1 - t0
        if conv_check is not None:
            converged, history = conv_check(state, env, history, ctm_args=ctm_args)
            t_obs += time.perf_counter() - t1
            if converged:
                if ctm_args.verbosity_ctm_convergence > 0:
                    print(f"CTMRG  converged at iter= {i}, history= {history['conv_crit'][-1] if isinstance(history, dict) else ''}")
                break
    return env, history, t_ctm, t_obs


def _absorb_tensors(direction, coord, state, env, P, Pt):
    vecs, sh = _ABS[direction]
    c = state.vertexToSite(coord)
    nb = state.vertexToSite((coord[0] + sh[0], coord[1] + sh[1]))
    return (env.C[(c, vecs[0])], env.T[(c, vecs[1])], env.T[(c, vecs[2])], env.T[(c, vecs[3])], env.C[(c, vecs[4])],
            state.site(coord), P[c], Pt[c], P[nb], Pt[nb])


# axis of the new (truncated) bond in nC1, nC2 and the two such axes of nT per direction (absorb output layouts)
_NEW_AX = {(0, -1): (0, 1, (0, 2)), (-1, 0): (0, 0, (0, 1)), (0, 1): (1, 1, (1, 2)), (1, 0): (0, 1, (0, 2))}


def _absorb(direction, coord, state, env, P, Pt, ctm_args, normalize=False):
    eng = get_engine()
    tens = _absorb_tensors(direction, coord, state, env, P, Pt)
    ncol = env.__dict__.get("_ncol")
    if ncol:
        # Columns of the projectors beyond the last S/S[0] > projector_svd_reltol are exact zeros, so the rows / columns of
        # nC1, nC2, nT they produce are exact zeros too: absorb with the non-zero prefix only and pad.  (The reference multiplies
        # the zeros through, ctmrg.py:351-425.)  Same numbers; pays when the truncation leaves many masked columns.
        chi = env.chi
        _, sh = _ABS[direction]
        c = state.vertexToSite(coord)
        nb = state.vertexToSite((coord[0] + sh[0], coord[1] + sh[1]))
        y = max(ncol.get((direction, c), chi), ncol.get((direction, nb), chi))
        yc = min(chi, max(16, (y + 15) // 16 * 16))
        if 2 * yc <= chi and all(t.shape[1] == chi for t in tens[6:10]):
            tens = tens[:6] + tuple(t[:, :yc].contiguous() for t in tens[6:10])
            nC1, nC2, nT = eng.absorb(direction, tens, normalize=normalize)
            a1, a2, aT = _NEW_AX[direction]

            def pad(x, axes):
                shape = list(x.shape)
                for ax in axes: shape[ax] = chi
                out = torch.zeros(shape, dtype=x.dtype, device=x.device)
                out[tuple(slice(0, x.shape[i]) for i in range(x.dim()))] = x
                return out
            return pad(nC1, (a1,)), pad(nC2, (a2,)), pad(nT, aT)
    return eng.absorb(direction, tens, normalize=normalize)


def absorb_truncate_CTM_MOVE_UP(coord, state, env, P, Pt, ctm_args=cfg.ctm_args): return _absorb((0, -1), coord, state, env, P, Pt, ctm_args)
def absorb_truncate_CTM_MOVE_LEFT(coord, state, env, P, Pt, ctm_args=cfg.ctm_args): return _absorb((-1, 0), coord, state, env, P, Pt, ctm_args)
def absorb_truncate_CTM_MOVE_DOWN(coord, state, env, P, Pt, ctm_args=cfg.ctm_args): return _absorb((0, 1), coord, state, env, P, Pt, ctm_args)
def absorb_truncate_CTM_MOVE_RIGHT(coord, state, env, P, Pt, ctm_args=cfg.ctm_args): return _absorb((1, 0), coord, state, env, P, Pt, ctm_args)
def absorb_truncate_CTM_MOVE_UP_c(*t): return get_engine().absorb((0, -1), t[:10], normalize=False)
def absorb_truncate_CTM_MOVE_LEFT_c(*t): return get_engine().absorb((-1, 0), t[:10], normalize=False)
def absorb_truncate_CTM_MOVE_DOWN_c(*t): return get_engine().absorb((0, 1), t[:10], normalize=False)
def absorb_truncate_CTM_MOVE_RIGHT_c(*t): return get_engine().absorb((1, 0), t[:10], normalize=False)


def ctm_MOVE(direction, state, env, ctm_args=cfg.ctm_args, global_args=cfg.global_args, verbosity=0, diagnostics=None):
    if ctm_args.projector_method == '4X4':
        get_projectors = ctm_get_projectors_4x4
    elif ctm_args.projector_method == '4X2':
        get_projectors = ctm_get_projectors_4x2
    else:
        raise ValueError("Invalid Projector method: " + str(ctm_args.projector_method))
    if direction not in _ABS:
        raise ValueError("Invalid direction: " + str(direction))
    # every route of the move (the whole-move native call below never enters ctm_get_projectors_*): an unknown truncation method is
    # refused and an aliased one announced once, as the reference's branch on the method does (ctm_projectors.py:213-257)
    if ctm_args.projector_svd_method not in SVD_METHODS:
        raise ValueError(f"Projector svd method \"{ctm_args.projector_svd_method}\" not implemented")
    _note_method(ctm_args.projector_svd_method)
    norm_kind = 1 if ctm_args.ctm_absorb_normalization == 'inf' else 2      # anything else is the 2-norm (ctmrg.py:212-214)
    from ctm.generic import ctm_ad
    if ctm_ad.wants_grad(state, env):
        # a tensor requires grad: the explicit, differentiable route (graph of native contraction / SVD nodes); under
        # torch.distributed the sites are sharded as below and the exchanges are autograd nodes (parallel.exchange_ad)
        return ctm_ad.ctm_MOVE(direction, state, env, ctm_args)
    eng = get_engine()
    coords = list(state.sites.keys())
    # which repetition of this direction's move inside a sweep this call is (consecutive calls with one direction cycle through them):
    # the warm-start workspaces of the projector units are kept per repetition (ctm_projectors._unit_inputs)
    nrep = max(1, state.lX if direction in [(-1, 0), (1, 0)] else state.lY)
    cnt = env.__dict__.setdefault("_rep_count", {})
    env.__dict__.setdefault("_rep", {})[direction] = cnt.get(direction, 0) % nrep
    _ctm_MOVE_units(direction, state, env, ctm_args, global_args, diagnostics, get_projectors, norm_kind, eng, coords)
    # only now: a move that raised has not happened, and the repetition counter (which warm-start workspace the next call uses) stays
    cnt[direction] = cnt.get(direction, 0) + 1


def _ctm_MOVE_units(direction, state, env, ctm_args, global_args, diagnostics, get_projectors, norm_kind, eng, coords):
    # twice as many ranks as sites (8 GPUs, 4-site cell): the two ranks {i, i + Nsites} SHARE unit i -- every corner pass of its truncation is
    # split by output columns inside the native solver and all-gathered in the pair (Engine.set_group; float64, 4X4 projectors); everything
    # else of the unit is replicated in the pair, and the exchanges between units go on as between single ranks (owner = the lower rank)
    shared = parallel.unit_group_size(len(coords)) == 2 and getattr(ctm_args, "share_units", True) and ctm_args.projector_method == '4X4' \
        and not next(iter(env.C.values())).dtype.is_complex and hasattr(eng, "set_group")
    mine = parallel.my_units(coords, shared=shared)
    chi = env.chi
    # number of non-zero projector columns per site of THIS move (filled by the fused projector path)
    # (worth its host-side bookkeeping where an absorb takes milliseconds: n >= absorb_skip_min_n)
    if getattr(ctm_args, "absorb_skip_zero_columns", True) and ctm_args.projector_method == '4X4' \
            and max(_proj_rows(direction, c, state, chi) for c in coords) >= getattr(ctm_args, "absorb_skip_min_n", 8192):
        env.__dict__["_ncol"] = {}
    else:
        env.__dict__.pop("_ncol", None)
    like = next(iter(env.C.values()))

    # my units of a phase are independent: issue them concurrently (own context + stream each) when that pays
    pool = None
    if getattr(ctm_args, "concurrent_units", True) and len(mine) > 1:
        import units
        # large n: all units of the move in flight -- while their truncations run the block Krylov solver (long latency-bound stages
        # that only overlap with other units') and, since the move is one native call (ctm_move: no host-language thread hop per unit),
        # also on low-rank environments: D = 8 chi = 256, prescribed state, 542 -> 530 ms per sweep with four instead of two
        # (rounds 2-4, units issued from Python threads: no difference, so two were kept for their smaller workspace)
        krylov = env.__dict__.get("_krylov_units", False)
        nmax = max(_proj_rows(direction, c, state, chi) for c in mine)
        # workspace of one unit: 14 n^2 elements when its four enlarged corners live in the arena; with the corner cache they live in the
        # cache buffers and the arena holds the corner intermediates and the Krylov bases (measured high-water marks: 3.3 n^2 elements at
        # n = 16384 float64) -- what lets configs[4] (n = 24576 complex128, 155 GB of cached corners) keep more than one unit in flight
        elem = 16 if like.dtype.is_complex else 8
        cached = getattr(ctm_args, "corner_cache", True) and ctm_args.projector_method == '4X4' and hasattr(eng, "corner_numel") and \
            4 * len(coords) * nmax * nmax * elem <= 0.6 * torch.cuda.get_device_properties(like.device).total_memory
        pool = units.pool_for(eng, len(mine), nmax, like.dtype.is_complex, est_bytes=(4.5 * nmax * nmax * elem) if cached else None,
                              large_n_units=None if krylov else int(os.environ.get("CTM_LOWRANK_UNITS", 4)))
        # a move of this environment ran out of workspace with more units in flight (the estimate above is one measured high-water
        # mark scaled by n^2): stay at the width that worked
        cap = env.__dict__.get("_units_cap")
        if pool is not None and cap is not None:
            pool = units.PoolView(pool.pool, min(pool.n, cap)) if cap >= 2 else None
    if pool is not None and nmax >= 8192 and hasattr(eng, "trim_own") and eng.own_stat("arena_total") > 2 ** 30:
        eng.trim_own()       # the units run on the workers' contexts: an arena this engine grew in an earlier (serial) phase is tens of GB of idle HBM
    lz_before = eng.stat("lz_hits") if hasattr(eng, "stat") else 0

    def _each(fn, items, stagger=0.0):
        return pool.map(fn, items, stagger=stagger) if pool is not None else [fn(it) for it in items]

    if getattr(ctm_args, "native_move", True) and ctm_args.projector_method == '4X4' and hasattr(eng, "move") and not parallel.is_distributed() \
            and float(getattr(ctm_args, "unit_stagger_ms", 0.0)) == 0.0 and not any(rectangular_unit(direction, c, state, env) for c in coords):
        # the whole move in ONE native call (ctm_move, include/ctm_hip.h; reference seam ctm_MOVE_c, ctmrg.py:233-283): both phases and
        # the threads that overlap their units live in the library, on the worker contexts of the device's unit pool
        workers = pool.engines() if pool is not None else ()
        for e in (eng,) + tuple(workers):
            _sync_warm_tol(e, ctm_args)
        index = {c: i for i, c in enumerate(coords)}
        ulist, fresh_all = [], []
        for c in coords:
            t16, basis, corners, fresh = _unit_inputs(eng, direction, c, state, env, ctm_args)
            vecs, sh = _ABS[direction]
            site = state.vertexToSite(c)
            nb = state.vertexToSite((c[0] + sh[0], c[1] + sh[1]))
            a6 = (env.C[(site, vecs[0])], env.T[(site, vecs[1])], env.T[(site, vecs[2])], env.T[(site, vecs[3])], env.C[(site, vecs[4])], state.site(c))
            ulist.append({"t16": t16, "basis": basis, "corners": corners, "absorb6": a6, "nb": index[nb]})
            fresh_all += fresh
        try:
            res = eng.move(direction, ulist, chi, _trunc_cfg(eng, ctm_args), normalize=norm_kind,
                           skip_zero_columns=env.__dict__.get("_ncol") is not None, workers=workers)
        except NativeError as e:
            if e.status != CTM_ERR_NOMEM or not workers:
                raise
            # out of HBM with len(workers) units in flight: a move writes nothing before it has succeeded (new tensors, and the corner
            # cache entries are committed below), so give every workspace arena back and repeat it with half the units in flight --
            # serially on this engine's own context at the end -- and keep later moves of this environment at that width
            width = len(workers)
            while True:
                eng.trim()                   # (this engine's arenas and every worker's)
                width //= 2
                env.__dict__["_units_cap"] = max(1, width)
                workers = tuple(workers[:width]) if width >= 2 else ()
                try:
                    res = eng.move(direction, ulist, chi, _trunc_cfg(eng, ctm_args), normalize=norm_kind,
                                   skip_zero_columns=env.__dict__.get("_ncol") is not None, workers=workers)
                    break
                except NativeError as e2:
                    if e2.status != CTM_ERR_NOMEM or not workers:
                        raise
        for key, entry in fresh_all:                  # only after the call succeeded: the buffers now hold these corners
            env.__dict__["_corner_cache"][key] = entry
        if env.__dict__.get("_ncol") is not None:
            for c, r in zip(coords, res):
                env.__dict__["_ncol"][(direction, state.vertexToSite(c))] = r[6]
        env.__dict__["_krylov_units"] = eng.stat("lz_hits") > lz_before
        r1, r2 = _REL[direction]
        for coord, r in zip(coords, res):
            nc = state.vertexToSite((coord[0] - direction[0], coord[1] - direction[1]))
            env.C[(nc, r1)], env.C[(nc, r2)], env.T[(nc, direction)] = r[3], r[4], r[5]
        return

    # phase A: projectors of my sites from the old env
    ownersA, mineA = None, mine
    if parallel.is_distributed():
        ownersA = parallel.owners_shifted(coords, state.vertexToSite, _OWNER_SHIFT[direction])
        mineA = [c for c, o in zip(coords, ownersA) if o == (parallel.world()[0] % len(coords) if shared else parallel.world()[0])]
    P, Pt = {}, {}
    stagger = float(getattr(ctm_args, "unit_stagger_ms", 0.0)) * 1e-3 if max(_proj_rows(direction, c, state, chi) for c in coords) >= 8192 else 0.0
    if shared:
        # (one unit per rank: issued serially on this engine; the group is attached only while its ranks make the same calls)
        nmax = max(_proj_rows(direction, c, state, chi) for c in coords)
        parallel.prepare_groups([parallel.unit_group(i, len(coords)) for i in range(len(coords))])      # (collective, cached: every rank names every group)
        eng.set_group(parallel.unit_group(parallel.world()[0] % len(coords), len(coords)), capacity_doubles=(2 * chi + 128) * ((nmax + 1) // 2))
    try:
        for coord, (p_, pt_) in zip(mineA, _each(lambda c: get_projectors(direction, c, state, env, ctm_args, global_args,
                                                                                  diagnostics=diagnostics), mineA, stagger)):
            P[coord], Pt[coord] = p_, pt_
    finally:
        if shared:
            eng.set_group(None)
    if hasattr(eng, "stat"):
        env.__dict__["_krylov_units"] = eng.stat("lz_hits") > lz_before
    if parallel.is_distributed():
        shp = {}
        for coord in coords:
            R_n = _proj_rows(direction, coord, state, chi)
            shp[coord] = (R_n, min(chi, R_n))
        w = None
        if env.__dict__.get("_ncol") is not None and len({s_ for s_ in shp.values()}) == 1:
            # only the non-zero prefix of the projector columns travels: agree on its (rounded) width first
            nc = env.__dict__["_ncol"]
            w = parallel.allreduce_max_int(max([nc.get((direction, state.vertexToSite(c)), chi) for c in mineA] + [0]), like.device)
            w = min(chi, max(16, (w + 15) // 16 * 16))
            if 2 * w > chi:
                w = None
        if w is not None:
            shw = {c: (shp[c][0], w) for c in coords}
            Pw = parallel.exchange({c: P[c][:, :w].contiguous() for c in mineA}, coords, shw, like, owners=ownersA)
            Ptw = parallel.exchange({c: Pt[c][:, :w].contiguous() for c in mineA}, coords, shw, like, owners=ownersA)

            def widen(x, c):
                full = torch.zeros(shp[c], dtype=x.dtype, device=x.device)
                full[:, :w] = x
                return full
            P = {c: widen(Pw[c], c) for c in coords}
            Pt = {c: widen(Ptw[c], c) for c in coords}
        else:
            P = parallel.exchange(P, coords, shp, like, owners=ownersA)
            Pt = parallel.exchange(Pt, coords, shp, like, owners=ownersA)
        if env.__dict__.get("_ncol") is not None:
            # every rank needs the count of non-zero projector columns of every site: read it off the gathered projectors
            # (the non-zero columns are a prefix)
            env.__dict__["_ncol"] = {(direction, state.vertexToSite(c)): int((P[c] != 0).any(0).sum()) for c in coords}

    # phase B: absorb + normalise my sites
    new = dict(zip(mine, _each(lambda c: _absorb(direction, c, state, env, P, Pt, ctm_args, normalize=norm_kind), mine)))
    if parallel.is_distributed():
        packs = {c: torch.cat([t.reshape(-1) for t in v]) for c, v in new.items()}
        shp = {c: (2 * chi * chi + chi * chi * _out_D2(direction, state.site(c)),) for c in coords}
        packs = parallel.exchange(packs, coords, shp, like)
        new = {}
        for c in coords:
            D2 = _out_D2(direction, state.site(c))
            v = packs[c]
            tshape = {(0, -1): (chi, D2, chi), (-1, 0): (chi, chi, D2), (0, 1): (D2, chi, chi), (1, 0): (chi, D2, chi)}[direction]
            new[c] = (v[:chi * chi].view(chi, chi), v[chi * chi:2 * chi * chi].view(chi, chi), v[2 * chi * chi:].view(tshape))

    r1, r2 = _REL[direction]
    for coord in coords:
        nc = state.vertexToSite((coord[0] - direction[0], coord[1] - direction[1]))
        env.C[(nc, r1)], env.C[(nc, r2)], env.T[(nc, direction)] = new[coord]


def _out_D2(direction, A):
    return A.size({(0, -1): 3, (-1, 0): 4, (0, 1): 1, (1, 0): 2}[direction]) ** 2


def _proj_rows(direction, coord, state, chi):
    # n = chi * D^2 of the truncated bond: the leg of the anchor site pointing along the move's cut
    A = state.site(coord)
    leg = {(0, -1): 2, (-1, 0): 3, (0, 1): 4, (1, 0): 1}[direction]
    return chi * A.size(leg) ** 2
