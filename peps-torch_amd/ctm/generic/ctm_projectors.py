"""Projectors of a directional move (reference ctm/generic/ctm_projectors.py:14-64,142-293)."""
import config as cfg
from backend import get_engine
from ctm.generic.ctm_components import _halves, _halves_t, _HALVES, _corner


# The reference switches between LAPACK gesdd, ARPACK, PROPACK and randomised SVD (ctm_projectors.py:225-255); here every
# choice lands on the same engine (leading-chi iteration / block Krylov / full Jacobi chosen by the spectrum, full-SVD
# semantics), so the names are accepted as synonyms.  'AF' (arrayfire) has no counterpart.
SVD_METHODS = ['DEFAULT', 'GESDD', 'GESDD_CPU', 'ARP', 'PROPACK', 'RSVD', 'RSVD_CUSTOM', 'QR']


_noted = set()


def _note_method(name):
    """The reference's alternative truncation routes (ARPACK, PROPACK, randomised, QR) are names for ONE native engine here; say
    so once instead of aliasing silently."""
    if name not in ('DEFAULT', 'GESDD', 'GESDD_CPU') and name not in _noted:
        _noted.add(name)
        import warnings
        warnings.warn(f"projector_svd_method={name}: executed by the native truncation engine (leading-chi iteration / block "
                      "Krylov / full Jacobi chosen by the spectrum, full-SVD semantics); the reference's method-specific "
                      "parameters are not used", stacklevel=3)


def _trunc_cfg(eng, ctm_args):
    return eng.cfg(svd_reltol=ctm_args.projector_svd_reltol, eps_multiplet=ctm_args.projector_eps_multiplet,
                   multiplet_abstol=ctm_args.projector_multiplet_abstol, keep_multiplets=True)


def _unit_inputs(eng, direction, coord, state, env, ctm_args):
    """What one (direction, site) projector unit of the fused path takes: the 16 tensors of its 2x2 window, its warm-start workspace
    (kept with the environment) and its cached enlarged corners ([(buffer, valid)] * 4 or None, plus the cache entries to store once
    the unit has run)."""
    t16 = _halves_t(direction, coord, state, env)
    basis = None
    if getattr(ctm_args, "projector_warm_start", True) and hasattr(eng, "warm_basis"):
        a = t16[3]
        n = env.chi * a.shape[{(0, -1): 2, (-1, 0): 3, (0, 1): 4, (1, 0): 1}[direction]] ** 2     # truncated bond chi * D_cut^2
        ws = env.__dict__.setdefault("_warm", {})
        # one workspace per (direction, site, repetition): a sweep repeats the move of a direction lX / lY times (ctmrg.py:62-66), and the
        # operators a site's unit sees in the repetitions of one sweep are different members of the fixed-point cycle -- same singular
        # values to 1e-10, singular vectors O(1) apart (measured: a basis handed from one repetition to the next has residual 3e-2 s_0) --
        # while the same repetition of the next sweep sees the operator again, moved by what the sweep moved the environment
        key = (direction, coord, env.__dict__.get("_rep", {}).get(direction, 0))
        k = env.chi + 1 if env.chi < n else n
        b = ws.get(key)
        if b is None or b.shape[1] != n or b.shape[0] != eng.warm_rows(env.chi, n, a.dtype)[0] or b.device != a.device:
            b = ws[key] = eng.warm_basis(env.chi, n, a.dtype)
        basis = b
    corners, fresh = _cached_corners(eng, direction, coord, state, env, t16, ctm_args)
    return t16, basis, corners, fresh


_CUT_LEG = {(0, -1): 2, (-1, 0): 3, (0, 1): 4, (1, 0): 1}


def rectangular_unit(direction, coord, state, env):
    """Bond dimensions that differ along the cut of this (direction, site) unit: its halves R, Rt are a x b with a != b (the reference only
    asserts R.shape == Rt.shape, ctm_projectors.py:209).  The fused implicit-operator path truncates square halves; such a unit goes the
    explicit way (ctm_halves -> ctm_projectors_rect)."""
    t16 = _halves_t(direction, coord, state, env)
    leg = _CUT_LEG[direction]
    return t16[3].shape[leg] != t16[7].shape[leg]


def _sync_warm_tol(eng, ctm_args):
    wtol = float(getattr(ctm_args, "projector_warm_tol", 0.0) or 0.0)
    if getattr(eng, "_warm_tol", 0.0) != wtol and hasattr(eng, "set_option"):      # (per engine: the units of a move run on worker engines)
        eng.set_option("warm_accept_tol", wtol)
        eng._warm_tol = wtol


def ctm_get_projectors_4x4(direction, coord, state, env, ctm_args=cfg.ctm_args, global_args=cfg.global_args,
                           diagnostics=None):
    if direction not in [(0, -1), (-1, 0), (0, 1), (1, 0)]:
        raise ValueError("Invalid direction: " + str(direction))
    if ctm_args.projector_svd_method not in SVD_METHODS:
        raise ValueError(f"Projector svd method \"{ctm_args.projector_svd_method}\" not implemented")
    _note_method(ctm_args.projector_svd_method)
    eng = get_engine()
    if hasattr(eng, "projectors_4x4") and not rectangular_unit(direction, coord, state, env):
        # fused native path: corners -> implicit M = R^T Rt -> P, Pt (halves never materialised).  The environment
        # remembers, per (direction, site), the right singular basis of the previous sweep: the leading-chi iteration of
        # the next sweep starts from it (residual-verified either way; `projector_warm_start=False` disables it).
        t16, basis, corners, fresh = _unit_inputs(eng, direction, coord, state, env, ctm_args)
        _sync_warm_tol(eng, ctm_args)
        P, Pt, S = eng.projectors_4x4(direction, t16, env.chi, _trunc_cfg(eng, ctm_args), return_S=True, basis=basis, corners=corners)
        for key, entry in fresh:                  # only after the call succeeded: the buffers now hold these corners
            env.__dict__["_corner_cache"][key] = entry
        ncol = env.__dict__.get("_ncol")
        if ncol is not None:
            # projector columns are scaled by rsqrt(S) where S/S[0] > projector_svd_reltol and are exact zeros elsewhere
            # (ctm_projectors.py:266-270); S is descending, so the non-zero columns are a prefix: its length
            ncol[(direction, state.vertexToSite(coord))] = int((S > ctm_args.projector_svd_reltol * S[0]).sum())
        return P, Pt
    R, Rt = _halves(direction, coord, state, env)
    return ctm_get_projectors_from_matrices(R, Rt, env.chi, ctm_args, global_args, diagnostics=diagnostics)


def _cached_corners(eng, direction, coord, state, env, t16, ctm_args):
    """Enlarged corners kept with the environment.  A corner depends on (C, T1, T2, a) of ITS site only, and a move rebinds
    two corner matrices and one T per site (ctmrg.py:302-319): the two corner types of the opposite side are the same
    tensors in the next move.  The reference rebuilds all four in every move (ctm_components.py:10-265); here a corner is
    rebuilt only when one of its inputs is a different (or in-place modified) tensor object.  Returns ([(buffer, valid)] * 4,
    [(key, entry)] to store after the call) or (None, []) when caching is off or the 4 x Nsites buffers do not fit."""
    import torch
    if not getattr(ctm_args, "corner_cache", True) or not hasattr(eng, "corner_numel"):
        return None, []
    a0 = t16[3]
    nsites = len(state.sites)
    sizes = [eng.corner_numel(c, t16[4 * i], t16[4 * i + 3]) for i, (c, _) in enumerate(_HALVES[direction])]
    total = torch.cuda.get_device_properties(a0.device).total_memory
    # the corners of the unit being computed live in these buffers instead of the workspace arena, so the extra footprint is
    # 4 (Nsites - 1) corners
    if 4 * nsites * max(sizes) * 8 > 0.6 * total:
        return None, []
    cache = env.__dict__.setdefault("_corner_cache", {})
    corners, fresh = [], []
    for i, (c, sh) in enumerate(_HALVES[direction]):
        ins = tuple(t16[4 * i:4 * i + 4])
        key = (c, state.vertexToSite((coord[0] + sh[0], coord[1] + sh[1])))
        e = cache.get(key)
        ok = (e is not None and e[2] == sizes[i] and all(x is y and x._version == v for x, y, v in zip(ins, e[0], e[1])))
        if ok:
            corners.append((e[3], True))
            continue
        buf = e[3] if (e is not None and e[2] == sizes[i] and e[3].device == a0.device) else \
            torch.empty(sizes[i], dtype=torch.float64, device=a0.device)
        cache.pop(key, None)                      # the buffer is about to be overwritten
        corners.append((buf, False))
        fresh.append((key, (ins, tuple(x._version for x in ins), sizes[i], buf)))
    return corners, fresh


# plain transposes applied to the corner next to the cut in each half (ctm_projectors.py:113-131): (R, Rt) per direction
_T4X2 = {(0, -1): (False, True), (-1, 0): (False, False), (0, 1): (True, True), (1, 0): (False, True)}


def ctm_get_projectors_4x2(direction, coord, state, env, ctm_args=cfg.ctm_args, global_args=cfg.global_args, diagnostics=None):
    """Projectors from the two enlarged corners of the 4x2 (2x4) network (reference ctm_projectors.py:66-136)."""
    if direction not in _T4X2:
        raise ValueError("Invalid direction: " + str(direction))
    (cR, shR), _, (cRt, shRt), _ = _HALVES[direction]
    R = _corner(cR, (coord[0] + shR[0], coord[1] + shR[1]), state, env, mode='sl')
    Rt = _corner(cRt, (coord[0] + shRt[0], coord[1] + shRt[1]), state, env, mode='sl')
    tR, tRt = _T4X2[direction]
    eng = get_engine()
    if tR: R = eng.permute(R, (1, 0))
    if tRt: Rt = eng.permute(Rt, (1, 0))
    return ctm_get_projectors_from_matrices(R, Rt, env.chi, ctm_args, global_args, diagnostics=diagnostics)


def ctm_get_projectors_from_matrices(R, Rt, chi, ctm_args=cfg.ctm_args, global_args=cfg.global_args, diagnostics=None):
    """M = R^T Rt ; U S V^T = M (leading chi) ; P = R conj(U) S^-1/2 , Pt = Rt V S^-1/2."""
    assert R.shape == Rt.shape
    assert len(R.shape) == 2
    if ctm_args.projector_svd_method not in SVD_METHODS:
        raise ValueError(f"Projector svd method \"{ctm_args.projector_svd_method}\" not implemented")
    eng = get_engine()
    return eng.projectors(R, Rt, chi, _trunc_cfg(eng, ctm_args))
