"""Reduced density matrices from the converged environment (reference ctm/generic/rdm.py:38-68,
71-302, 304-500, 622-826, 1306-1592).  The network contraction (open enlarged corners -> halves ->
trace) runs natively; the final p^N x p^N hermitisation / clamping / trace normalisation
(`_sym_pos_def_rdm`) is host-side on <= 256 numbers."""
import logging
import torch
import config as cfg
from backend import get_engine
from ctm.generic.ctm_components import _corner_t, LU, RU, RD, LD

log = logging.getLogger(__name__)


def _cast_to_real(t, fail_on_check=False, warn_on_check=True, imag_eps=1.0e-8, who="unknown", **kwargs):
    if t.is_complex():
        if abs(t.imag) / (abs(t.real) + 1.0e-8) > imag_eps:
            if warn_on_check: log.warning("Unexpected imaginary part " + who + " " + str(t))
            if fail_on_check: raise RuntimeError("Unexpected imaginary part " + who + " " + str(t))
        return t.real
    return t


def _sym_pos_def_matrix(rdm, sym_pos_def=False, verbosity=0, who="unknown", **kwargs):
    rdm = 0.5 * (rdm + rdm.conj().t())
    if sym_pos_def:
        D, U = torch.linalg.eigh(rdm.detach().cpu())   # <= 16 x 16, host
        if D.min() < 0:
            log.info(f"{who} max(diag(rdm)) {D.max()} min(diag(rdm)) {D.min()}")
            D = torch.clamp(D, min=0)
            # the reference overwrites the values under no_grad (rdm.py:44-53): the projection carries no gradient of its own
            rdm = rdm + ((U @ torch.diag(D).to(U.dtype) @ U.conj().t()).to(rdm.device) - rdm).detach()
    norm = _cast_to_real(rdm.diagonal().sum(), who=who, **kwargs)
    return rdm / norm


def _sym_pos_def_rdm(rdm, sym_pos_def=False, verbosity=0, who=None, **kwargs):
    assert len(rdm.size()) % 2 == 0, "invalid rank of RDM"
    nsites = len(rdm.size()) // 2
    shape = rdm.size()
    n = 1
    for d in shape[:nsites]: n *= d
    m = _sym_pos_def_matrix(rdm.reshape(n, -1), sym_pos_def=sym_pos_def, verbosity=verbosity, who=who)
    return m.reshape(shape)


def rdm2x2(coord, state, env, open_sites=[0, 1, 2, 3], unroll=[], checkpoint_unrolled=False, checkpoint_on_device=False,
           sym_pos_def=False, force_cpu=False, verbosity=0, global_args=cfg.global_args):
    """rho(s0 s1 s2 s3 ; s0' s1' s2' s3') of the plaquette coord, +x, +y, +x+y (s0 s1 / s2 s3)."""
    open_sites = sorted(set(open_sites))
    if any(i not in (0, 1, 2, 3) for i in open_sites):
        raise ValueError("rdm2x2: open_sites must be a subset of [0,1,2,3]")
    x, y = coord
    t = _corner_t(LU, (x, y), state, env) + _corner_t(RU, (x + 1, y), state, env) \
        + _corner_t(RD, (x + 1, y + 1), state, env) + _corner_t(LD, (x, y + 1), state, env)
    eng = get_engine()
    from ctm.generic import ctm_ad
    if ctm_ad.wants_grad(state, env):
        # differentiable route: the same contraction as a graph of native contraction nodes
        raw = ctm_ad.rdm2x2(coord, state, env)
        if open_sites != [0, 1, 2, 3]:
            ket, bra = list("abcd"), list("efgh")
            for i in range(4):
                if i not in open_sites:
                    bra[i] = ket[i]
            keep = [ket[i] for i in open_sites] + [bra[i] for i in open_sites]
            raw = torch.einsum("".join(ket + bra) + "->" + "".join(keep), raw)
        return _sym_pos_def_rdm(raw.contiguous(), sym_pos_def=sym_pos_def, verbosity=verbosity, who="rdm2x2")
    if hasattr(eng, "trim"):
        # the open halves need n^2 (p^4 + 2 p^2) elements: at large n give the worker contexts' arenas (concurrent sweep
        # units) back to the device first
        a = t[3]
        n = env.chi * a.shape[1] ** 2
        need = n * n * (a.shape[0] ** 4 + 2 * a.shape[0] ** 2 + 4) * a.element_size()
        total = torch.cuda.get_device_properties(a.device).total_memory
        if need > 0.15 * total:
            eng.trim(workers_only=True)
            # ... and the enlarged corners cached with the environment (rebuilt by the next move) when both would not fit
            cache = env.__dict__.get("_corner_cache")
            if cache and need + sum(e[2] for e in cache.values()) * 8 > 0.7 * total:
                env.__dict__.pop("_corner_cache", None)
    raw = _rdm2x2_raw(eng, t, env)
    if open_sites != [0, 1, 2, 3]:
        # fewer open sites = partial trace of the full plaquette RDM over the closed ones (rdm.py:1306-1360 contracts
        # their physical legs inside the corners; same numbers, the native kernel always opens all four)
        ket, bra = list("abcd"), list("efgh")
        for i in range(4):
            if i not in open_sites:
                bra[i] = ket[i]
        keep = [ket[i] for i in open_sites] + [bra[i] for i in open_sites]
        raw = torch.einsum("".join(ket + bra) + "->" + "".join(keep), raw).contiguous()
    return _sym_pos_def_rdm(raw, sym_pos_def=sym_pos_def, verbosity=verbosity, who="rdm2x2")


def _rdm2x2_raw(eng, t, env, group=None):
    """Raw plaquette RDM.  The contraction is a sum over p^4 independent lower-half slices (ctm_rdm2x2_part): they are
      * looped over in chunks on one GPU when the open halves (n^2 (p^4 + 2 p^2 + 1) elements) do not fit in free HBM
        (D = 8, chi = 384, complex128: 241 GB at once, 126 GB in chunks of four slices), and
      * shared among the ranks of `group` (parallel.site_group: the GPUs assigned to this site when there are more ranks than
        sites) with one all-reduce of the p^8 numbers."""
    import parallel
    a = t[3]
    p = a.shape[0]
    members = group if group is not None else [parallel.world()[0]]
    me = members.index(parallel.world()[0])
    if not hasattr(eng, "rdm2x2_part"):
        return eng.rdm2x2(t)
    n = env.chi * a.shape[1] ** 2
    el = a.element_size()
    P4 = p ** 4
    chunk = P4
    if a.is_cuda:
        free, _ = torch.cuda.mem_get_info(a.device)
        # the bytes this engine's own arena holds right now are reused by the call (NOT the high-water marks of the worker contexts,
        # whose arenas were given back before an evaluation of this size)
        arena = eng.own_stat("arena_total") if hasattr(eng, "own_stat") else 0
        budget = 0.85 * (free + arena)
        while chunk > 1 and 1.15 * n * n * (2 * p * p + 1 + chunk) * el > budget:
            chunk //= 2
    if len(members) > 1:
        # the members of a group see different amounts of free HBM: the partition of the p^4 slices must not depend on the rank
        chunk = parallel.allreduce_min_int_group(chunk, members, a.device)
    nparts = -(-P4 // chunk)
    if nparts == 1 and len(members) == 1:
        return eng.rdm2x2(t)
    if len(members) > 1:                                       # at least one part per member
        while nparts < len(members) and chunk > 1:
            chunk //= 2; nparts = -(-P4 // chunk)
    R = torch.zeros(P4, P4, dtype=a.dtype, device=a.device)
    ranges = [(i * chunk, min(P4, (i + 1) * chunk)) for i in range(nparts)]
    assert ranges[0][0] == 0 and ranges[-1][1] == P4 and all(a_[1] == b_[0] for a_, b_ in zip(ranges, ranges[1:])), ranges
    for i, (lo0, lo1) in enumerate(ranges):
        if i % len(members) != me:
            continue
        R[:, lo0:lo1] = eng.rdm2x2_part(t, lo0, lo1)
    if len(members) > 1:
        R = parallel.allreduce_sum_group(R, members)
    return eng.rdm2x2_from_parts(R, p)


rdm2x2_legacy = lambda coord, state, env, sym_pos_def=False, verbosity=0: rdm2x2(coord, state, env, sym_pos_def=sym_pos_def)


def rdm1x1(coord, state, env, operator=None, sym_pos_def=False, force_cpu=False, verbosity=0, mode='sl', **kwargs):
    c = state.vertexToSite(coord)
    t = (env.C[(c, (-1, -1))], env.C[(c, (1, -1))], env.C[(c, (1, 1))], env.C[(c, (-1, 1))],
         env.T[(c, (0, -1))], env.T[(c, (1, 0))], env.T[(c, (0, 1))], env.T[(c, (-1, 0))], state.site(coord))
    from ctm.generic import ctm_ad
    raw = ctm_ad.rdm1x1(coord, state, env) if ctm_ad.wants_grad(state, env) else get_engine().rdm1x1(t)
    rdm = _sym_pos_def_rdm(raw, sym_pos_def=sym_pos_def, verbosity=verbosity, who="rdm1x1")
    if operator is not None:
        return torch.einsum('ij,ji', rdm, operator.to(rdm.device))
    return rdm


def rdm2x1(coord, state, env, sym_pos_def=False, force_cpu=False, verbosity=0, mode='sl', **kwargs):
    """Horizontal pair coord, coord+(1,0); index order s0 s1 ; s0' s1'."""
    x, y = coord
    c0, c1 = state.vertexToSite((x, y)), state.vertexToSite((x + 1, y))
    t = (env.C[(c0, (-1, -1))], env.T[(c0, (0, -1))], env.T[(c0, (-1, 0))], env.C[(c0, (-1, 1))], env.T[(c0, (0, 1))], state.site((x, y)),
         env.C[(c1, (1, -1))], env.T[(c1, (1, 0))], env.C[(c1, (1, 1))], env.T[(c1, (0, -1))], env.T[(c1, (0, 1))], state.site((x + 1, y)))
    from ctm.generic import ctm_ad
    raw = ctm_ad.rdm2x1(coord, state, env) if ctm_ad.wants_grad(state, env) else get_engine().rdm2x1(t)
    return _sym_pos_def_rdm(raw, sym_pos_def=sym_pos_def, verbosity=verbosity, who="rdm2x1")


def rdm1x2(coord, state, env, sym_pos_def=False, force_cpu=False, verbosity=0, mode='sl', **kwargs):
    """Vertical pair coord, coord+(0,1); index order s0 s1 ; s0' s1'."""
    x, y = coord
    c0, c1 = state.vertexToSite((x, y)), state.vertexToSite((x, y + 1))
    t = (env.C[(c0, (-1, -1))], env.T[(c0, (0, -1))], env.C[(c0, (1, -1))], env.T[(c0, (-1, 0))], env.T[(c0, (1, 0))], state.site((x, y)),
         env.C[(c1, (-1, 1))], env.T[(c1, (0, 1))], env.C[(c1, (1, 1))], env.T[(c1, (-1, 0))], env.T[(c1, (1, 0))], state.site((x, y + 1)))
    from ctm.generic import ctm_ad
    raw = ctm_ad.rdm1x2(coord, state, env) if ctm_ad.wants_grad(state, env) else get_engine().rdm1x2(t)
    return _sym_pos_def_rdm(raw, sym_pos_def=sym_pos_def, verbosity=verbosity, who="rdm1x2")
