"""CTM environment container, initialisation and convergence check (reference ctm/generic/env.py).

ENV holds, for every site of the unit cell, four corners C[(coord,(+-1,+-1))] (chi x chi) and four
half-row/column tensors T[(coord,(0,-1))] (chi,D^2,chi), (-1,0) (chi,chi,D^2), (0,1) (D^2,chi,chi),
(1,0) (chi,D^2,chi) -- float64 device tensors living in HBM for the whole run.
"""
import logging
from math import sqrt
import torch
import config as cfg
from backend import get_engine

log = logging.getLogger(__name__)


class EnvError(Exception):
    pass


class ENV():
    def __init__(self, chi, state=None, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
        if state:
            self.dtype, self.device = state.dtype, state.device
        else:
            self.dtype, self.device = global_args.torch_dtype, global_args.device
        self.chi = chi
        self.C = dict()
        self.T = dict()
        if state is not None:
            o = dict(dtype=self.dtype, device=self.device)
            numl = 2 if len(next(iter(state.sites.values())).size()) > 4 else 1
            for coord, site in state.sites.items():
                self.T[(coord, (0, -1))] = torch.empty((chi, site.size(-4) ** numl, chi), **o)
                self.T[(coord, (-1, 0))] = torch.empty((chi, chi, site.size(-3) ** numl), **o)
                self.T[(coord, (0, 1))] = torch.empty((site.size(-2) ** numl, chi, chi), **o)
                self.T[(coord, (1, 0))] = torch.empty((chi, site.size(-1) ** numl, chi), **o)
                for vec in [(-1, -1), (-1, 1), (1, -1), (1, 1)]:
                    self.C[(coord, vec)] = torch.empty((chi, chi), **o)

    def __deepcopy__(self, memo):
        # engine-side caches attached by the move (warm-start bases, enlarged corners: tens of GB at large chi) are not part
        # of the environment's value
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in ("_warm", "_corner_cache", "_warm_ad"):
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def drop_caches(self):
        """Release the warm-start bases and cached enlarged corners kept with this environment (they are rebuilt on demand)."""
        self.__dict__.pop("_warm", None)
        self.__dict__.pop("_corner_cache", None)

    def __str__(self):
        s = f"ENV chi={self.chi}\n"
        for cr, t in self.C.items(): s += f"C({cr[0]} {cr[1]}): {t.size()}\n"
        for cr, t in self.T.items(): s += f"T({cr[0]} {cr[1]}): {t.size()}\n"
        return s

    def _like(self, f, ctm_args, global_args):
        e = ENV(self.chi, ctm_args=ctm_args, global_args=global_args)
        e.dtype, e.device = self.dtype, self.device
        e.C = {k: f(c) for k, c in self.C.items()}
        e.T = {k: f(t) for k, t in self.T.items()}
        if "_warm_ad" in self.__dict__:          # solver workspaces of the differentiable route travel with the environment
            e.__dict__["_warm_ad"] = self.__dict__["_warm_ad"]
        return e

    def clone(self, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
        return self._like(lambda t: t.clone(), ctm_args, global_args)

    def detach(self, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
        return self._like(lambda t: t.detach(), ctm_args, global_args)

    def detach_(self):
        for c in self.C.values(): c.detach_()
        for t in self.T.values(): t.detach_()

    def min_chi(self):
        return min([c.size(0) for c in self.C.values()] + [c.size(1) for c in self.C.values()])

    def extend(self, new_chi, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
        """Zero-padded copy with bond dimension new_chi (env.py:164-202)."""
        e = ENV(new_chi, ctm_args=ctm_args, global_args=global_args)
        e.dtype, e.device = self.dtype, self.device
        o = dict(dtype=self.dtype, device=self.device)
        x = min(self.chi, new_chi)
        for k, c in self.C.items():
            e.C[k] = torch.zeros(new_chi, new_chi, **o); e.C[k][:x, :x] = c[:x, :x]
        for k, t in self.T.items():
            if k[1] in [(0, -1), (1, 0)]:
                e.T[k] = torch.zeros((new_chi, t.size(1), new_chi), **o); e.T[k][:x, :, :x] = t[:x, :, :x]
            elif k[1] == (-1, 0):
                e.T[k] = torch.zeros((new_chi, new_chi, t.size(2)), **o); e.T[k][:x, :x, :] = t[:x, :x, :]
            elif k[1] == (0, 1):
                e.T[k] = torch.zeros((t.size(0), new_chi, new_chi), **o); e.T[k][:, :x, :x] = t[:, :x, :x]
            else:
                raise Exception(f"Unexpected direction {k[1]}")
        return e

    def get_spectra(self):
        """Normalised singular values of every corner (env.py:204-209), native Jacobi svdvals."""
        eng = get_engine()
        spec = {}
        for k, c in self.C.items():
            s = eng.svdvals(c)
            spec[k] = s / s[0]
        return spec

    def get_site_env_t(self, coord, state):
        c = state.vertexToSite(coord)
        return (self.C[(c, (-1, -1))], self.C[(c, (1, -1))], self.C[(c, (1, 1))], self.C[(c, (-1, 1))],
                self.T[(c, (0, -1))], self.T[(c, (1, 0))], self.T[(c, (0, 1))], self.T[(c, (-1, 0))])


def init_env(state, env, ctm_args=cfg.ctm_args):
    if len(next(iter(state.sites.values())).size()) == 4 and ctm_args.ctm_env_init_type not in ["PROD", "CTMRG_OBC", "RANDOM"]:
        raise RuntimeError("Incompatible ENV initialization")
    if ctm_args.ctm_env_init_type == 'PROD':
        init_prod(state, env, ctm_args.verbosity_initialization)
    elif ctm_args.ctm_env_init_type == 'RANDOM':
        init_random(env, ctm_args.verbosity_initialization)
    elif ctm_args.ctm_env_init_type == 'CTMRG':
        init_from_ipeps_pbc(state, env, ctm_args.verbosity_initialization)
    elif ctm_args.ctm_env_init_type == 'CTMRG_OBC':
        init_from_ipeps_obc(state, env, ctm_args.verbosity_initialization)
    else:
        raise ValueError("Invalid environment initialization: " + str(ctm_args.ctm_env_init_type))


def init_random(env, verbosity=0):
    for key, t in env.C.items():
        env.C[key] = torch.rand(t.size(), dtype=env.dtype, device=env.device)
    for key, t in env.T.items():
        env.T[key] = torch.rand(t.size(), dtype=env.dtype, device=env.device)


# per T direction: (legs of a[s,u,l,d,r] that stay open in the PROD init / leg traced between the layers, shape builder)
_PROD_T = {(0, -1): ('miefg,miebg->fb', lambda chi, n: (chi, n, chi), lambda T, a: T.__setitem__((0, slice(None), 0), a), False),
           (-1, 0): ('meifg,meifc->gc', lambda chi, n: (chi, chi, n), lambda T, a: T.__setitem__((0, 0, slice(None)), a), True),
           (0, 1): ('mefig,mafig->ea', lambda chi, n: (n, chi, chi), lambda T, a: T.__setitem__((slice(None), 0, 0), a), True),
           (1, 0): ('mefgi,mebgi->fb', lambda chi, n: (chi, n, chi), lambda T, a: T.__setitem__((0, slice(None), 0), a), True)}


def init_prod(state, env, verbosity=0):
    """env.py:274-365 (5-leg sites): C = e_00; every T holds, in its chi x chi corner (0,0), the double-layer partial trace of the
    neighbouring site over everything but the leg pointing at `coord`.  As in the reference the T of direction (0,-1) is the only
    one NOT divided by its max-abs (:297-298 vs :319, :341, :363)."""
    eng = get_engine()
    chi = env.chi
    o = dict(dtype=env.dtype, device=env.device)
    for key, t in env.C.items():
        C = torch.zeros(t.size(), **o)
        C[0, 0] = 1.0
        env.C[key] = C
    for coord in state.sites.keys():
        for vec, (expr, shape, put, normalise) in _PROD_T.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            if A.dim() != 5:
                raise NotImplementedError("init_prod: double-layer (4-leg) sites are not supported by the native engine")
            a = eng.einsum(expr, A, A, conj=(1,)).reshape(-1)
            if normalise:
                a = a / a.abs().max()
            T = torch.zeros(shape(chi, a.numel()), **o)
            put(T, a)
            env.T[(coord, vec)] = T


# open boundary: the outward legs of EACH layer are summed separately (contracted with vectors of ones), no conjugation (:538-716)
_OBC_C = {(-1, -1): ((1, 2), 3, 4), (1, -1): ((1, 4), 2, 3), (1, 1): ((3, 4), 1, 2), (-1, 1): ((2, 3), 1, 4)}
_OBC_T = {(0, -1): (1, (2, 3, 4)), (-1, 0): (2, (1, 3, 4)), (0, 1): (3, (1, 2, 4)), (1, 0): (4, (1, 2, 3))}


def init_from_ipeps_obc(state, env, verbosity=0):
    """env.py:538-716 (5-leg sites): C and T from the neighbouring site with its outward legs summed out layer by layer
    (`einsum('mijef,mklab->eafb', A, A)`: both layers un-conjugated, as the reference has it), divided by the max-abs, zero padded."""
    eng = get_engine()
    chi = env.chi
    o = dict(dtype=env.dtype, device=env.device)
    for coord in state.sites.keys():
        for vec, (summed, i0, i1) in _OBC_C.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            if A.dim() != 5:
                raise NotImplementedError("init_from_ipeps_obc: double-layer (4-leg) sites are not supported by the native engine")
            B = A.sum(dim=summed).contiguous()                                  # [m, e, f]
            a = eng.einsum('mef,mab->eafb', B, B).reshape(A.size(i0) ** 2, A.size(i1) ** 2)
            a = a / a.abs().max()
            C = torch.zeros(chi, chi, **o)
            m0, m1 = min(chi, a.size(0)), min(chi, a.size(1))
            C[:m0, :m1] = a[:m0, :m1]
            env.C[(coord, vec)] = C
        for vec, (summed, (i0, i1, i2)) in _OBC_T.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            B = A.sum(dim=summed).contiguous()                                  # [m, e, f, g]
            a = eng.einsum('mefg,mabc->eafbgc', B, B).reshape(A.size(i0) ** 2, A.size(i1) ** 2, A.size(i2) ** 2)
            a = a / a.abs().max()
            if vec in ((0, -1), (1, 0)):
                T = torch.zeros((chi, a.size(1), chi), **o); m0, m2 = min(chi, a.size(0)), min(chi, a.size(2))
                T[:m0, :, :m2] = a[:m0, :, :m2]
            elif vec == (-1, 0):
                T = torch.zeros((chi, chi, a.size(2)), **o); m0, m1 = min(chi, a.size(0)), min(chi, a.size(1))
                T[:m0, :m1, :] = a[:m0, :m1, :]
            else:
                T = torch.zeros((a.size(0), chi, chi), **o); m1, m2 = min(chi, a.size(1)), min(chi, a.size(2))
                T[:, :m1, :m2] = a[:, :m1, :m2]
            env.T[(coord, vec)] = T


_C_KIND = {(-1, -1): (0, 3, 4), (1, -1): (1, 2, 3), (1, 1): (2, 1, 2), (-1, 1): (3, 1, 4)}
_T_KIND = {(0, -1): 4, (-1, 0): 5, (0, 1): 6, (1, 0): 7}


_PIECE_EXPR = ['mijef,mijab->eafb', 'miefj,miabj->eafb', 'mefij,mabij->eafb', 'meijf,maijb->eafb',
               'miefg,miabc->eafbgc', 'meifg,maibc->eafbgc', 'mefig,mabic->eafbgc', 'mefgi,mabci->eafbgc']


def _init_piece(eng, kind, A):
    """Double-layer partial trace number `kind` of the site A, fused leg pairs, divided by its max-abs.  A site that is being
    optimised gets it as a differentiable native contraction node, the scale taken without gradient (env.py:369-372)."""
    from linalg.native_einsum import needs_grad, einsum
    if not needs_grad(A):
        return eng.init_piece(kind, A)
    r = einsum(_PIECE_EXPR[kind], A, A, conj=(1,))
    sh = r.shape
    r = r.reshape([sh[2 * i] * sh[2 * i + 1] for i in range(len(sh) // 2)])
    return r / r.detach().abs().max()


def init_from_ipeps_pbc(state, env, verbosity=0):
    """env.py:367-536: each env tensor of `coord` = normalised double-layer partial trace of the
    neighbouring site in direction vec (native kernel), zero padded to chi."""
    eng = get_engine()
    chi = env.chi
    o = dict(dtype=env.dtype, device=env.device)
    for coord in state.sites.keys():
        for vec, (kind, i0, i1) in _C_KIND.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            a = _init_piece(eng, kind, A)
            C = torch.zeros(chi, chi, **o)
            m0, m1 = min(chi, A.size(i0) ** 2), min(chi, A.size(i1) ** 2)
            C[:m0, :m1] = a[:m0, :m1]
            env.C[(coord, vec)] = C
        for vec, kind in _T_KIND.items():
            A = state.site((coord[0] + vec[0], coord[1] + vec[1]))
            d = A.size()
            a = _init_piece(eng, kind, A)
            if vec == (0, -1):
                T = torch.zeros((chi, d[3] ** 2, chi), **o); m0, m2 = min(chi, d[2] ** 2), min(chi, d[4] ** 2)
                T[:m0, :, :m2] = a[:m0, :, :m2]
            elif vec == (-1, 0):
                T = torch.zeros((chi, chi, d[4] ** 2), **o); m0, m1 = min(chi, d[1] ** 2), min(chi, d[3] ** 2)
                T[:m0, :m1, :] = a[:m0, :m1, :]
            elif vec == (0, 1):
                T = torch.zeros((d[1] ** 2, chi, chi), **o); m1, m2 = min(chi, d[2] ** 2), min(chi, d[4] ** 2)
                T[:, :m1, :m2] = a[:, :m1, :m2]
            else:
                T = torch.zeros((chi, d[2] ** 2, chi), **o); m0, m2 = min(chi, d[1] ** 2), min(chi, d[3] ** 2)
                T[:m0, :, :m2] = a[:m0, :, :m2]
            env.T[(coord, vec)] = T


@torch.no_grad()
def ctmrg_conv_specC(state, env, history, p='inf', ctm_args=cfg.ctm_args):
    """Convergence on corner spectra (env.py:816-875): sqrt of the max (or sum) over corners of
    sum (s_i - s_i_prev)^2 below ctm_conv_tol."""
    def _diff(s1, s2):
        n = min(s1.size(0), s2.size(0))
        return (sum((s1[:n] - s2[:n]) ** 2) + sum(s1[n:] ** 2) + sum(s2[n:] ** 2)).item()
    if not history:
        history = {'spec': [], 'diffs': [], 'conv_crit': []}
    conv_crit, diffs = float('inf'), None
    spec = {k: v.sort(descending=True)[0].cpu() for k, v in env.get_spectra().items()}
    if len(history['spec']) > 0:
        old = history['spec'][-1]
        diffs = [_diff(spec[k], old[k]) for k in spec.keys()]
        conv_crit = sqrt(sum(diffs)) if p in ['fro', 2] else sqrt(max(diffs))
    history['spec'].append(spec); history['diffs'].append(diffs); history['conv_crit'].append(conv_crit)
    if (len(history['diffs']) > 1 and conv_crit < ctm_args.ctm_conv_tol) or len(history['diffs']) >= ctm_args.ctm_max_iter:
        log.info({"history_length": len(history['diffs']), "history": history['diffs']})
        return True, history
    return False, history
