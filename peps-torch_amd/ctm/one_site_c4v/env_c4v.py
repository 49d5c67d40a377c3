"""C4v-symmetric single-site environment (reference ctm/one_site_c4v/env_c4v.py:7-154,166-311):
one corner C (chi x chi, diagonal after every move) and one half-row tensor T (chi, chi, D^2)."""
import torch
import config as cfg
from backend import get_engine
from linalg.custom_eig import truncated_eig_sym


class ENV_C4V():
    def __init__(self, chi, state=None, bond_dim=None, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
        assert state or bond_dim, "either state or bond_dim must be supplied"
        self.dtype, self.device = global_args.torch_dtype, global_args.device
        if state:
            assert len(state.sites) == 1, "Not a 1-site ipeps"
            site = next(iter(state.sites.values()))
            assert site.size(-4) == site.size(-3) == site.size(-2) == site.size(-1), \
                "bond dimensions of on-site tensor are not equal"
            bond_dim = site.size(-1)
            self.dtype, self.device = site.dtype, site.device
        self.chi = chi
        self.bond_dim = bond_dim
        self.keyC = ((0, 0), (-1, -1))
        self.keyT = ((0, 0), (-1, 0))
        self.C = {self.keyC: torch.zeros((chi, chi), dtype=self.dtype, device=self.device)}
        self.T = {self.keyT: torch.zeros((chi, chi, bond_dim ** 2), dtype=self.dtype, device=self.device)}

    def __str__(self):
        return f"ENV_C4V chi={self.chi}\nC {self.get_C().size()}\nT {self.get_T().size()}\n"

    def get_C(self): return self.C[self.keyC]
    def get_T(self): return self.T[self.keyT]

    def _like(self, f, ctm_args, global_args):
        e = ENV_C4V(self.chi, bond_dim=self.bond_dim, ctm_args=ctm_args, global_args=global_args)
        e.dtype, e.device = self.dtype, self.device
        e.C[e.keyC] = f(self.get_C()); e.T[e.keyT] = f(self.get_T())
        if "_warm_ad" in self.__dict__:          # solver workspaces of the differentiable route travel with the environment
            e.__dict__["_warm_ad"] = self.__dict__["_warm_ad"]
        return e

    def clone(self, ctm_args=cfg.ctm_args, global_args=cfg.global_args): return self._like(lambda t: t.clone(), ctm_args, global_args)
    def detach(self, ctm_args=cfg.ctm_args, global_args=cfg.global_args): return self._like(lambda t: t.detach(), ctm_args, global_args)

    def detach_(self):
        self.get_C().detach_(); self.get_T().detach_()

    def extend(self, new_chi, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
        e = ENV_C4V(new_chi, bond_dim=self.bond_dim, ctm_args=ctm_args, global_args=global_args)
        e.dtype, e.device = self.dtype, self.device
        e.C[e.keyC] = torch.zeros((new_chi, new_chi), dtype=self.dtype, device=self.device)
        e.T[e.keyT] = torch.zeros((new_chi, new_chi, self.bond_dim ** 2), dtype=self.dtype, device=self.device)
        x = min(self.chi, new_chi)
        e.C[e.keyC][:x, :x] = self.get_C()[:x, :x]
        e.T[e.keyT][:x, :x, :] = self.get_T()[:x, :x, :]
        return e

    def get_spectra(self):
        d = torch.abs(torch.diagonal(self.get_C()))
        d, _ = torch.sort(d, descending=True)
        return {self.keyC: d / d[0]}


def init_env(state, env, C_and_T=None, ctm_args=cfg.ctm_args):
    if C_and_T:
        x = C_and_T[0].size(0)
        env.C[env.keyC][:x, :x] = C_and_T[0]
        env.T[env.keyT] = torch.zeros((env.chi, env.chi, C_and_T[1].size(2)), dtype=env.dtype, device=env.device)
        env.T[env.keyT][:x, :x, :] = C_and_T[1]
        return
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
    c = torch.rand(env.get_C().size(), dtype=env.dtype, device=env.device)
    env.C[env.keyC] = 0.5 * (c + c.conj().t())
    env.T[env.keyT] = torch.rand(env.get_T().size(), dtype=env.dtype, device=env.device)


def init_prod(state, env, verbosity=0):
    """env_c4v.py:215-246 (5-leg site): C = e_00; T[0,0,:] = the leading eigenvector of the double-layer partial trace over the
    physical, up and down legs ('meifj,maibj->eafb'), which must be unique."""
    eng = get_engine()
    A = state.site()
    if A.dim() != 5:
        raise NotImplementedError("init_prod: double-layer (4-leg) sites are not supported by the native engine")
    C = torch.zeros(env.chi, env.chi, dtype=env.dtype, device=env.device)
    C[0, 0] = 1.0
    env.C[env.keyC] = C
    a = eng.einsum('meifj,maibj->eafb', A, A, conj=(1,)).reshape(A.size(1) ** 2, A.size(3) ** 2)
    a = a / a.abs().max()
    assert torch.norm(a.conj().t() - a) / a.abs().max() < 1.0e-8, "a is not symmetric"
    Dv, U = truncated_eig_sym(a, 2)
    assert torch.abs(Dv[0] - Dv[1]) > 1.0e-8, "Leading eigenvector of T not unique"
    T = torch.zeros((env.chi, env.chi, A.size(3) ** 2), dtype=env.dtype, device=env.device)
    T[0, 0, :] = U[:, 0]
    env.T[env.keyT] = T


def init_from_ipeps_obc(state, env, verbosity=0):
    """env_c4v.py:315-355: C and T from the site with its outward legs summed out in each layer separately (ket and conjugated bra:
    `einsum('mijef,mklab->eafb', A, A.conj())`), divided by the max-abs, zero padded to chi."""
    eng = get_engine()
    A = state.site()
    if A.dim() != 5:
        raise RuntimeError("Incompatible ENV_C4V initialization")
    d = A.size()
    o = dict(dtype=env.dtype, device=env.device)
    B = A.sum(dim=(1, 2)).contiguous()                                    # [m, e, f]
    a = eng.einsum('mef,mab->eafb', B, B, conj=(1,)).reshape(d[3] ** 2, d[4] ** 2)
    a = a / a.abs().max()
    C = torch.zeros(env.chi, env.chi, **o)
    m0, m1 = min(env.chi, d[3] ** 2), min(env.chi, d[4] ** 2)
    C[:m0, :m1] = a[:m0, :m1]
    env.C[env.keyC] = C
    B = A.sum(dim=2).contiguous()                                         # [m, e, f, g]
    t = eng.einsum('mefg,mabc->eafbgc', B, B, conj=(1,)).reshape(d[1] ** 2, d[3] ** 2, d[4] ** 2)
    t = t / t.abs().max()
    T = torch.zeros((env.chi, env.chi, d[4] ** 2), **o)
    m0, m1 = min(env.chi, d[1] ** 2), min(env.chi, d[3] ** 2)
    T[:m0, :m1, :] = t[:m0, :m1, :]
    env.T[env.keyT] = T


def init_from_ipeps_pbc(state, env, verbosity=0):
    """env_c4v.py:262-311: corner = eig-decomposed double-layer partial trace, C = diag(D); T rotated
    into the eigenbasis, T_ijs = sum_ab U_ai t_abs conj(U_bj)."""
    eng = get_engine()
    a = state.site()
    D2 = a.size(1) ** 2
    from linalg.native_einsum import needs_grad, einsum
    if needs_grad(a):
        # the site is being optimised: the same construction as a graph of differentiable native nodes (the reference
        # differentiates through the initial environment too, optim_j1j2_c4v.py:104-106); scales without gradient (:277-279,303-305)
        c = einsum('mijef,mijab->eafb', a, a, conj=(1,)).reshape(D2, D2)
        c = c / c.detach().abs().max()
        assert torch.norm(c.detach().conj().t() - c.detach()) / c.detach().abs().max() < 1.0e-8, "a is not symmetric"
        Dv, U = truncated_eig_sym(c, D2)
        m = min(env.chi, D2)
        C = torch.zeros(env.chi, env.chi, dtype=env.dtype, device=env.device)
        C[:m, :m] = torch.diag(Dv.to(env.dtype))[:m, :m]
        env.C[env.keyC] = C
        t = einsum('meifg,maibc->eafbgc', a, a, conj=(1,)).reshape(D2, D2, D2)
        t = t / t.detach().abs().max()
        t2 = einsum('ai,abs,bj->ijs', U.contiguous(), t, U.contiguous(), conj=(2,))
        T = torch.zeros((env.chi, env.chi, D2), dtype=env.dtype, device=env.device)
        T[:m, :m, :] = t2[:m, :m, :]
        env.T[env.keyT] = T
        return
    c = eng.init_piece(0, a)                                   # 'mijef,mijab->eafb', /max-abs
    asym = torch.norm(c.conj().t() - c) / c.abs().max()
    assert asym < 1.0e-8, "a is not symmetric"
    Dv, U = truncated_eig_sym(c, c.size(0))
    m = min(env.chi, D2)
    C = torch.zeros(env.chi, env.chi, dtype=env.dtype, device=env.device)
    C[:m, :m] = torch.diag(Dv)[:m, :m]
    env.C[env.keyC] = C
    t = eng.init_piece(5, a)                                   # 'meifg,maibc->eafbgc' -> (D^2, D^2, D^2), /max-abs
    # 'ai,abs,bj->ijs' with conj(U) on the bra leg, as two GEMMs (U^T, then U^H: trans = 2 is the conjugate transpose)
    t1 = eng.gemm(U, t.reshape(D2, D2 * D2), transA=True).reshape(D2, D2, D2)         # [i, b, s]
    t2 = eng.gemm(U, eng.permute(t1, (1, 0, 2)).reshape(D2, D2 * D2), transA=(2 if U.is_complex() else True))   # [j, (i s)]
    t2 = eng.permute(t2.reshape(D2, D2, D2), (1, 0, 2))                                 # [i, j, s]
    asym = (t2 - t2.permute(1, 0, 2).conj()).norm() / t2.abs().max()
    assert asym < 1.0e-8, "a is not symmetric"
    T = torch.zeros((env.chi, env.chi, D2), dtype=env.dtype, device=env.device)
    T[:m, :m, :] = t2[:m, :m, :]
    env.T[env.keyT] = T


def compute_multiplets(env, eps_multiplet_gap=1.0e-10):
    """Sizes of the groups of (numerically) degenerate |eigenvalues| of the corner matrix, in descending order
    (reference env_c4v.py:401-417): a group ends where the gap to the next value exceeds eps_multiplet_gap."""
    eng = get_engine()
    C = env.C[env.keyC]
    D, _ = eng.truncated_eigh(C, env.chi, eng.cfg(keep_multiplets=False))     # ordered by |D| descending
    d = torch.cat([D.abs(), torch.zeros(1, dtype=D.dtype, device=D.device)]).cpu()
    m, l = [], 0
    for i in range(env.chi):
        l += 1
        if d[i] - d[i + 1] > eps_multiplet_gap:
            m.append(l)
            l = 0
    return m
