"""J1-J2(-h) Heisenberg model on the square lattice: plaquette Hamiltonian and the energy / observable
evaluation that consumes the native RDMs (reference models/j1j2.py:60-247,427-474,591-679).

Hamiltonian tensors are tiny (<= 2^8 numbers) and live on the host (CPU); RDMs come back from the
engine and are contracted with them on the host."""
import itertools
from math import sqrt
import torch
import groups.su2 as su2
import config as cfg
from ctm.generic import rdm
from ctm.one_site_c4v import rdm_c4v
import parallel


def eval_nnnn_per_site(coord, state, env, obs_ops):
    """Third-neighbour S.S along x and y from the distance-2 two-point functions (reference models/j1j2.py:27-44)."""
    from ctm.generic import corrf
    f = lambda d, o1, o2: corrf.corrf_1sO1sO(coord, d, state, env, obs_ops[o1], (lambda r: obs_ops[o2]), 2)[1]
    return (f((1, 0), "sz", "sz") + f((0, 1), "sz", "sz")
            + 0.5 * (f((1, 0), "sp", "sm") + f((0, 1), "sp", "sm") + f((1, 0), "sm", "sp") + f((0, 1), "sm", "sp")))


def _cast_to_real(t):
    return t.real if t.is_complex() else t


def _cyclic_shift_4(m, dtype):
    """P4[s0' s1' s2' s3'; s0 s1 s2 s3] = swap(2,3) swap(1,2) swap(0,1) as an operator on four spins (out; in)."""
    def swap(i, j):
        S = torch.zeros([m] * 8, dtype=dtype)
        for idx in itertools.product(range(m), repeat=4):
            o = list(idx); o[i], o[j] = o[j], o[i]
            S[tuple(o) + tuple(idx)] = 1
        return S.reshape(m ** 4, m ** 4)
    return (swap(2, 3) @ swap(1, 2) @ swap(0, 1)).reshape([m] * 8)


class J1J2():
    def __init__(self, j1=1.0, j2=0, j3=0, hz_stag=0.0, delta_zz=1.0, lmbd=0, h_uni=[0, 0, 0], global_args=cfg.global_args):
        self.dtype = global_args.torch_dtype
        if lmbd != 0 and not torch.zeros(1, dtype=self.dtype).is_complex():
            raise AssertionError("Invalid dtype: Lambda requires complex numbers")        # models/j1j2.py:97-98
        self.device = 'cpu'
        self.phys_dim = 2
        self.j1, self.j2, self.j3, self.lmbd, self.hz_stag, self.delta_zz = j1, j2, j3, lmbd, hz_stag, delta_zz
        self.h_uni = torch.as_tensor(h_uni, dtype=self.dtype)
        s2 = su2.SU2(self.phys_dim, dtype=self.dtype, device=self.device)
        id2, id3 = s2.I_N(N=2), s2.I_N(N=3)
        kron = 'ij,ab->iajb'
        self.SS_delta_zz = s2.SS(xyz=(delta_zz, 1., 1.))
        self.SS = s2.SS()
        h_uni_1x1 = torch.einsum('x,xia->ia', self.h_uni, s2.S())
        hz_2x1_nn = torch.einsum(kron, s2.SZ(), s2.I()) + torch.einsum(kron, s2.I(), -s2.SZ())
        huni_2x1_nn = torch.einsum(kron, h_uni_1x1, s2.I()) + torch.einsum(kron, s2.I(), h_uni_1x1)
        rot = s2.BP_rot()
        _rot2 = lambda h: torch.einsum('ki,kjcb,ca->ijab', rot, h, rot).contiguous()
        self.SS_rot, self.SS_delta_zz_rot = _rot2(self.SS), _rot2(self.SS_delta_zz)
        self.hz_2x1_rot, self.huni_2x1_rot = _rot2(hz_2x1_nn), _rot2(huni_2x1_nn)
        hSSd = torch.einsum('ijab,klcd->ijklabcd', self.SS_delta_zz, id2)
        hSS = torch.einsum('ijab,klcd->ijklabcd', self.SS, id2)
        hz = torch.einsum('ia,jklbcd->ijklabcd', s2.SZ(), id3)
        hu = torch.einsum('ia,jklbcd->ijklabcd', h_uni_1x1, id3)

        def get_hp(coord):
            # all terms inside one plaquette s0 s1 / s2 s3 so that E/site = <h_p>
            hp = 0.5 * self.j1 * (hSSd + hSSd.permute(0, 2, 1, 3, 4, 6, 5, 7) + hSSd.permute(2, 3, 0, 1, 6, 7, 4, 5)
                                  + hSSd.permute(3, 1, 2, 0, 7, 5, 6, 4)) \
                + self.j2 * (hSS.permute(0, 3, 2, 1, 4, 7, 6, 5) + hSS.permute(2, 1, 0, 3, 6, 5, 4, 7)) \
                - 0.25 * self.hz_stag * ((-1) ** (coord[0] + coord[1])) * (hz - hz.permute(3, 0, 1, 2, 7, 4, 5, 6)
                                                                            - hz.permute(2, 3, 0, 1, 6, 7, 4, 5) + hz.permute(1, 2, 3, 0, 5, 6, 7, 4)) \
                + 0.25 * (hu + hu.permute(2, 3, 0, 1, 6, 7, 4, 5) + hu.permute(3, 0, 1, 2, 7, 4, 5, 6) + hu.permute(1, 2, 3, 0, 5, 6, 7, 4))
            return hp
        self.get_hp = get_hp
        self.hp_rot = torch.einsum('xj,yk,ixylauvd,ub,vc->ijklabcd', rot, rot, self.get_hp((0, 0)), rot, rot).contiguous()
        # scalar-chirality term i (P4 - P4^dagger) of the plaquette (models/j1j2.py:147-175): P4 moves the spin of site q to
        # site q+1 around the plaquette; the RDM orders the sites s0 s1 / s2 s3, the cycle runs s0 -> s1 -> s3 -> s2
        self.chiral_term = self.chiral_term_rot = self.hp_chiral_rot = 0 * s2.I_N(N=4)
        if self.phys_dim == 2 and self.lmbd != 0:
            P4 = _cyclic_shift_4(self.phys_dim, self.dtype)
            ch = 1.0j * (P4 - P4.reshape(16, 16).t().reshape([2] * 8))
            self.chiral_term = ch.permute(0, 1, 3, 2, 4, 5, 7, 6)
            self.chiral_term_rot = torch.einsum('xj,yk,ixylauvd,ub,vc->ijklabcd', rot, rot, self.chiral_term, rot, rot).contiguous()
            self.hp_chiral_rot = self.lmbd * self.chiral_term_rot
        self.obs_ops = self.get_obs_ops()

    def get_obs_ops(self):
        s2 = su2.SU2(self.phys_dim, dtype=self.dtype, device=self.device)
        return {"sz": s2.SZ(), "sp": s2.SP(), "sm": s2.SM()}

    def energy_per_site(self, state, env):
        """Mean over the unit cell of tr(rho_2x2(coord) h_p(coord)) (models/j1j2.py:223-247).  The per-site
        RDMs are independent: with torch.distributed each rank evaluates its sites and the partial sums
        are all-reduced."""
        coords = list(state.sites.keys())
        from ctm.generic import ctm_ad
        if ctm_ad.wants_grad(state, env):
            # differentiable route: the value stays a graph (no host floats), plaquettes one after the other
            e = 0.
            for coord in coords:
                r = rdm.rdm2x2(coord, state, env).cpu()
                e = e + _cast_to_real(torch.einsum('ijklabcd,ijklabcd', r, self.get_hp(coord).to(r.dtype)))
                if abs(self.lmbd) > 0:    # chiral plaquette term, un-rotated as in the reference (models/j1j2.py:240-241)
                    e = e + _cast_to_real(self.lmbd * torch.einsum('ijklabcd,ijklabcd', r, self.chiral_term.to(r.dtype)))
                if abs(self.j3) > 0:      # evaluated at (0,0) for every site of the cell (models/j1j2.py:243-244): transfer-matrix correlators as a graph
                    e = e + _cast_to_real(self.j3 * eval_nnnn_per_site((0, 0), state, env, self.obs_ops)).cpu()
            return e / len(coords)
        groups = parallel.site_groups(len(coords))
        if any(len(g) > 1 for g in groups):
            return self._energy_per_site_grouped(state, env, coords, groups)
        mine = parallel.my_units(coords)
        # the plaquette RDMs of my sites are independent: overlap them on streams when the open halves are small enough
        pool = None
        if len(mine) > 1 and getattr(cfg.ctm_args, "concurrent_units", True):
            import units
            from backend import get_engine
            a = state.site(mine[0])
            n = env.chi * a.shape[1] ** 2
            # rdm2x2 workspace: n^2 (p^4 + 2 p^2 + 4) elements, times 3 for the arena's slab-growth overshoot
            est = 3.0 * n * n * (a.shape[0] ** 4 + 2 * a.shape[0] ** 2 + 4) * a.element_size()
            pool = units.pool_for(get_engine(), len(mine), n, a.is_complex(), est_bytes=est)
        rdms = pool.map(lambda c: rdm.rdm2x2(c, state, env), mine) if pool is not None else [rdm.rdm2x2(c, state, env) for c in mine]
        e = 0.
        for coord, r in zip(mine, rdms):
            r = r.cpu()
            e += float(_cast_to_real(torch.einsum('ijklabcd,ijklabcd', r, self.get_hp(coord).to(r.dtype))))
            if abs(self.lmbd) > 0:    # models/j1j2.py:240-241
                e += float(_cast_to_real(self.lmbd * torch.einsum('ijklabcd,ijklabcd', r, self.chiral_term.to(r.dtype))))
            if abs(self.j3) > 0:      # the reference evaluates this term at (0,0) for every site of the cell (models/j1j2.py:243-244)
                e += float(_cast_to_real(self.j3 * eval_nnnn_per_site((0, 0), state, env, self.obs_ops)))
        e = parallel.allreduce_sum_scalar(e, state.device)
        return torch.as_tensor(e / len(coords), dtype=torch.float64)

    def _energy_per_site_grouped(self, state, env, coords, groups):
        """More ranks than sites (e.g. 8 GPUs on a 4-site cell): the ranks {r : r mod Nsites == i} share the plaquette RDM of
        site i (p^4 lower-half slices split among them, one all-reduce of p^8 numbers inside the group); the group's first rank
        adds tr(rho h_p) to the energy, which is then summed over all ranks."""
        from backend import get_engine
        from ctm.generic.ctm_components import _corner_t, LU, RU, RD, LD
        parallel.prepare_groups(groups)
        rank = parallel.world()[0]
        eng = get_engine()
        e = 0.
        for coord, members in zip(coords, groups):
            if rank not in members:
                continue
            x, y = coord
            t = _corner_t(LU, (x, y), state, env) + _corner_t(RU, (x + 1, y), state, env) \
                + _corner_t(RD, (x + 1, y + 1), state, env) + _corner_t(LD, (x, y + 1), state, env)
            raw = rdm._rdm2x2_raw(eng, t, env, group=members)
            if rank == members[0]:
                r = rdm._sym_pos_def_rdm(raw, who="rdm2x2").cpu()
                e += float(_cast_to_real(torch.einsum('ijklabcd,ijklabcd', r, self.get_hp(coord).to(r.dtype))))
                if abs(self.lmbd) > 0:
                    e += float(_cast_to_real(self.lmbd * torch.einsum('ijklabcd,ijklabcd', r, self.chiral_term.to(r.dtype))))
                if abs(self.j3) > 0:
                    e += float(_cast_to_real(self.j3 * eval_nnnn_per_site((0, 0), state, env, self.obs_ops)))
        e = parallel.allreduce_sum_scalar(e, state.device)
        return torch.as_tensor(e / len(coords), dtype=torch.float64)

    def energy_2x2_2site(self, state, env): return self.energy_per_site(state, env)
    def energy_2x2_4site(self, state, env): return self.energy_per_site(state, env)
    def energy_2x2_8site(self, state, env): return self.energy_per_site(state, env)

    def energy_2x2_1site_BP(self, state, env):
        assert self.h_uni[:2].norm() == 0
        r = rdm.rdm2x2((0, 0), state, env).cpu()
        e = torch.einsum('ijklabcd,ijklabcd', r, self.hp_rot.to(r.dtype))
        if abs(self.lmbd) > 0:
            e = e + torch.einsum('ijklabcd,ijklabcd', r, self.hp_chiral_rot.to(r.dtype))
        if abs(self.j3) > 0:                                                   # models/j1j2.py:216-218
            e = e + self.j3 * eval_nnnn_per_site((0, 0), state, env, self.obs_ops)
        return _cast_to_real(e)

    def _eval_obs(self, state, env, ss):
        obs = {"avg_m": 0.}
        for coord in state.sites.keys():
            r1 = rdm.rdm1x1(coord, state, env).cpu()
            for label, op in self.obs_ops.items():
                obs[f"{label}{coord}"] = torch.trace(r1 @ op.to(r1.dtype))
            obs[f"m{coord}"] = sqrt(abs(obs[f"sz{coord}"] ** 2 + obs[f"sp{coord}"] * obs[f"sm{coord}"]))
            obs["avg_m"] += obs[f"m{coord}"]
        obs["avg_m"] = obs["avg_m"] / len(state.sites.keys())
        for coord in state.sites.keys():
            r21, r12 = rdm.rdm2x1(coord, state, env).cpu(), rdm.rdm1x2(coord, state, env).cpu()
            obs[f"SS2x1{coord}"] = _cast_to_real(torch.einsum('ijab,ijab', r21, ss.to(r21.dtype)))
            obs[f"SS1x2{coord}"] = _cast_to_real(torch.einsum('ijab,ijab', r12, ss.to(r12.dtype)))
        labels = ["avg_m"] + [f"m{c}" for c in state.sites.keys()] \
            + [f"{lc[1]}{lc[0]}" for lc in itertools.product(state.sites.keys(), self.obs_ops.keys())] \
            + [f"SS2x1{c}" for c in state.sites.keys()] + [f"SS1x2{c}" for c in state.sites.keys()]
        return [obs[l] for l in labels], labels

    def _bilat(self, op, conjugate):
        """models/j1j2.py:18-25: the operator on every second site carries the sublattice rotation when `conjugate`."""
        if not conjugate:
            return lambda r: op
        rot = su2.get_rot_op(self.phys_dim, dtype=op.dtype, device=op.device)
        op_rot = torch.einsum('ki,kl,lj->ij', rot, op, rot)
        return lambda r: op_rot if r % 2 == 0 else op

    def eval_corrf_SS(self, coord, direction, state, env, dist, conjugate=False):
        """<S(r).S(0)> and its zz / xx / yy parts for r = 1 .. dist + 1 from `coord` along `direction` (models/j1j2.py:477-497)."""
        from ctm.generic import corrf
        sx = 0.5 * (self.obs_ops["sp"] + self.obs_ops["sm"])
        isy = -0.5 * (self.obs_ops["sp"] - self.obs_ops["sm"])
        zz = corrf.corrf_1sO1sO(coord, direction, state, env, self.obs_ops["sz"], self._bilat(self.obs_ops["sz"], conjugate), dist)
        xx = corrf.corrf_1sO1sO(coord, direction, state, env, sx, self._bilat(sx, conjugate), dist)
        nyy = corrf.corrf_1sO1sO(coord, direction, state, env, isy, self._bilat(isy, conjugate), dist)
        return dict({"ss": zz + xx - nyy, "szsz": zz, "sxsx": xx, "sysy": -nyy})

    def eval_corrf_SpSm(self, coord, direction, state, env, dist, conjugate=False):
        """<S^+(0) S^-(r)> and <S^-(0) S^+(r)> (models/j1j2.py:499-527)."""
        from ctm.generic import corrf
        sp, sm = self.obs_ops["sp"], self.obs_ops["sm"]
        return dict({"spsm": corrf.corrf_1sO1sO(coord, direction, state, env, sp, self._bilat(sm, conjugate), dist),
                     "smsp": corrf.corrf_1sO1sO(coord, direction, state, env, sm, self._bilat(sp, conjugate), dist)})

    def eval_obs(self, state, env): return self._eval_obs(state, env, self.SS)
    def eval_obs_1site_BP(self, state, env): return self._eval_obs(state, env, self.SS_rot)


class J1J2_C4V_BIPARTITE(J1J2):
    def energy_1x1(self, state, env_c4v, force_cpu=False, **kwargs):
        r = rdm_c4v.rdm2x2(state, env_c4v, sym_pos_def=True).cpu()
        e = torch.einsum('ijklabcd,ijklabcd', r, self.hp_rot.to(r.dtype))
        if abs(self.lmbd) > 0:                                             # models/j1j2.py:630-631
            e = e + torch.einsum('ijklabcd,ijklabcd', r, self.hp_chiral_rot.to(r.dtype))
        if abs(self.j3) > 0:                                               # :632-636
            r31 = rdm_c4v.rdm3x1(state, env_c4v, sym_pos_def=True).cpu()
            e = e + 2.0 * self.j3 * torch.einsum('ijab,ijab', r31, self.SS.to(r.dtype))
        return _cast_to_real(e)

    def energy_1x1_lowmem(self, state, env_c4v, force_cpu=False):
        nn = rdm_c4v.rdm2x2_NN_lowmem_sl(state, env_c4v, sym_pos_def=True).cpu()
        dt = nn.dtype
        e = 2.0 * self.j1 * torch.einsum('ijkl,ijkl', nn, self.SS_delta_zz_rot.to(dt)) \
            - 0.5 * self.hz_stag * torch.einsum('ijkl,ijkl', nn, self.hz_2x1_rot.to(dt))
        if abs(self.h_uni.norm()) > 0:
            e = e + 0.5 * torch.einsum('ijkl,ijkl', nn, self.huni_2x1_rot.to(dt))
        if abs(self.j2) > 0:
            nnn = rdm_c4v.rdm2x2_NNN_lowmem_sl(state, env_c4v, sym_pos_def=True).cpu()
            e = e + 2.0 * self.j2 * torch.einsum('ijkl,ijkl', nnn, self.SS.to(dt))
        if abs(self.j3) > 0:                                               # models/j1j2.py:672-676
            r31 = rdm_c4v.rdm3x1_sl(state, env_c4v, sym_pos_def=True).cpu()
            e = e + 2.0 * self.j3 * torch.einsum('ijab,ijab', r31, self.SS.to(dt))
        return _cast_to_real(e)

    def eval_corrf_SS(self, state, env_c4v, dist, canonical=False, rl_0=None):
        """<S(r).S(0)> and its zz / xx / yy parts for r = 0 .. dist along a row (models/j1j2.py:803-865): transfer-matrix correlators
        with the sublattice rotation applied to the operator on every second site; `canonical` first rotates the spin operators so
        that z points along the spontaneous magnetisation read off rho_1x1."""
        from ctm.one_site_c4v import corrf_c4v
        dt = state.site().dtype
        Sop = torch.zeros((3, self.phys_dim, self.phys_dim), dtype=dt, device="cpu")
        Sop[0] = self.obs_ops["sz"]
        Sop[1] = 0.5 * (self.obs_ops["sp"] + self.obs_ops["sm"])
        Sop[2] = -0.5 * (self.obs_ops["sp"] - self.obs_ops["sm"])
        if canonical:
            r1 = rdm_c4v.rdm1x1(state, env_c4v).cpu()
            zpm = [torch.trace(r1 @ self.obs_ops[l].to(r1.dtype)) for l in ("sz", "sp", "sm")]
            v = torch.stack([zpm[0], 0.5 * (zpm[1] + zpm[2]), 0.5 * (zpm[1] - zpm[2])]).to(dt)
            v = v / torch.norm(v)
            R = torch.zeros(3, 3, dtype=dt)
            R[0, 0], R[0, 1], R[1, 0], R[1, 1], R[2, 2] = v[0], -v[1], v[1], v[0], 1.0
            Sop = torch.einsum('ab,bij->aij', R.t(), Sop)
        rot = su2.get_rot_op(self.phys_dim, dtype=dt, device="cpu")

        def bilat(op):
            op_rot = torch.einsum('ki,kl,lj->ij', rot, op, rot)
            return lambda r: op_rot if r % 2 == 0 else op

        zz = corrf_c4v.corrf_1sO1sO(state, env_c4v, Sop[0], bilat(Sop[0]), dist, rl_0=rl_0)
        xx = corrf_c4v.corrf_1sO1sO(state, env_c4v, Sop[1], bilat(Sop[1]), dist, rl_0=rl_0)
        nyy = corrf_c4v.corrf_1sO1sO(state, env_c4v, Sop[2], bilat(Sop[2]), dist, rl_0=rl_0)
        return dict({"ss": zz + xx - nyy, "szsz": zz, "sxsx": xx, "sysy": -nyy})

    def eval_corrf_DD_H(self, state, env_c4v, dist, verbosity=0):
        """Horizontal dimer-dimer correlator <(S(r+3).S(r+2)) (S(1).S(0))>, r = 0 .. dist (models/j1j2.py:867-897)."""
        from ctm.one_site_c4v import corrf_c4v
        dt = state.site().dtype
        rot = su2.get_rot_op(self.phys_dim, dtype=dt, device="cpu")
        SS_rot = torch.einsum('ki,kjcb,ca->ijab', rot, self.SS.to(dt), rot)        # rotation on the first spin
        op_rot = SS_rot.permute(1, 0, 3, 2).contiguous()                           # ... on the second spin
        return dict({"dd": corrf_c4v.corrf_2sOH2sOH_E1(state, env_c4v, SS_rot, lambda r: SS_rot if r % 2 == 0 else op_rot, dist)})

    def eval_corrf_DD_V(self, state, env_c4v, dist, verbosity=0):
        """Vertical dimer-dimer correlator in the width-2 channel, r = 0 .. dist (models/j1j2.py:899-926)."""
        from ctm.one_site_c4v import corrf_c4v
        dt = state.site().dtype
        rot = su2.get_rot_op(self.phys_dim, dtype=dt, device="cpu")
        SS_rot = torch.einsum('ki,kjcb,ca->ijab', rot, self.SS.to(dt), rot)
        op_rot = SS_rot.permute(1, 0, 3, 2).contiguous()
        return dict({"dd": corrf_c4v.corrf_2sOV2sOV_E2(state, env_c4v, SS_rot, lambda r: SS_rot if r % 2 == 0 else op_rot, dist)})

    def eval_obs(self, state, env_c4v, force_cpu=False):
        """<m>, <S^z>, <S^+>, <S^->, nearest-neighbour S.S from rho_2x1 and -- as the couplings are switched on -- S.S of the
        diagonal pair (j2), of the 3x1 pair (j3) and the chiral term (lambda) (models/j1j2.py:710-770, same labels and order)."""
        obs = dict()
        with torch.no_grad():
            if abs(self.j3) > 0:
                r31 = rdm_c4v.rdm3x1(state, env_c4v).cpu()
                obs["SS3x1"] = torch.einsum('ijab,ijab', r31, self.SS.to(r31.dtype))
            if abs(self.lmbd) > 0:
                r22 = rdm_c4v.rdm2x2(state, env_c4v).cpu()
                obs["ChiralT"] = torch.einsum('ijklabcd,ijklabcd', r22, self.chiral_term_rot.to(r22.dtype))
            if abs(self.j2) > 0:
                rd = rdm_c4v.rdm2x2_NNN_lowmem_sl(state, env_c4v).cpu()
                obs["SS_nnn"] = torch.einsum('ijab,ijab', rd, self.SS.to(rd.dtype))
            r2 = rdm_c4v.rdm2x1_sl(state, env_c4v).cpu()
            obs["SS2x1"] = _cast_to_real(torch.einsum('ijab,ijab', r2, self.SS_rot.to(r2.dtype)))
            r1 = torch.einsum('ijaj->ia', r2)
            r1 = r1 / torch.trace(r1)
            for l, op in self.obs_ops.items():
                obs[l] = torch.trace(r1 @ op.to(r1.dtype))
            obs["m"] = sqrt(abs(obs["sz"] ** 2 + obs["sp"] * obs["sm"]))
        labels = ["m"] + list(self.obs_ops.keys()) + ["SS2x1"]
        if abs(self.j2) > 0: labels += ["SS_nnn"]
        if abs(self.j3) > 0: labels += ["SS3x1"]
        if abs(self.lmbd) > 0: labels += ["ChiralT"]
        return [obs[l] for l in labels], labels
