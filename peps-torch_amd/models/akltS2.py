"""AKLT S=2 model on the square lattice (reference models/akltS2.py:14-170): the nearest-neighbour projector onto total
spin 4, h = (1/14)(S.S + 7/10 (S.S)^2 + 7/45 (S.S)^3 + 1/90 (S.S)^4); energy from the native rdm2x1 / rdm1x2."""
import itertools
from math import sqrt
import torch
import config as cfg
import groups.su2 as su2
from ctm.generic import rdm

_cast_to_real = rdm._cast_to_real


class AKLTS2():
    def __init__(self, global_args=cfg.global_args):
        self.dtype = global_args.torch_dtype
        self.device = 'cpu'                      # operators are <= 25 x 25: the RDMs come back to the host
        self.phys_dim = 5
        self.h, self.SS = self.get_h()
        self.obs_ops = self.get_obs()

    def get_h(self):
        pd = self.phys_dim
        s5 = su2.SU2(pd, dtype=self.dtype, device=self.device)
        k = 'ij,ab->iajb'
        SS = torch.einsum(k, s5.SZ(), s5.SZ()) + 0.5 * (torch.einsum(k, s5.SP(), s5.SM()) + torch.einsum(k, s5.SM(), s5.SP()))
        SS = SS.reshape(pd * pd, pd * pd)
        SS2 = SS @ SS
        h = (1. / 14) * (SS + (7. / 10.) * SS2 + (7. / 45.) * SS2 @ SS + (1. / 90.) * SS2 @ SS2)
        return h.reshape(pd, pd, pd, pd), SS.reshape(pd, pd, pd, pd)

    def get_obs(self):
        s5 = su2.SU2(self.phys_dim, dtype=self.dtype, device=self.device)
        return {"sz": s5.SZ(), "sp": s5.SP(), "sm": s5.SM()}

    def energy_2x1_1x2(self, state, env, **kwargs):
        """E/site = (1/N) sum_sites tr(rho_2x1 h) + tr(rho_1x2 h)   (models/akltS2.py:56-118)."""
        e = 0.
        for coord in state.sites.keys():
            r21 = rdm.rdm2x1(coord, state, env).cpu()
            r12 = rdm.rdm1x2(coord, state, env).cpu()
            e = e + torch.einsum('ijab,ijab', r21, self.h.to(r21.dtype)) + torch.einsum('ijab,ijab', r12, self.h.to(r12.dtype))
        return _cast_to_real(e / len(state.sites))

    def eval_obs(self, state, env):
        """avg m, m per site, <S^z>,<S^+>,<S^-> per site, nearest-neighbour S.S on the bonds (models/akltS2.py:120-165)."""
        obs = {"avg_m": 0.}
        with torch.no_grad():
            for coord in state.sites.keys():
                r = rdm.rdm1x1(coord, state, env).cpu()
                for label, op in self.obs_ops.items():
                    obs[f"{label}{coord}"] = torch.trace(r @ op.to(r.dtype))
                obs[f"m{coord}"] = sqrt(abs(obs[f"sz{coord}"] ** 2 + obs[f"sp{coord}"] * obs[f"sm{coord}"]))
                obs["avg_m"] += obs[f"m{coord}"]
            obs["avg_m"] = obs["avg_m"] / len(state.sites)
            for coord in state.sites.keys():
                r21 = rdm.rdm2x1(coord, state, env).cpu()
                r12 = rdm.rdm1x2(coord, state, env).cpu()
                obs[f"SS2x1{coord}"] = _cast_to_real(torch.einsum('ijab,ijab', r21, self.SS.to(r21.dtype)))
                obs[f"SS1x2{coord}"] = _cast_to_real(torch.einsum('ijab,ijab', r12, self.SS.to(r12.dtype)))
        labels = ["avg_m"] + [f"m{c}" for c in state.sites.keys()] \
            + [f"{lc[1]}{lc[0]}" for lc in itertools.product(state.sites.keys(), self.obs_ops.keys())]
        labels += [f"SS2x1{c}" for c in state.sites.keys()] + [f"SS1x2{c}" for c in state.sites.keys()]
        return [obs[l] for l in labels], labels
