"""Single-site C4v-symmetric iPEPS (reference ipeps/ipeps_c4v.py:6-128)."""
import torch
import config as cfg
import ipeps.ipeps as ipeps
from groups.pg import make_c4v_symm


class IPEPS_C4V(ipeps.IPEPS):
    def __init__(self, site=None, peps_args=cfg.peps_args, global_args=cfg.global_args):
        if site is not None:
            assert isinstance(site, torch.Tensor), "site is not a torch.Tensor"
            sites = {(0, 0): site}
        else:
            sites = dict()
        super().__init__(sites, lX=1, lY=1, peps_args=peps_args, global_args=global_args)

    def site(self, coord=None):
        return self.sites[(0, 0)]

    def add_noise(self, noise, symmetrize=False):
        r = torch.rand(self.site().size(), dtype=self.dtype, device=self.device)
        self.sites[(0, 0)] = self.site() + noise * r
        if symmetrize:
            self.sites[(0, 0)] = _symm(self.site())

    def write_to_file(self, outputfile, symmetrize=True, **kwargs):
        ipeps.write_ipeps(to_ipeps_c4v(self) if symmetrize else self, outputfile, **kwargs)


def _symm(A):
    if A.is_complex():
        return make_c4v_symm(A.real) + make_c4v_symm(A.imag, irreps=["A2"]) * 1.0j
    return make_c4v_symm(A)


def extend_bond_dim(state, new_d):
    return ipeps.extend_bond_dim(state, new_d)


def to_ipeps_c4v(state, normalize=False):
    assert len(state.sites.items()) == 1, "state has more than a single on-site tensor"
    A = _symm(next(iter(state.sites.values())))
    if normalize:
        A = A / A.norm()
    return IPEPS_C4V(A)


def read_ipeps_c4v(jsonfile, aux_seq=[0, 1, 2, 3], peps_args=cfg.peps_args, global_args=cfg.global_args):
    state = ipeps.read_ipeps(jsonfile, aux_seq=aux_seq, peps_args=peps_args, global_args=global_args)
    assert len(state.sites.items()) == 1, "state has more than a single on-site tensor"
    return IPEPS_C4V(next(iter(state.sites.values())), peps_args=peps_args, global_args=global_args)
