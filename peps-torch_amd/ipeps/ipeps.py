"""iPEPS state container with the reference's surface (ipeps/ipeps.py:89-247, 339-441, 467-534).

Site tensors are float64 device tensors a[s,u,l,d,r] (physical, up, left, down, right) keyed by the
coordinates of the elementary unit cell; `vertexToSite` maps any lattice vertex into the cell.
"""
import json
import warnings
from collections import OrderedDict
import torch
import config as cfg
from ipeps.tensor_io import (read_bare_json_tensor_np, read_bare_json_tensor_np_legacy,
                             serialize_bare_tensor_legacy, serialize_bare_tensor_np)


def from_pattern(pattern):
    """pattern = rows (y) of labels along x  ->  (sites-by-label, coord->label)."""
    if isinstance(pattern, dict):
        site2index = dict(pattern)
    else:
        site2index = {(x, y): lab for y, row in enumerate(pattern) for x, lab in enumerate(row)}
    sites = OrderedDict()
    for c, lab in site2index.items():
        sites.setdefault(lab, c)
    return sites, site2index


class IPEPS():
    def __init__(self, sites=None, vertexToSite=None, pattern=None, lX=None, lY=None, peps_args=cfg.peps_args,
                 global_args=cfg.global_args):
        self.dtype = global_args.torch_dtype
        self.device = global_args.device
        self.sites = OrderedDict(sites) if sites else OrderedDict()
        if self.sites:
            t0 = next(iter(self.sites.values()))
            self.dtype, self.device = t0.dtype, t0.device
        if pattern is not None:
            _s, s2i = from_pattern(pattern)
            self.lX = max(c[0] for c in s2i) + 1
            self.lY = max(c[1] for c in s2i) + 1
        elif lX and lY:
            self.lX, self.lY = lX, lY
        elif self.sites:
            xs = [c[0] for c in self.sites]; ys = [c[1] for c in self.sites]
            self.lX, self.lY = max(xs) - min(xs) + 1, max(ys) - min(ys) + 1
        else:
            raise Exception("lX and lY has to be set either directly or implicitly by sites or pattern")
        if pattern is not None:
            if not self.sites:
                raise Exception("Pattern provided, but sites not set. Please provide pattern and sites.")
            self._pattern = pattern
            _sites, self._site2index = from_pattern(pattern)
            self._label2index = {self._site2index[c]: c for c in self.sites.keys()}
            self.vertexToSite = lambda x: self._label2index[self._site2index[
                ((x[0] + abs(x[0]) * self.lX) % self.lX, (x[1] + abs(x[1]) * self.lY) % self.lY)]]
        elif vertexToSite is not None:
            self.vertexToSite = vertexToSite
        else:
            def _pbc(coord):
                x, y = coord
                return ((x + abs(x) * self.lX) % self.lX, (y + abs(y) * self.lY) % self.lY)
            self.vertexToSite = _pbc

    def site(self, coord):
        return self.sites[self.vertexToSite(coord)]

    def get_parameters(self):
        return self.sites.values()

    def get_checkpoint(self):
        return self.sites

    def load_checkpoint(self, checkpoint_file):
        """ipeps.py:268-284: on-site tensors from an optimiser checkpoint (`optim.ad_optim_lbfgs_mod.store_checkpoint`)."""
        checkpoint = torch.load(checkpoint_file, map_location=self.device, weights_only=False)
        self.sites = checkpoint["parameters"]
        for site_t in self.sites.values():
            site_t.requires_grad_(False)
        if True in [s.is_complex() for s in self.sites.values()]:
            self.dtype = torch.complex128

    def get_aux_bond_dims(self):
        return [d for key in self.sites.keys() for d in self.sites[key].size()[1:]]

    def add_noise(self, noise, noise_f=None):
        for coord in self.sites.keys():
            r = noise_f(self.sites[coord].size(), dtype=self.dtype, device=self.device) if noise_f else \
                torch.rand(self.sites[coord].size(), dtype=self.dtype, device=self.device) - 0.5
            self.sites[coord] = self.sites[coord] + noise * r

    def normalize_(self):
        for c in self.sites.keys():
            self.sites[c] = self.sites[c] / self.sites[c].abs().max()

    def write_to_file(self, outputfile, aux_seq=[0, 1, 2, 3], tol=1.0e-14, normalize=False):
        write_ipeps(self, outputfile, aux_seq=aux_seq, tol=tol, normalize=normalize)

    def __str__(self):
        s = f"lX x lY: {self.lX} x {self.lY}\n"
        keys = list(self.sites.keys())
        for i, (c, t) in enumerate(self.sites.items()):
            s += f"a{i} {c}: {tuple(t.size())}\n"
        for y in range(-self.lY, 2 * self.lY):
            s += f"{y:+} " + " ".join(f"a{keys.index(self.vertexToSite((x, y)))}" for x in range(-self.lX, 2 * self.lX)) + "\n"
        return s


def read_ipeps(jsonfile, vertexToSite=None, aux_seq=[0, 1, 2, 3], peps_args=cfg.peps_args, global_args=cfg.global_args):
    """JSON -> IPEPS.  `aux_seq` (or the file's "aux_ind_seq") gives the order of the auxiliary legs in
    the file relative to [up, left, down, right]."""
    asq = [x + 1 for x in aux_seq]
    sites = OrderedDict()
    promoted = False
    with open(jsonfile) as j:
        raw = json.load(j)
    if "aux_ind_seq" in raw:
        asq = [x + 1 for x in raw["aux_ind_seq"]]
    by_id = {s["siteId"]: s for s in raw["sites"]}
    for ts in raw["map"]:
        coord = (ts["x"], ts["y"])
        if ts["siteId"] not in by_id:
            raise Exception("Tensor with siteId: " + ts["siteId"] + " NOT FOUND in \"sites\"")
        t = by_id[ts["siteId"]]
        X = read_bare_json_tensor_np(t) if t.get("format") == "1D" else read_bare_json_tensor_np_legacy(t)
        X = torch.from_numpy(X).permute((0, *asq)).contiguous()
        if global_args.torch_dtype.is_complex and not X.is_complex():
            X = X + 0.j
            promoted = True
        sites[coord] = X.to(global_args.device)
    if promoted:
        warnings.warn("Some of the tensors were promoted from float to complex dtype", Warning)
    lX = raw["sizeM"] if "sizeM" in raw else raw["lX"]
    lY = raw["sizeN"] if "sizeN" in raw else raw["lY"]
    pattern = raw["pattern"] if (vertexToSite is None and "pattern" in raw) else None
    if pattern is not None:
        # labels in the file's pattern are siteIds: translate sites to label keys
        id_of = {(m["x"], m["y"]): m["siteId"] for m in raw["map"]}
        st = IPEPS(sites, lX=lX, lY=lY, peps_args=peps_args, global_args=global_args)
        coord_of_label = {lab: c for c, lab in id_of.items()}
        st.vertexToSite = lambda v: coord_of_label[pattern[(v[1] + abs(v[1]) * lY) % lY][(v[0] + abs(v[0]) * lX) % lX]]
        return st
    return IPEPS(sites, vertexToSite, lX=lX, lY=lY, peps_args=peps_args, global_args=global_args)


def extend_bond_dim(state, new_d):
    for coord, site in state.sites.items():
        dims = site.size()
        if any(new_d < d for d in dims[1:]):
            raise ValueError("Desired dimension is smaller than following aux dimensions: " + str(dims[1:]))
        ns = torch.zeros((dims[0], new_d, new_d, new_d, new_d), dtype=state.dtype, device=state.device)
        ns[:, :dims[1], :dims[2], :dims[3], :dims[4]] = site
        state.sites[coord] = ns
    return state


def write_ipeps(state, outputfile, aux_seq=[0, 1, 2, 3], tol=1.0e-14, normalize=False, peps_args=cfg.peps_args,
                global_args=cfg.global_args):
    asq = [x + 1 for x in aux_seq]
    js = {"lX": state.lX, "lY": state.lY, "sites": [], "siteIds": [], "map": []}
    for nid, (coord, site) in enumerate(state.sites.items()):
        if normalize:
            site = site / site.abs().max()
        t = site.permute((0, *asq))
        jt = serialize_bare_tensor_np(t) if global_args.tensor_io_format == "1D" else serialize_bare_tensor_legacy(t, tol)
        jt["siteId"] = f"A{nid}"
        js["sites"].append(jt); js["siteIds"].append(jt["siteId"])
        js["map"].append({"siteId": jt["siteId"], "x": coord[0], "y": coord[1]})
    id_of = {(m["x"], m["y"]): m["siteId"] for m in js["map"]}
    js["pattern"] = [[id_of[state.vertexToSite((x, y))] for x in range(state.lX)] for y in range(state.lY)]
    with open(outputfile, 'w') as f:
        json.dump(js, f, indent=4, separators=(',', ': '))
