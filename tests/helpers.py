"""Shared helpers of the GPU parity tests: fixtures -> device state/env, oracle twins."""
import numpy as np
import torch
from conftest import golden

DIRS = {'UP': (0, -1), 'LEFT': (-1, 0), 'DOWN': (0, 1), 'RIGHT': (1, 0)}


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def sites_from(g):
    return {tuple(int(v) for v in k.split('_')[1:]): g[k] for k in g.files if k.startswith('site_')}


def env_from(g, prefix):
    C, T = {}, {}
    for k in g.files:
        if k.startswith(prefix + 'C_') or k.startswith(prefix + 'T_'):
            x, y, vx, vy = (int(v) for v in k[len(prefix) + 2:].split('_'))
            (C if k[len(prefix)] == 'C' else T)[((x, y), (vx, vy))] = g[k]
    return C, T


def device_state_env(sites, C, T, chi):
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    st = IPEPS({k: dev(v) for k, v in sites.items()})
    env = ENV(chi, st)
    env.C = {k: dev(v) for k, v in C.items()}
    env.T = {k: dev(v) for k, v in T.items()}
    return st, env


def oracle_state_env(sites, C, T, chi):
    from oracle import ctm_oracle as O
    ost = O.State(sites)
    oe = O.Env(chi)
    oe.C = {k: v.copy() for k, v in C.items()}
    oe.T = {k: v.copy() for k, v in T.items()}
    return ost, oe


def relerr(a, b):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
