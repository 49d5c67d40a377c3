"""GPU probe: how far the operator M = R^T Rt of one (direction, site) unit moves from sweep to sweep, as a matrix and in its singular
values (the Krylov basis / Ritz matrix of its truncation can only be continuous from sweep to sweep if the matrix itself is)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "peps-torch_amd"))
import numpy as np, torch
import config as cfg
cfg.global_args.device = "cuda:0"
import _native
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg, ctm_components as cc
D, chi, ns = (int(a) for a in (sys.argv[1:4] + ["4", "64", "9"][len(sys.argv) - 1:]))
eng = _native.engine()
if os.environ.get("SIGN_FOLLOW"):
    for e in [eng] + list(eng.workers): e.set_option("sign_follow", 1)
rng = np.random.default_rng(5)
sites = {}
for y in range(2):
    for x in range(2):
        A = rng.random((2, D, D, D, D)) - 0.5
        sites[(x, y)] = torch.from_numpy(A / np.abs(A).max()).cuda()
st = IPEPS(dict(sites)); env = ENV(chi, st); init_env(st, env)
prev = None
for s in range(ns):
    R, Rt = cc.halves_of_4x4_CTM_MOVE_UP((0, 0), st, env)
    M = R.t() @ Rt
    M = M / M.abs().max()
    S = torch.linalg.svdvals(M)
    if prev is not None:
        Mp, Sp = prev
        print(f"sweep {s}: |dM|/|M| = {float((M - Mp).norm() / M.norm()):.3e}   | |M| - |Mp| |/|M| = {float((M.abs() - Mp.abs()).norm() / M.norm()):.3e}   "
              f"max |dS|/S0 (leading {chi + 1}) = {float((S[:chi + 1] / S[0] - Sp[:chi + 1] / Sp[0]).abs().max()):.3e}", flush=True)
    prev = (M, S)
    for d in cfg.ctm_args.ctm_move_sequence:
        for _ in range(2):
            ctmrg.ctm_MOVE(d, st, env)
