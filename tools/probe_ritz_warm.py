"""GPU probe: Jacobi sweeps of the Ritz extraction of the block Krylov solver from sweep to sweep, with and without its warm start
(option ritz_warm; csrc/svd_leading.hip).  Usage: python tools/probe_ritz_warm.py [D chi nsweeps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "peps-torch_amd"))
import numpy as np, torch
import config as cfg
cfg.global_args.device = "cuda:0"
import _native
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
D, chi, ns = (int(a) for a in (sys.argv[1:4] + ["4", "64", "6"][len(sys.argv) - 1:]))
eng = _native.engine()
if os.environ.get("SIGN_FOLLOW"):
    for e in [eng] + list(eng.workers): e.set_option("sign_follow", 1)
rng = np.random.default_rng(5)
sites = {}
for y in range(2):
    for x in range(2):
        A = rng.random((2, D, D, D, D)) - 0.5
        sites[(x, y)] = torch.from_numpy(A / np.abs(A).max()).cuda()
for warm in (1, 0):
    for e in [eng] + list(eng.workers): e.set_option("ritz_warm", warm)
    st = IPEPS(dict(sites)); env = ENV(chi, st); init_env(st, env)
    cfg.ctm_args.concurrent_units = False
    for s in range(ns):
        x0, w0, s0 = eng.stat("lz_extractions"), eng.stat("ritz_warm_starts"), eng.stat("ritz_sweeps")
        if s == ns - 1: eng.set_option("jacobi_verbose", 2)
        for d in cfg.ctm_args.ctm_move_sequence:
            for _ in range(2):
                ctmrg.ctm_MOVE(d, st, env)
                if s == ns - 1: eng.set_option("jacobi_verbose", 0)      # (one move's worth of lines)
        torch.cuda.synchronize()
        nx = eng.stat("lz_extractions") - x0
        print(f"ritz_warm={warm} sweep {s}: {int(nx)} extractions, {int(eng.stat('ritz_warm_starts') - w0)} warm started, "
              f"{(eng.stat('ritz_sweeps') - s0) / max(nx, 1):.2f} Jacobi sweeps per extraction", flush=True)
