"""Time J1J2.energy_per_site (4 x rdm2x2 + contraction with h_p) on the engine for a bench configuration."""
import sys, os, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "peps-torch_amd")); sys.path.insert(0, REPO)
import torch, numpy as np
import bench, config as cfg, _native
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
from models import j1j2
name = sys.argv[1] if len(sys.argv) > 1 else "generic_D6_chi128"
kind, D, chi, dtype = bench.CONFIGS[name]
dev = torch.device("cuda", 0); cfg.global_args.device = "cuda:0"
sites = bench.synth_sites(kind, D, dtype=dtype)
st = IPEPS({k: torch.from_numpy(v).to(dev) for k, v in sites.items()})
env = ENV(chi, st); init_env(st, env)
for d in cfg.ctm_args.ctm_move_sequence:
    for _ in range(2): ctmrg.ctm_MOVE(d, st, env)
m = j1j2.J1J2(j1=1.0, j2=0.5)
eng = _native.engine()
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = float(m.energy_per_site(st, env))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "energy_per_site", e, f"{dt:.3f} s", "arena_high GB", eng.stat("arena_high") / 1e9)
