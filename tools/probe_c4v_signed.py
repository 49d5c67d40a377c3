"""C4v D=4 chi=64 on signed random tensors: per-sweep time and solver route.  usage: probe_c4v_signed.py [old|new] [nsweeps] [opt=value ...]
old = round-3 construction 2 * (sym(U) / max) - 1, new = sym(2U - 1) / max (bench.synth_sites(signed=True))."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "peps-torch_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, config as cfg
cfg.global_args.device = "cuda:0"
import _native
from bench import synth_sites
from ipeps.ipeps_c4v import IPEPS_C4V
from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
from ctm.one_site_c4v import ctmrg_c4v
which = sys.argv[1] if len(sys.argv) > 1 else "new"
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 40
eng = _native.engine()
for kv in [a for a in sys.argv[3:] if "=" in a]:
    k, v = kv.split("="); eng.set_option(k, float(v))
if which == "old":
    A = synth_sites("c4v", 4)[(0, 0)]; A = 2.0 * A - 1.0; A = A / np.abs(A).max()
else:
    A = synth_sites("c4v", 4, signed=True)[(0, 0)]
st = IPEPS_C4V(torch.from_numpy(A).cuda())
env = ENV_C4V(64, st); init_env(st, env)
a = st.site()
keys = ("eigh_warm_hits", "eigh_warm_rejects", "eigh_orth_hits", "eigh_orth_fails", "si_hits", "lz_hits", "si_total_iters")
prev = {k: 0 for k in keys}
for sw in range(ns):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctmrg_c4v.ctm_MOVE_sl(a, env)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cur = {k: eng.stat(k) for k in keys}
    d = {k: int(cur[k] - prev[k]) for k in keys}; prev = cur
    print(f"sweep {sw+1:3d} {dt*1e3:8.2f} ms  " + " ".join(f"{k}={v}" for k, v in d.items() if v), flush=True)
