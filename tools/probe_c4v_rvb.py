"""Per-move time of a C4v CTMRG on the RVB state (D = 3) at chi = 64: the flat-leading-spectrum case of the symmetric truncation
(DESIGN.md section 4).  Run on the GPU box:  python tools/probe_c4v_rvb.py [chi] [moves]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "peps-torch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import config as cfg
from ipeps.ipeps_c4v import IPEPS_C4V
from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
from ctm.one_site_c4v import ctmrg_c4v
import _native
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 64
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = np.load(os.path.join(ROOT, "tests", "golden", "rvb_c4v.npz"))
st = IPEPS_C4V(torch.as_tensor(g["site"]).cuda())
env = ENV_C4V(chi, st)
init_env(st, env)
eng = _native.engine()
times = []
def conv(state, env, history, ctm_args=cfg.ctm_args):
    torch.cuda.synchronize()
    history = history if history is not None else []
    history.append(time.perf_counter())
    return False, history
cfg.ctm_args.ctm_max_iter = moves
eng.timers(reset=True)
_, hist, *_ = ctmrg_c4v.run(st, env, conv_check=conv)
dt = np.diff(np.array(hist)) * 1e3
print("ms per move:", " ".join(f"{x:.1f}" for x in dt))
print(f"median of the last {len(dt) // 2}: {np.median(dt[len(dt) // 2:]):.2f} ms")
print({k: eng.stat(k) for k in ("eigh_orth_hits", "eigh_orth_fails", "eigh_warm_hits", "si_hits", "si_fallbacks", "si_total_iters")})
