"""Development probe: C4v CTM of the RVB state (slowly decaying corner spectrum) at chi = 64: time per move and which solver ran."""
import sys, os, time, numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "peps-torch_amd"))
import _native
from ipeps.ipeps_c4v import IPEPS_C4V
from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
from ctm.one_site_c4v import ctmrg_c4v
eng = _native.engine()
for kv in sys.argv[2:]:
    k_, v_ = kv.split("="); eng.set_option(k_, float(v_))
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = np.load(os.path.join(R, "tests", "golden", "rvb_c4v.npz"))
st = IPEPS_C4V(torch.from_numpy(g["site"]).cuda())
env = ENV_C4V(chi, st); init_env(st, env)
eng.timers(reset=True)
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if i < 12 or i % 5 == 0:
        d = torch.diagonal(env.get_C()).abs()
        print(f"move {i:3d} {1e3*dt:8.2f} ms  si_hits {int(eng.stat('si_hits'))} fallbacks {int(eng.stat('si_fallbacks'))} warm_hits {int(eng.stat('eigh_warm_hits'))} "
              f"si_iters {int(eng.stat('si_total_iters'))}  C[8]/C[0] {float(d[8]/d[0]):.3e} C[-1]/C[0] {float(d[-1]/d[0]):.3e}", flush=True)
