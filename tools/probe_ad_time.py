"""Development probe: wall time of energy + gradient through N differentiable C4v moves (explicit route) on the engine."""
import sys, os, time, numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "peps-torch_amd"))
import _native, config as cfg
from ipeps.ipeps_c4v import IPEPS_C4V
from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
from ctm.one_site_c4v import ctmrg_c4v
from groups.pg import make_c4v_symm
from models import j1j2
eng = _native.engine()
D, chi, nmoves = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1)
a0 = make_c4v_symm(torch.from_numpy(rng.random((2, D, D, D, D)))).cuda()
st0 = IPEPS_C4V(a0.clone()); env0 = ENV_C4V(chi, st0); init_env(st0, env0)
for _ in range(8):
    ctmrg_c4v.ctm_MOVE_sl(st0.site(), env0)
C0, T0 = env0.get_C().clone(), env0.get_T().clone()
model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.5)
for ck in (False, True):
    cfg.ctm_args.fwd_checkpoint_move = ck
    for rep in range(2):
        a = a0.clone().requires_grad_(True)
        st = IPEPS_C4V(a); env = ENV_C4V(chi, st); env.C[env.keyC] = C0.clone(); env.T[env.keyT] = T0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(nmoves):
            ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
        e = model.energy_1x1_lowmem(st, env)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        e.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"D={D} chi={chi} n={chi*D*D} moves={nmoves} checkpoint={ck}: forward {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB, |grad| {float(a.grad.norm()):.3e}", flush=True)
