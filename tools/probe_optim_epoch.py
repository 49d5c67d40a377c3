"""Wall time of ONE optimisation epoch of examples/j1j2/optim_j1j2_c4v.py -- loss_fn (symmetrise, init_env, N CTM moves, energy) and
its backward pass -- on the engine (default) or, with --reference, with the reference's modules on this machine's CPU cores
(/root/reference, build container only).  usage: probe_optim_epoch.py D chi moves [--reference] [--threads T]"""
import sys, os, time
import numpy as np
import torch
ref = "--reference" in sys.argv
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, "/root/reference" if ref else os.path.join(R, "peps-torch_amd"))
if "--threads" in sys.argv:
    torch.set_num_threads(int(sys.argv[sys.argv.index("--threads") + 1]))
import config as cfg
from ipeps.ipeps_c4v import IPEPS_C4V, to_ipeps_c4v
from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
from ctm.one_site_c4v import ctmrg_c4v
from groups.pg import make_c4v_symm
from models import j1j2
D, chi, nmoves = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = "cpu" if ref else "cuda"
cfg.global_args.device = dev
rng = np.random.default_rng(1)
a0 = make_c4v_symm(torch.from_numpy(rng.random((2, D, D, D, D)))).to(dev)
a0 = a0 / a0.norm()
model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.5)
cfg.ctm_args.ctm_max_iter = nmoves
sync = (lambda: None) if ref else torch.cuda.synchronize


def conv(state, env, history, ctm_args=cfg.ctm_args):
    history = (history or []) + [0.]
    return len(history) >= ctm_args.ctm_max_iter, history


env = None
step = float(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 1e-3
for rep in range(4):
    # every evaluation moves the tensor a little (an optimisation step) and reuses the environment object of the previous one
    a = (a0 + rep * step * make_c4v_symm(torch.from_numpy(rng.random((2, D, D, D, D)))).to(dev) / D ** 2).requires_grad_(True)
    st = IPEPS_C4V(a)
    sync(); t0 = time.perf_counter()
    ss = to_ipeps_c4v(st, normalize=True)
    if env is None:
        env = ENV_C4V(chi, ss)
    init_env(ss, env)
    env, *_ = ctmrg_c4v.run(ss, env, conv_check=conv)
    e = model.energy_1x1_lowmem(ss, env)
    sync(); t1 = time.perf_counter()
    e.backward()
    sync(); t2 = time.perf_counter()
    env = env.detach()
    print(f"{'reference CPU (%d threads)' % torch.get_num_threads() if ref else 'engine'}: D={D} chi={chi} n={chi*D*D} moves={nmoves} evaluation {rep}: "
          f"loss {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, epoch {1e3*(t2-t0):.1f} ms, E={float(e):.12f} |grad|={float(a.grad.norm()):.6e}", flush=True)
