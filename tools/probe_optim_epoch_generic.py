"""Wall time of ONE optimisation epoch of examples/j1j2/optim_j1j2.py on a 2x2 cell -- loss_fn (init_env, N CTM iterations of 8
directional moves, plaquette energy of the four sites) and its backward pass -- on the engine (default) or, with --reference, with
the reference's modules on this machine's CPU cores (/root/reference, build container only).
usage: probe_optim_epoch_generic.py D chi iterations [--reference] [--threads T]"""
import sys, os, time
import numpy as np
import torch
ref = "--reference" in sys.argv
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, "/root/reference" if ref else os.path.join(R, "peps-torch_amd"))
if "--threads" in sys.argv:
    torch.set_num_threads(int(sys.argv[sys.argv.index("--threads") + 1]))
import config as cfg
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg, rdm
from models import j1j2
D, chi, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = "cpu" if ref else "cuda"
cfg.global_args.device = dev
rng = np.random.default_rng(1)
s0 = {(x, y): torch.from_numpy(rng.random((2, D, D, D, D)) - 0.5).to(dev) for y in range(2) for x in range(2)}
model = j1j2.J1J2(j1=1.0, j2=0.3)
cfg.ctm_args.ctm_max_iter = iters
sync = (lambda: None) if ref else torch.cuda.synchronize


def conv(state, env, history, ctm_args=cfg.ctm_args):
    history = (history or []) + [0.]
    return len(history) >= ctm_args.ctm_max_iter, history


def energy(state, env):
    if not ref:
        return model.energy_2x2_4site(state, env)
    e = 0.          # energy_per_site with rdm2x2_legacy standing in for rdm2x2 (opt_einsum is not installed here)
    for c in state.sites.keys():
        e = e + torch.einsum('ijklabcd,ijklabcd', rdm.rdm2x2_legacy(c, state, env), model.get_hp(c))
    return e / len(state.sites)


if not ref and os.environ.get("CTM_OPTS"):
    from backend import get_engine
    for kv in os.environ["CTM_OPTS"].split(","):
        get_engine().set_option(kv.split("=")[0], float(kv.split("=")[1]))
env = None
for rep in range(int(os.environ.get("REPS", 4))):
    sites = {c: (t + rep * 1e-3 * torch.from_numpy(rng.random((2, D, D, D, D)) - 0.5).to(dev)).requires_grad_(True) for c, t in s0.items()}
    st = IPEPS(sites, lX=2, lY=2)
    sync(); t0 = time.perf_counter()
    if env is None:
        env = ENV(chi, st)
    init_env(st, env)
    ctmrg.run(st, env, conv_check=conv)
    e = energy(st, env)
    sync(); t1 = time.perf_counter()
    e.backward()
    sync(); t2 = time.perf_counter()
    env = env.detach()
    g = torch.cat([t.grad.reshape(-1) for t in sites.values()]).norm()
    if not ref:
        from backend import get_engine
        eng = get_engine()
        print(f"    decompositions {int(eng.stat('jacobi_calls'))}, Jacobi sweeps {int(eng.stat('total_sweeps'))}, warm starts {int(eng.stat('eigh_warm_hits'))}", flush=True)
        eng.timers(reset=True)
    print(f"{'reference CPU (%d threads)' % torch.get_num_threads() if ref else 'engine'}: 2x2 cell D={D} chi={chi} n={chi*D*D} iterations={iters} "
          f"evaluation {rep}: loss {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, epoch {1e3*(t2-t0):.1f} ms, E={float(e):.12f} |grad|={float(g):.6e}", flush=True)
