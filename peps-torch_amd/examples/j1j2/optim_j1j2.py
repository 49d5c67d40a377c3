"""Energy minimisation of an iPEPS of the J1-J2 model on a generic unit cell by L-BFGS on gradients taken through the CTMRG --
same flags, tilings and output lines as the reference script (examples/j1j2/optim_j1j2.py:14-232), forward and backward passes
on the MI355X engine.

    python examples/j1j2/optim_j1j2.py --tiling BIPARTITE --bond_dim 2 --chi 16 --opt_max_iter 20 --seed 123 --out_prefix ex
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import json
import logging
import torch
import config as cfg
from ipeps.ipeps import IPEPS, read_ipeps, extend_bond_dim
from ctm.generic.env import ENV, init_env, ctmrg_conv_specC
from ctm.generic import ctmrg, transferops
from models import j1j2
from optim.ad_optim_lbfgs_mod import optimize_state

log = logging.getLogger(__name__)

parser = cfg.get_args_parser()
parser.add_argument("--j1", type=float, default=1., help="nearest-neighbour coupling")
parser.add_argument("--j2", type=float, default=0., help="next nearest-neighbour coupling")
parser.add_argument("--j3", type=float, default=0., help="next-to-next nearest-neighbour coupling")
parser.add_argument("--lmbd", type=float, default=0., help="chiral plaquette interaction")
parser.add_argument("--hz_stag", type=float, default=0., help="staggered mag. field")
parser.add_argument("--h_uni", nargs=3, type=float, default=[0, 0, 0], help="uniform mag. field with components in directions h^z, h^x, h^y")
parser.add_argument("--delta_zz", type=float, default=1., help="easy-axis (nearest-neighbour) anisotropy")
parser.add_argument("--tiling", default="BIPARTITE", help="tiling of the lattice", choices=["BIPARTITE", "1SITE", "2SITE", "4SITE", "8SITE"])
parser.add_argument("--top_freq", type=int, default=-1, help="frequency of transfer operator spectrum evaluation")
parser.add_argument("--top_n", type=int, default=2, help="number of leading eigenvalues of transfer operator to compute")
parser.add_argument("--ctm_conv_crit", default="CSPEC", help="ctm convergence criterion", choices=["CSPEC", "ENERGY"])

TILINGS = {
    "BIPARTITE": lambda c: ((((c[0] + abs(c[0]) * 2) % 2) + abs(c[1])) % 2, 0),
    "1SITE": lambda c: (0, 0),
    "2SITE": lambda c: ((c[0] + abs(c[0]) * 2) % 2, 0),
    "4SITE": lambda c: ((c[0] + abs(c[0]) * 2) % 2, (c[1] + abs(c[1]) * 2) % 2),
    "8SITE": lambda c: ((c[0] + 2 * (c[1] // 2)) % 4, c[1] % 2),
}
CELLS = {"BIPARTITE": (2, 1), "1SITE": (1, 1), "2SITE": (2, 1), "4SITE": (2, 2), "8SITE": (4, 2)}


def main(args=None):
    args, _ = parser.parse_known_args(args)
    cfg.configure(args)
    cfg.print_config()
    torch.set_num_threads(args.omp_cores)
    torch.manual_seed(args.seed)
    mk = lambda: j1j2.J1J2(j1=args.j1, j2=args.j2, j3=args.j3, lmbd=args.lmbd, hz_stag=args.hz_stag, h_uni=args.h_uni, delta_zz=args.delta_zz)
    model = mk()
    lattice_to_site = TILINGS[args.tiling]
    lX, lY = CELLS[args.tiling]
    dev, dt = cfg.global_args.device, cfg.global_args.torch_dtype

    if args.instate is not None:
        state = read_ipeps(args.instate, vertexToSite=lattice_to_site)
        if args.bond_dim > max(state.get_aux_bond_dims()):
            state = extend_bond_dim(state, args.bond_dim)
        state.add_noise(args.instate_noise)
    elif args.opt_resume is not None:
        state = IPEPS(dict(), lX=lX, lY=lY, vertexToSite=lattice_to_site)
        state.load_checkpoint(args.opt_resume)
    elif args.ipeps_init_type == 'RANDOM':
        D = args.bond_dim
        sites = {}
        for y in range(lY):
            for x in range(lX):
                if lattice_to_site((x, y)) == (x, y):
                    # zero tensor + uniform noise in [-0.5, 0.5), then max-abs normalisation (:99-113); drawn on the host
                    sites[(x, y)] = (torch.rand((model.phys_dim, D, D, D, D), dtype=dt, device='cpu') - 0.5).to(dev)
        state = IPEPS(sites, vertexToSite=lattice_to_site)
        state.normalize_()
    else:
        raise ValueError("Missing trial state: -instate=None and -ipeps_init_type= " + str(args.ipeps_init_type) + " is not supported")
    if not state.dtype == model.dtype:
        cfg.global_args.torch_dtype = state.dtype
        print(f"dtype of initial state {state.dtype} and model {model.dtype} do not match.")
        print(f"Setting default dtype to {cfg.global_args.torch_dtype} and reinitializing  the model")
        model = mk()
    print(state)

    energy_f = model.energy_2x2_1site_BP if args.tiling == "1SITE" else model.energy_per_site
    eval_obs_f = model.eval_obs_1site_BP if args.tiling == "1SITE" else model.eval_obs

    @torch.no_grad()
    def ctmrg_conv_energy(state, env, history, ctm_args=cfg.ctm_args):
        if not history:
            history = []
        history.append(energy_f(state, env).item())
        if (len(history) > 1 and abs(history[-1] - history[-2]) < ctm_args.ctm_conv_tol) or len(history) >= ctm_args.ctm_max_iter:
            log.info({"history_length": len(history), "history": history})
            return True, history
        return False, history

    ctmrg_conv_f = ctmrg_conv_specC if args.ctm_conv_crit == "CSPEC" else ctmrg_conv_energy

    ctm_env = ENV(args.chi, state)
    init_env(state, ctm_env)
    ctm_env, *ctm_log = ctmrg.run(state, ctm_env, conv_check=ctmrg_conv_f)
    loss0 = energy_f(state, ctm_env)
    obs_values, obs_labels = eval_obs_f(state, ctm_env)
    print(", ".join(["epoch", "energy"] + obs_labels))
    print(", ".join([f"{-1}", f"{loss0}"] + [f"{v}" for v in obs_values]))

    def loss_fn(state, ctm_env_in, opt_context):
        """environment by CTMRG (re-initialised from the state if opt_ctm_reinit), energy of the unit cell (:186-203)"""
        ctm_args, opt_args = opt_context["ctm_args"], opt_context["opt_args"]
        if opt_args.opt_ctm_reinit:
            init_env(state, ctm_env_in)
        ctm_env_out, *ctm_log = ctmrg.run(state, ctm_env_in, conv_check=ctmrg_conv_f, ctm_args=ctm_args)
        loss = energy_f(state, ctm_env_in)
        return (loss, ctm_env_in, *ctm_log)

    def _to_json(l):
        return dict({"re": [l[i, 0].item() for i in range(l.size()[0])], "im": [l[i, 1].item() for i in range(l.size()[0])]})

    @torch.no_grad()
    def obs_fn(state, ctm_env, opt_context):
        if not opt_context.get("line_search", False):
            epoch = len(opt_context["loss_history"]["loss"])
            loss = opt_context["loss_history"]["loss"][-1]
            obs_values, obs_labels = eval_obs_f(state, ctm_env)
            print(", ".join([f"{epoch}", f"{loss}"] + [f"{v}" for v in obs_values]))
            log.info("Norm(sites): " + ", ".join([f"{t.norm()}" for c, t in state.sites.items()]))
            if args.top_freq > 0 and epoch % args.top_freq == 0:
                for c, d in [((0, 0), (1, 0)), ((0, 0), (0, 1)), ((1, 1), (1, 0)), ((1, 1), (0, 1))]:
                    print(f"TOP spectrum(T)[{c},{d}] ", end="")
                    l = transferops.get_Top_spec(args.top_n, c, d, state, ctm_env)
                    print("TOP " + json.dumps(_to_json(l)))

    optimize_state(state, ctm_env, loss_fn, obs_fn=obs_fn)

    # final observables of the best variational state
    outputstatefile = args.out_prefix + "_state.json"
    state = read_ipeps(outputstatefile, vertexToSite=state.vertexToSite)
    ctm_env = ENV(args.chi, state)
    init_env(state, ctm_env)
    ctm_env, *ctm_log = ctmrg.run(state, ctm_env, conv_check=ctmrg_conv_f)
    loss0 = energy_f(state, ctm_env)
    obs_values, obs_labels = eval_obs_f(state, ctm_env)
    print(", ".join([f"{args.opt_max_iter}", f"{loss0}"] + [f"{v}" for v in obs_values]))
    return float(loss0)


if __name__ == '__main__':
    main()
