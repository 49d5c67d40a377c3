"""Energy minimisation of a single-site C4v iPEPS of the J1-J2 model by L-BFGS on gradients taken through the CTMRG -- same flags
and output lines as the reference script (examples/j1j2/optim_j1j2_c4v.py:20-177), forward and backward passes on the MI355X engine.

    python examples/j1j2/optim_j1j2_c4v.py --bond_dim 2 --chi 16 --opt_max_iter 20 --seed 123 --out_prefix ex-c4v
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import json
import torch
import config as cfg
from ipeps.ipeps_c4v import IPEPS_C4V, read_ipeps_c4v, extend_bond_dim, to_ipeps_c4v
from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
from ctm.one_site_c4v import ctmrg_c4v, transferops_c4v
from ctm.one_site_c4v.rdm_c4v import rdm2x1_sl
from models import j1j2
from optim.ad_optim_lbfgs_mod import optimize_state

parser = cfg.get_args_parser()
parser.add_argument("--j1", type=float, default=1., help="nearest-neighbour coupling")
parser.add_argument("--j2", type=float, default=0., help="next nearest-neighbour coupling")
parser.add_argument("--j3", type=float, default=0., help="next-to-next nearest-neighbour coupling")
parser.add_argument("--hz_stag", type=float, default=0., help="staggered mag. field")
parser.add_argument("--h_uni", nargs=3, type=float, default=[0, 0, 0], help="uniform mag. field with components in directions h^z, h^x, h^y")
parser.add_argument("--delta_zz", type=float, default=1., help="easy-axis (nearest-neighbour) anisotropy")
parser.add_argument("--top_freq", type=int, default=-1, help="frequency of transfer operator spectrum evaluation")
parser.add_argument("--top_n", type=int, default=2, help="number of leading eigenvalues of transfer operator to compute")
parser.add_argument("--force_cpu", action='store_true', help="accepted for compatibility; the energy is evaluated on the GPU")


def main(args=None):
    args, _ = parser.parse_known_args(args)
    cfg.configure(args)
    cfg.print_config()
    torch.set_num_threads(args.omp_cores)
    torch.manual_seed(args.seed)
    model = j1j2.J1J2_C4V_BIPARTITE(j1=args.j1, j2=args.j2, j3=args.j3, hz_stag=args.hz_stag, h_uni=args.h_uni, delta_zz=args.delta_zz)
    energy_f = model.energy_1x1_lowmem
    dev, dt = cfg.global_args.device, cfg.global_args.torch_dtype

    if args.instate is not None:
        state = read_ipeps_c4v(args.instate)
        if args.bond_dim > max(state.get_aux_bond_dims()):
            state = extend_bond_dim(state, args.bond_dim)
        state.add_noise(args.instate_noise)
        state.sites[(0, 0)] = state.site() / state.site().norm()
    elif args.opt_resume is not None:
        state = IPEPS_C4V()
        state.load_checkpoint(args.opt_resume)
    elif args.ipeps_init_type == 'RANDOM':
        D = args.bond_dim
        A = torch.rand((model.phys_dim, D, D, D, D), dtype=dt, device='cpu')          # the generator of the host: the same draw on any device
        state = IPEPS_C4V((A / A.norm()).to(dev))
    else:
        raise ValueError("Missing trial state: -instate=None and -ipeps_init_type= " + str(args.ipeps_init_type) + " is not supported")
    print(state)

    @torch.no_grad()
    def ctmrg_conv_f(state, ctm_env, history, ctm_args=cfg.ctm_args):
        """distance of successive rho_2x1 (:72-86)"""
        if not history:
            history = dict({"log": []})
        rdm2x1 = rdm2x1_sl(state, ctm_env)
        dist = float('inf')
        if len(history["log"]) > 0:
            dist = torch.dist(rdm2x1, history["rdm"], p=2).item()
        history["rdm"] = rdm2x1
        history["log"].append(dist)
        if dist < ctm_args.ctm_conv_tol or len(history["log"]) >= ctm_args.ctm_max_iter:
            return True, history
        return False, history

    state_sym = to_ipeps_c4v(state)
    ctm_env = ENV_C4V(args.chi, state_sym)
    init_env(state_sym, ctm_env)
    ctm_env, *ctm_log = ctmrg_c4v.run(state_sym, ctm_env, conv_check=ctmrg_conv_f)
    loss = energy_f(state_sym, ctm_env)
    obs_values, obs_labels = model.eval_obs(state_sym, ctm_env)
    print(", ".join(["epoch", "energy"] + obs_labels))
    print(", ".join([f"{-1}", f"{loss}"] + [f"{v}" for v in obs_values]))

    def loss_fn(state, ctm_env, opt_context):
        """symmetrise + normalise the parameters (tracked), environment by CTMRG (re-initialised if opt_ctm_reinit), energy (:99-120)"""
        ctm_args, opt_args = opt_context["ctm_args"], opt_context["opt_args"]
        state_sym = to_ipeps_c4v(state, normalize=True)
        if opt_args.opt_ctm_reinit:
            init_env(state_sym, ctm_env)
        ctm_env, *ctm_log = ctmrg_c4v.run(state_sym, ctm_env, conv_check=ctmrg_conv_f, ctm_args=ctm_args)
        loss = energy_f(state_sym, ctm_env)
        return (loss, ctm_env, *ctm_log)

    def _to_json(l):
        return dict({"re": [l[i, 0].item() for i in range(l.size()[0])], "im": [l[i, 1].item() for i in range(l.size()[0])]})

    @torch.no_grad()
    def obs_fn(state, ctm_env, opt_context):
        if opt_context["line_search"]:
            epoch = len(opt_context["loss_history"]["loss_ls"])
            loss = opt_context["loss_history"]["loss_ls"][-1]
            print("LS", end=" ")
        else:
            epoch = len(opt_context["loss_history"]["loss"])
            loss = opt_context["loss_history"]["loss"][-1]
        state_sym = to_ipeps_c4v(state, normalize=True)
        obs_values, obs_labels = model.eval_obs(state_sym, ctm_env)
        print(", ".join([f"{epoch}", f"{loss}"] + [f"{v}" for v in obs_values] + [f"{torch.max(torch.abs(state.site((0, 0))))}"]))
        if (not opt_context["line_search"]) and args.top_freq > 0 and epoch % args.top_freq == 0:
            print(f"TOP spectrum(T)[{(0, 0)},{(1, 0)}] ", end="")
            l = transferops_c4v.get_Top_spec_c4v(args.top_n, state_sym, ctm_env)
            print("TOP " + json.dumps(_to_json(l)))

    optimize_state(state, ctm_env, loss_fn, obs_fn=obs_fn)

    # final observables of the best variational state
    outputstatefile = args.out_prefix + "_state.json"
    state = read_ipeps_c4v(outputstatefile)
    ctm_env = ENV_C4V(args.chi, state)
    init_env(state, ctm_env)
    ctm_env, *ctm_log = ctmrg_c4v.run(state, ctm_env, conv_check=ctmrg_conv_f)
    opt_energy = energy_f(state, ctm_env)
    obs_values, obs_labels = model.eval_obs(state, ctm_env)
    print(", ".join([f"{args.opt_max_iter}", f"{opt_energy}"] + [f"{v}" for v in obs_values]))
    return float(opt_energy)


if __name__ == '__main__':
    main()
