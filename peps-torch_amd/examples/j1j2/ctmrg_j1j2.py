"""CTMRG for the J1-J2 model on a generic unit cell -- same flags, tilings, convergence check and
`FINAL` line as the reference script (examples/j1j2/ctmrg_j1j2.py:14-204), running on the MI355X engine.

    python examples/j1j2/ctmrg_j1j2.py --tiling 4SITE --bond_dim 4 --chi 64 --j2 0.5
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import config as cfg
from ipeps.ipeps import IPEPS, read_ipeps, extend_bond_dim
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
from models import j1j2

parser = cfg.get_args_parser()
parser.add_argument("--j1", type=float, default=1.)
parser.add_argument("--j2", type=float, default=0.)
parser.add_argument("--j3", type=float, default=0.)
parser.add_argument("--lmbd", type=float, default=0.)
parser.add_argument("--hz_stag", type=float, default=0.)
parser.add_argument("--h_uni", nargs=3, type=float, default=[0, 0, 0], help="uniform field, components h^z, h^x, h^y")
parser.add_argument("--delta_zz", type=float, default=1.)
parser.add_argument("--tiling", default="BIPARTITE", help="BIPARTITE, 1SITE, 2SITE, 4SITE, 8SITE")
parser.add_argument("--top_freq", type=int, default=-1)
parser.add_argument("--corrf_r", type=int, default=1, help="maximal correlation function distance")
parser.add_argument("--top_n", type=int, default=2, help="number of leading eigenvalues of the transfer operator to compute")

TILINGS = {
    "BIPARTITE": lambda c: ((((c[0] + abs(c[0]) * 2) % 2) + abs(c[1])) % 2, 0),
    "1SITE": lambda c: (0, 0),
    "2SITE": lambda c: ((c[0] + abs(c[0]) * 2) % 2, 0),
    "4SITE": lambda c: ((c[0] + abs(c[0]) * 2) % 2, (c[1] + abs(c[1]) * 2) % 2),
    "8SITE": lambda c: ((c[0] + 2 * (c[1] // 2)) % 4, c[1] % 2),          # 4x2 cell with a shift of 2 every second row
}
NSITES = {"BIPARTITE": [(0, 0), (1, 0)], "1SITE": [(0, 0)], "2SITE": [(0, 0), (1, 0)],
          "4SITE": [(0, 0), (1, 0), (0, 1), (1, 1)],
          "8SITE": [(0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (1, 1), (2, 1), (3, 1)]}


def main(args=None):
    args, _ = parser.parse_known_args(args)
    cfg.configure(args)
    torch.set_num_threads(args.omp_cores)
    torch.manual_seed(args.seed)
    if args.tiling not in TILINGS:
        raise ValueError("Invalid tiling: " + str(args.tiling))
    model = j1j2.J1J2(j1=args.j1, j2=args.j2, j3=args.j3, lmbd=args.lmbd, hz_stag=args.hz_stag, delta_zz=args.delta_zz, h_uni=args.h_uni)
    lattice_to_site = TILINGS[args.tiling]
    energy_f = model.energy_2x2_1site_BP if args.tiling == "1SITE" else model.energy_per_site
    eval_obs_f = model.eval_obs_1site_BP if args.tiling == "1SITE" else model.eval_obs

    dev, dt = cfg.global_args.device, cfg.global_args.torch_dtype
    if args.instate is not None:
        state = read_ipeps(args.instate, vertexToSite=lattice_to_site)
        if args.bond_dim > max(state.get_aux_bond_dims()):
            state = extend_bond_dim(state, args.bond_dim)
        state.add_noise(args.instate_noise)
    elif args.ipeps_init_type == 'RANDOM':
        D = args.bond_dim
        sites = {}
        for c in NSITES[args.tiling]:
            A = torch.rand((model.phys_dim, D, D, D, D), dtype=dt, device='cpu') - 0.5
            sites[c] = (A / torch.max(torch.abs(A))).to(dev)
        state = IPEPS(sites, vertexToSite=lattice_to_site)
    else:
        raise ValueError("Missing trial state: --instate=None and --ipeps_init_type= " + str(args.ipeps_init_type) + " is not supported")
    print(state)

    def ctmrg_conv_energy(state, env, history, ctm_args=cfg.ctm_args):
        if not history:
            history = []
        e_curr = energy_f(state, env)
        history.append(e_curr.item())
        if (len(history) > 1 and abs(history[-1] - history[-2]) < ctm_args.ctm_conv_tol) or len(history) >= ctm_args.ctm_max_iter:
            return True, history
        return False, history

    env = ENV(args.chi, state)
    init_env(state, env)
    print(env)
    e0 = energy_f(state, env)
    obs_values, obs_labels = eval_obs_f(state, env)
    print(", ".join(["epoch", "energy"] + obs_labels))
    print(", ".join([f"{-1}", f"{e0}"] + [f"{v}" for v in obs_values]))
    env, history, t_ctm, t_obs = ctmrg.run(state, env, conv_check=ctmrg_conv_energy)
    e = energy_f(state, env)
    obs_values, obs_labels = eval_obs_f(state, env)
    print(", ".join([f"{len(history)}", f"{e}"] + [f"{v}" for v in obs_values]))
    print(f"TIMINGS ctm: {t_ctm} conv_check: {t_obs}")
    print("FINAL " + ", ".join([f"{e}"] + [f"{v}" for v in obs_values]))

    # additional observables, as the reference script prints them after FINAL (examples/j1j2/ctmrg_j1j2.py:176-201)
    from ctm.generic import transferops
    from backend import get_engine
    for direction in ((1, 0), (0, 1)):
        corrSS = model.eval_corrf_SS((0, 0), direction, state, env, args.corrf_r)
        print(f"\n\nSS[(0,0),{direction}] r " + " ".join(corrSS.keys()))
        for i in range(args.corrf_r):
            print(f"{i} " + " ".join([f"{corrSS[label][i]}" for label in corrSS.keys()]))
    print("\n")
    for c_loc, c_ten in env.C.items():
        s = get_engine().svdvals(c_ten)
        print(f"spectrum C[{c_loc}]")
        for i in range(args.chi):
            print(f"{i} {s[i]}")
    for sdp in (((0, 0), (1, 0)), ((0, 0), (0, 1))):
        print(f"\n\nspectrum(T)[{sdp[0]},{sdp[1]}]")
        l = transferops.get_Top_spec(args.top_n, *sdp, state, env)
        for i in range(l.size()[0]):
            print(f"{i} {l[i, 0]} {l[i, 1]}")
    return float(e), obs_values


if __name__ == '__main__':
    main()
