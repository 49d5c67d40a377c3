"""CTMRG for a single-site C4v iPEPS of the J1-J2 model -- same flags and `FINAL` line as the reference
script (examples/j1j2/ctmrg_j1j2_c4v.py:14-196), running on the MI355X engine.

    python examples/j1j2/ctmrg_j1j2_c4v.py --instate RVB_1x1.in --chi 16 --j2 0.5 --CTMARGS_ctm_max_iter 200
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import config as cfg
from ipeps.ipeps_c4v import IPEPS_C4V, read_ipeps_c4v, extend_bond_dim
from groups.pg import make_c4v_symm
from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
from ctm.one_site_c4v import ctmrg_c4v
from ctm.one_site_c4v.rdm_c4v import rdm2x1_sl
from models import j1j2

parser = cfg.get_args_parser()
parser.add_argument("--j1", type=float, default=1.)
parser.add_argument("--j2", type=float, default=0.)
parser.add_argument("--hz_stag", type=float, default=0.)
parser.add_argument("--delta_zz", type=float, default=1.)
parser.add_argument("--top_freq", type=int, default=-1)
parser.add_argument("--corrf_canonical", action='store_true', help="align spin operators with the vector of spontaneous magnetization")
parser.add_argument("--corrf_r", type=int, default=1, help="maximal correlation function distance")
parser.add_argument("--top_n", type=int, default=2, help="number of leading eigenvalues of the transfer operator to compute")
parser.add_argument("--corrf_dd_v", action='store_true', help="compute vertical dimer-dimer correlation function")
parser.add_argument("--top2", action='store_true', help="compute transfer matrix for width-2 channel")


def main(args=None):
    args, _ = parser.parse_known_args(args)
    cfg.configure(args)
    torch.set_num_threads(args.omp_cores)
    torch.manual_seed(args.seed)
    model = j1j2.J1J2_C4V_BIPARTITE(j1=args.j1, j2=args.j2, hz_stag=args.hz_stag, delta_zz=args.delta_zz)
    energy_f = model.energy_1x1_lowmem
    dev, dt = cfg.global_args.device, cfg.global_args.torch_dtype
    if args.instate is not None:
        state = read_ipeps_c4v(args.instate)
        if args.bond_dim > max(state.get_aux_bond_dims()):        # --bond_dim below the file's D: no extension
            state = extend_bond_dim(state, args.bond_dim)
        state.add_noise(args.instate_noise)
        state.sites[(0, 0)] = state.sites[(0, 0)] / torch.max(torch.abs(state.sites[(0, 0)]))
    elif args.ipeps_init_type == 'RANDOM':
        D = args.bond_dim
        A = torch.rand((model.phys_dim, D, D, D, D), dtype=dt, device='cpu')
        A = make_c4v_symm(A)
        state = IPEPS_C4V((A / torch.max(torch.abs(A))).to(dev))
    else:
        raise ValueError("Missing trial state: --instate=None and --ipeps_init_type= " + str(args.ipeps_init_type) + " is not supported")
    print(state)

    def ctmrg_conv_rdm2x1(state, env, history, ctm_args=cfg.ctm_args):
        """distance of successive rho_2x1 (reference examples/j1j2/ctmrg_j1j2_c4v.py:97-125)"""
        if not history:
            history = dict({"log": []})
        rdm = rdm2x1_sl(state, env).cpu()
        dist = float('inf')
        if len(history["log"]) > 0:
            dist = torch.dist(rdm, history["rdm"], p=2).item()
        history["rdm"] = rdm
        history["log"].append(dist)
        if dist < ctm_args.ctm_conv_tol or len(history["log"]) >= ctm_args.ctm_max_iter:
            return True, history
        return False, history

    env = ENV_C4V(args.chi, state)
    init_env(state, env)
    e0 = energy_f(state, env)
    obs_values, obs_labels = model.eval_obs(state, env)
    print(", ".join(["epoch", "energy"] + obs_labels))
    print(", ".join([f"{-1}", f"{e0}"] + [f"{v}" for v in obs_values]))
    env, history, t_ctm, t_obs = ctmrg_c4v.run(state, env, conv_check=ctmrg_conv_rdm2x1)
    e = energy_f(state, env)
    obs_values, obs_labels = model.eval_obs(state, env)
    print(", ".join([f"{len(history['log'])}", f"{e}"] + [f"{v}" for v in obs_values]))
    print(f"TIMINGS ctm: {t_ctm} conv_check: {t_obs}")
    print("FINAL " + ", ".join([f"{e}"] + [f"{v}" for v in obs_values]))

    # additional observables, as the reference script prints them after FINAL (examples/j1j2/ctmrg_j1j2_c4v.py:153-183)
    corrSS = model.eval_corrf_SS(state, env, args.corrf_r, canonical=args.corrf_canonical)
    print("\n\nSS r " + " ".join(corrSS.keys()) + f" canonical {args.corrf_canonical}")
    for i in range(args.corrf_r):
        print(f"{i} " + " ".join([f"{corrSS[label][i]}" for label in corrSS.keys()]))
    corrDD = model.eval_corrf_DD_H(state, env, args.corrf_r)
    print("\n\nDD r " + " ".join(corrDD.keys()))
    for i in range(args.corrf_r):
        print(f"{i} " + " ".join([f"{corrDD[label][i]}" for label in corrDD.keys()]))
    if args.corrf_dd_v:
        corrDD_V = model.eval_corrf_DD_V(state, env, args.corrf_r)
        print("\n\nDD_v r " + " ".join(corrDD_V.keys()))
        for i in range(args.corrf_r):
            print(f"{i} " + " ".join([f"{corrDD_V[label][i]}" for label in corrDD_V.keys()]))
    print("\n\nspectrum(C)")
    s = get_engine_svdvals(env.C[env.keyC])
    for i in range(args.chi):
        print(f"{i} {s[i]}")
    print("\n\nspectrum(T)")
    from ctm.one_site_c4v import transferops_c4v
    l = transferops_c4v.get_Top_spec_c4v(args.top_n, state, env)
    for i in range(l.size()[0]):
        print(f"{i} {l[i, 0]} {l[i, 1]}")
    if args.top2:
        print("\n\nspectrum(T2)")
        l = transferops_c4v.get_Top2_spec_c4v(args.top_n, state, env)
        for i in range(l.size()[0]):
            print(f"{i} {l[i, 0]} {l[i, 1]}")
    return float(e), obs_values


def get_engine_svdvals(C):
    from backend import get_engine
    return get_engine().svdvals(C)


if __name__ == '__main__':
    main()
