"""fixture unpacking without torch/GPU dependencies"""


def sites_from(g):
    return {tuple(int(v) for v in k.split('_')[1:]): g[k] for k in g.files if k.startswith('site_')}


def env_from(g, prefix):
    C, T = {}, {}
    for k in g.files:
        if k.startswith(prefix + 'C_') or k.startswith(prefix + 'T_'):
            x, y, vx, vy = (int(v) for v in k[len(prefix) + 2:].split('_'))
            (C if k[len(prefix)] == 'C' else T)[((x, y), (vx, vy))] = g[k]
    return C, T


def run_c4v_optimizer(g, tmpdir, device="cpu", line_search="default", epochs=None):
    """The loss function of examples/j1j2/optim_j1j2_c4v.py (symmetrise + normalise -> init_env -> a fixed number of CTM moves ->
    energy_1x1_lowmem) under optim.ad_optim_lbfgs_mod.optimize_state, from the start tensor of the golden trajectory `g`
    (oracle/gen_golden.py c4v_optim_case).  Returns (losses per epoch, final parameters, best state read back from the file)."""
    import copy, os
    import torch
    import config as cfg
    from ipeps.ipeps_c4v import IPEPS_C4V, to_ipeps_c4v, read_ipeps_c4v
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    from ctm.one_site_c4v import ctmrg_c4v
    from ctm.one_site_c4v.rdm_c4v import rdm2x1_sl
    from models import j1j2
    from optim.ad_optim_lbfgs_mod import optimize_state
    st = IPEPS_C4V(torch.from_numpy(g["site0"].copy()).to(device))
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=float(g["j2"]))
    ctm_args = copy.deepcopy(cfg.ctm_args); ctm_args.ctm_max_iter = int(g["ctm_iter"]); ctm_args.ctm_conv_tol = -1.0
    opt_args = copy.deepcopy(cfg.opt_args); opt_args.line_search = line_search; opt_args.opt_logging = False
    main_args = copy.deepcopy(cfg.main_args); main_args.opt_max_iter = int(g["epochs"]) if epochs is None else epochs
    main_args.out_prefix = os.path.join(str(tmpdir), "o"); main_args.opt_resume = None

    @torch.no_grad()
    def conv_f(state, env, history, ctm_args=ctm_args):
        if not history:
            history = dict({"log": []})
        r = rdm2x1_sl(state, env)
        dist = float('inf')
        if len(history["log"]) > 0:
            dist = torch.dist(r, history["rdm"], p=2).item()
        history["rdm"] = r; history["log"].append(dist)
        return (dist < ctm_args.ctm_conv_tol or len(history["log"]) >= ctm_args.ctm_max_iter), history

    def loss_fn(state, env, ctx):
        ss = to_ipeps_c4v(state, normalize=True)
        if ctx["opt_args"].opt_ctm_reinit:
            init_env(ss, env)
        env, *log_ = ctmrg_c4v.run(ss, env, conv_check=conv_f, ctm_args=ctx["ctm_args"])
        return (model.energy_1x1_lowmem(ss, env), env, *log_)
    hist = {}

    def post(state, env, ctx):
        hist["loss"] = list(ctx["loss_history"]["loss"])
    env = ENV_C4V(int(g["chi"]), to_ipeps_c4v(st))
    init_env(to_ipeps_c4v(st), env)
    optimize_state(st, env, loss_fn, post_proc=post, main_args=main_args, opt_args=opt_args, ctm_args=ctm_args)
    best = read_ipeps_c4v(main_args.out_prefix + "_state.json")
    return hist["loss"], st.site().detach().cpu().numpy(), best.site().cpu().numpy()


def run_generic_optimizer(g, tmpdir, device="cpu"):
    """The loss function of examples/j1j2/optim_j1j2.py (init_env -> a fixed number of CTM iterations -> energy_2x2_4site) under
    optim.ad_optim_lbfgs_mod.optimize_state, from the start tensors of the golden trajectory `g` (oracle/gen_golden.py
    generic_optim_case).  Returns (losses per epoch, final parameters by site)."""
    import copy, os
    import torch
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from models import j1j2
    from optim.ad_optim_lbfgs_mod import optimize_state
    sites = {tuple(int(v) for v in k.split('_')[1:]): torch.from_numpy(g[k].copy()).to(device) for k in g.files if k.startswith('site0_')}
    st = IPEPS(sites, lX=max(k[0] for k in sites) + 1, lY=max(k[1] for k in sites) + 1)
    model = j1j2.J1J2(j1=1.0, j2=float(g["j2"]))
    ctm_args = copy.deepcopy(cfg.ctm_args); ctm_args.ctm_max_iter = int(g["ctm_iter"])
    opt_args = copy.deepcopy(cfg.opt_args); opt_args.opt_logging = False
    main_args = copy.deepcopy(cfg.main_args); main_args.opt_max_iter = int(g["epochs"])
    os.makedirs(str(tmpdir), exist_ok=True)
    main_args.out_prefix = os.path.join(str(tmpdir), "o"); main_args.opt_resume = None

    @torch.no_grad()
    def conv_f(state, env, history, ctm_args=ctm_args):
        history = (history or []) + [0.]
        return len(history) >= ctm_args.ctm_max_iter, history

    def loss_fn(state, env, ctx):
        if ctx["opt_args"].opt_ctm_reinit:
            init_env(state, env)
        env_out, *log_ = ctmrg.run(state, env, conv_check=conv_f, ctm_args=ctx["ctm_args"])
        return (model.energy_2x2_4site(state, env), env, *log_)
    hist = {}

    def post(state, env, ctx):
        hist["loss"] = list(ctx["loss_history"]["loss"])
    env = ENV(int(g["chi"]), st)
    init_env(st, env)
    optimize_state(st, env, loss_fn, post_proc=post, main_args=main_args, opt_args=opt_args, ctm_args=ctm_args)
    return hist["loss"], {c: t.detach().cpu().numpy() for c, t in st.sites.items()}
