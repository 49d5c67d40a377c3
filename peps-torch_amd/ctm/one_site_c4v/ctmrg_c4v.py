"""One-site C4v CTMRG (reference ctm/one_site_c4v/ctmrg_c4v.py:16-108,182-197,325-463).

A sweep is ONE native call: enlarged corner -> truncated symmetric eigendecomposition -> C = diag(D),
T = P.T.a.a*.P* symmetrised -> normalisation; nothing leaves HBM between the steps."""
import time
import logging
import torch
import config as cfg
from backend import get_engine

log = logging.getLogger(__name__)


def run(state, env, conv_check=None, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
    if ctm_args.projector_svd_method not in ['DEFAULT', 'SYMEIG']:
        raise Exception(f"Projector eig/svd method \"{ctm_args.projector_svd_method}\" not implemented")
    a = next(iter(state.sites.values()))
    eng = get_engine()
    t_obs = t_ctm = 0.
    history = None
    for i in range(ctm_args.ctm_max_iter):
        t0 = time.perf_counter()
        ctm_MOVE_sl(a, env, None, ctm_args=ctm_args, global_args=global_args)
        eng.sync()
        t1 = time.perf_counter()
        t_ctm += t1 - t0
        if conv_check is not None:
            converged, history = conv_check(state, env, history, ctm_args=ctm_args)
            t_obs += time.perf_counter() - t1
            if converged:
                if ctm_args.verbosity_ctm_convergence > 0:
                    print(f"CTMRG converged at iter= {i}")
                break
    return env, history, t_ctm, t_obs


def ctm_MOVE_sl(a, env, f_c2x2_decomp=None, ctm_args=cfg.ctm_args, global_args=cfg.global_args, past_steps_data=None):
    """One C4v move.  `f_c2x2_decomp` is accepted for signature compatibility; the native move always
    uses the truncated symmetric eigendecomposition with keep_multiplets (ctmrg_c4v.py:49-52)."""
    norm_kind = 1 if ctm_args.ctm_absorb_normalization == 'inf' else 2        # anything else is the 2-norm (ctmrg_c4v.py:183-185)
    eng = get_engine()
    cfgT = eng.cfg(eps_multiplet=1.0e-12, multiplet_abstol=1.0e-14, keep_multiplets=True)
    basis = None
    if getattr(ctm_args, "projector_warm_start", True) and hasattr(eng, "warm_basis_c4v"):
        # the environment remembers the invariant subspace of the previous enlarged corner (residual-verified restart)
        n = env.chi * a.shape[1] ** 2
        basis = env.__dict__.get("_warm")
        k = env.chi + 1 if env.chi < n else n
        if basis is None or tuple(basis.shape) != (min(n, k + 8), n) or basis.device != a.device:
            basis = env.__dict__["_warm"] = eng.warm_basis_c4v(env.chi, n)
    nC, nT, _D = eng.move_c4v(a, env.C[env.keyC], env.T[env.keyT], cfgT, normalize=norm_kind, **({"basis": basis} if basis is not None else {}))
    env.C[env.keyC] = nC
    env.T[env.keyT] = nT


ctm_MOVE_dl = ctm_MOVE_sl
run_dl = run
