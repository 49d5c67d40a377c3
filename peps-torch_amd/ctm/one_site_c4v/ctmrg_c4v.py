"""One-site C4v CTMRG (reference ctm/one_site_c4v/ctmrg_c4v.py:16-108,182-197,325-463).

A sweep is ONE native call: enlarged corner -> truncated symmetric eigendecomposition -> C = diag(D),
T = P.T.a.a*.P* symmetrised -> normalisation; nothing leaves HBM between the steps."""
import time
import logging
import torch
import config as cfg
from backend import get_engine
from linalg.native_einsum import einsum, needs_grad

log = logging.getLogger(__name__)


def run(state, env, conv_check=None, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
    # SYMARP / SYMLOBPCG (ctmrg_c4v.py:53-60: ARPACK / LOBPCG for the leading chi eigenpairs, forward only) name what the native
    # truncation does anyway -- the leading-chi solver; they are accepted for runs that carry no gradient (e.g. the line search
    # of the optimiser, OPTARGS.line_search_svd_method)
    if ctm_args.projector_svd_method not in ['DEFAULT', 'SYMEIG', 'SYMARP', 'SYMLOBPCG']:
        raise Exception(f"Projector eig/svd method \"{ctm_args.projector_svd_method}\" not implemented")
    a = next(iter(state.sites.values()))
    if ctm_args.projector_svd_method in ['SYMARP', 'SYMLOBPCG'] and needs_grad(a, env.C[env.keyC], env.T[env.keyT]):
        raise NotImplementedError(f"projector_svd_method {ctm_args.projector_svd_method} is forward only; use SYMEIG for a differentiable run")
    eng = get_engine()
    t_obs = t_ctm = 0.
    history = None
    for i in range(ctm_args.ctm_max_iter):
        t0 = time.perf_counter()
        env.__dict__["_move_index"] = i          # the differentiable route keeps one eigenbasis per move of a run (see _ctm_MOVE_sl_ad)
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
    if needs_grad(a, env.C[env.keyC], env.T[env.keyT]):
        return _ctm_MOVE_sl_ad(a, env, f_c2x2_decomp, norm_kind, ctm_args)
    eng = get_engine()
    cfgT = eng.cfg(eps_multiplet=1.0e-12, multiplet_abstol=1.0e-14, keep_multiplets=True)
    basis = None
    if getattr(ctm_args, "projector_warm_start", True) and hasattr(eng, "warm_basis_c4v"):
        # the environment remembers the invariant subspace of the previous enlarged corner (residual-verified restart)
        n = env.chi * a.shape[1] ** 2
        basis = env.__dict__.get("_warm")
        k = env.chi + 1 if env.chi < n else n
        if basis is None or tuple(basis.shape) != ((2 if a.is_complex() else 1) * min(n, k + 8) + 1, n) or basis.device != a.device:
            basis = env.__dict__["_warm"] = eng.warm_basis_c4v(env.chi, n, a.dtype)
    nC, nT, _D = eng.move_c4v(a, env.C[env.keyC], env.T[env.keyT], cfgT, normalize=norm_kind, **({"basis": basis} if basis is not None else {}))
    env.C[env.keyC] = nC
    env.T[env.keyT] = nT


def _ctm_MOVE_sl_ad(a, env, f_c2x2_decomp, norm_kind, ctm_args):
    """The same move as a graph of differentiable native nodes (reference ctmrg_c4v.py:349-463 under torch autograd): enlarged
    corner (one contraction node) -> FULL symmetric eigendecomposition with the regularised backward, truncated by slicing
    (custom_eig.py:7-65) -> C = diag(D), T = P.T.a.a*.P* (one contraction node) symmetrised -> scales taken without gradient
    (`_move_normalize_c`, :182-197).  `fwd_checkpoint_move` recomputes the move in the backward pass like the reference."""
    from ctm.one_site_c4v.ctm_components_c4v import c2x2_sl
    from linalg.custom_eig import truncated_eig_sym
    if f_c2x2_decomp is None:
        # An optimisation evaluates the same sequence of moves again and again on slowly changing tensors: move i of this run sees
        # almost the matrix move i of the previous run saw.  The environment therefore keeps the eigenvectors of every move of a
        # run (n x n each, handed on by ENV_C4V.detach()), and the full decomposition of the next evaluation starts from them.
        basis = None
        idx = env.__dict__.get("_move_index")
        eng = get_engine()
        if idx is not None and getattr(ctm_args, "projector_warm_start", True) and hasattr(eng, "warm_basis_c4v") and a.is_cuda:
            n = env.chi * a.shape[1] ** 2
            ws = env.__dict__.setdefault("_warm_ad", {})
            basis = ws.get(idx)
            if basis is None or tuple(basis.shape) != ((2 if a.is_complex() else 1) * n + 1, n) or basis.device != a.device:
                if len(ws) * n * n * 8 * (2 if a.is_complex() else 1) < 0.05 * torch.cuda.get_device_properties(a.device).total_memory:
                    basis = ws[idx] = eng.warm_basis_c4v(n, n, a.dtype)
                else:
                    basis = None

        def f_c2x2_decomp(M, chi):
            return truncated_eig_sym(M, chi, keep_multiplets=True, eps_multiplet=1.0e-12, abs_tol=1.0e-14,
                                     ad_decomp_reg=ctm_args.ad_decomp_reg, basis=basis)
    chi = env.chi

    def core(a, C, T):
        D_ = a.shape[1]
        C2X2 = c2x2_sl(a, C, T)
        Dv, P = f_c2x2_decomp(C2X2, chi)
        nC = torch.diag(Dv.to(a.dtype))
        Pv = P.reshape(chi, D_, D_, chi)
        Tv = T.reshape(chi, chi, D_, D_)
        nT = einsum('xuUi,xelL,suldr,sULDR,edDj->ijrR', Pv, Tv, a, a, Pv, conj=(3, 4)).reshape(chi, chi, D_ * D_)
        nT = 0.5 * (nT + nT.conj().permute(1, 0, 2))
        with torch.no_grad():
            sC = nC[0, 0].abs()
            sT = nT.abs().max() if norm_kind == 1 else torch.linalg.vector_norm(nT)
        return nC / sC, nT / sT

    tensors = (a, env.C[env.keyC], env.T[env.keyT])
    if getattr(ctm_args, "fwd_checkpoint_move", False):
        from torch.utils.checkpoint import checkpoint
        nC, nT = checkpoint(core, *tensors, use_reentrant=False)
    else:
        nC, nT = core(*tensors)
    env.C[env.keyC] = nC
    env.T[env.keyT] = nT


ctm_MOVE_dl = ctm_MOVE_sl
run_dl = run
