"""Gradient-based optimisation of an iPEPS with L-BFGS: the caller of the differentiable CTM path.

Interface of the reference's optim/ad_optim_lbfgs_mod.py:19-360 (`optimize_state`, `store_checkpoint`, the `opt_context` dictionary
handed to `loss_fn` / `obs_fn` / `post_proc`, best state -> `<out_prefix>_state.json`, checkpoint -> `<out_prefix>_checkpoint.p`).
Here the optimiser is torch's own `torch.optim.LBFGS`: with `OPTARGS.line_search = "default"` (fixed step `lr`, the reference's
default) its iterates are those of the reference's `LBFGS_MOD.step_2c` (lbfgs_modified.py:154-400: same two-loop recursion,
same first step min(1, 1/|g|_1) lr, same stopping rules); "strong_wolfe" is torch's line search (the reference's is a copy of it
that also keeps the last evaluated gradient for the next epoch: one closure evaluation less per epoch, same points);
"backtracking" is a derivative-free Armijo search on the loss evaluated without autograd (lbfgs_modified.py:13-82,312-328)."""
import copy
import json
import logging
import time
import torch
import config as cfg
import parallel

log = logging.getLogger(__name__)


def store_checkpoint(checkpoint_file, state, optimizer, current_epoch, current_loss, verbosity=0):
    if parallel.world()[0] != 0:            # replicated run: one writer
        return
    torch.save({'epoch': current_epoch, 'loss': current_loss, 'parameters': state.get_checkpoint(),
                'optimizer_state_dict': optimizer.state_dict()}, checkpoint_file)
    if verbosity > 0:
        print(checkpoint_file)


def create_optimizer(state, main_args=cfg.main_args, opt_args=cfg.opt_args, ctm_args=cfg.ctm_args, global_args=cfg.global_args):
    parameters = list(state.get_parameters())
    for A in parameters:
        A.requires_grad_(True)
    ls = opt_args.line_search
    if ls not in (None, "default", "strong_wolfe", "backtracking"):
        raise RuntimeError("unsupported line search")
    optimizer = torch.optim.LBFGS(parameters, max_iter=opt_args.max_iter_per_epoch, lr=opt_args.lr,
                                  tolerance_grad=opt_args.tolerance_grad, tolerance_change=opt_args.tolerance_change,
                                  history_size=opt_args.history_size, line_search_fn="strong_wolfe" if ls == "strong_wolfe" else None)
    optimizer.zero_grad()
    return parameters, optimizer


def load_optimizer_state_(optimizer, state, main_args=cfg.main_args, opt_args=cfg.opt_args, ctm_args=cfg.ctm_args,
                          global_args=cfg.global_args):
    """Resume from `main_args.opt_resume`; with `opt_resume_override_params` the stored step size, tolerances and history size
    are replaced by the current OPTARGS (the L-BFGS memory is shortened to the new history size)."""
    print(f"INFO: resuming from check point. resume = {main_args.opt_resume}")
    checkpoint = torch.load(main_args.opt_resume, map_location=state.device, weights_only=False)
    sd = checkpoint["optimizer_state_dict"]
    group = sd["param_groups"][0]
    hist = sd["state"][group["params"][0]]
    if main_args.opt_resume_override_params:
        group.update(lr=opt_args.lr, max_iter=opt_args.max_iter_per_epoch, tolerance_grad=opt_args.tolerance_grad,
                     tolerance_change=opt_args.tolerance_change,
                     line_search_fn="strong_wolfe" if opt_args.line_search == "strong_wolfe" else None)
        if opt_args.history_size < group["history_size"]:
            for k in ("old_dirs", "old_stps", "ro"):
                if hist.get(k) is not None:
                    hist[k] = hist[k][-opt_args.history_size:]
        if hist.get("al") is not None:
            al = [x for x in hist["al"] if x is not None][-opt_args.history_size:]
            hist["al"] = al + [None] * (opt_args.history_size - len(al))
        group["history_size"] = opt_args.history_size
    optimizer.load_state_dict(sd)
    print(f"checkpoint.loss = {checkpoint['loss']}")


def _armijo(phi, phi0, derphi0, c1=1e-4, alpha0=1.0, amin=1.0e-8):
    """Backtracking with quadratic, then cubic interpolation until phi(alpha) <= phi0 + c1 alpha phi'(0) (Nocedal & Wright 3.5)."""
    phi_a0 = phi(alpha0)
    if phi_a0 <= phi0 + c1 * alpha0 * derphi0:
        return alpha0, phi_a0
    alpha1 = -derphi0 * alpha0 ** 2 / 2.0 / (phi_a0 - phi0 - derphi0 * alpha0)
    phi_a1 = phi(alpha1)
    if phi_a1 <= phi0 + c1 * alpha1 * derphi0:
        return alpha1, phi_a1
    while alpha1 > amin:
        f = alpha0 ** 2 * alpha1 ** 2 * (alpha1 - alpha0)
        a = (alpha0 ** 2 * (phi_a1 - phi0 - derphi0 * alpha1) - alpha1 ** 2 * (phi_a0 - phi0 - derphi0 * alpha0)) / f
        b = (-alpha0 ** 3 * (phi_a1 - phi0 - derphi0 * alpha1) + alpha1 ** 3 * (phi_a0 - phi0 - derphi0 * alpha0)) / f
        alpha2 = (-b + max(b * b - 3 * a * derphi0, 0.0) ** 0.5) / (3.0 * a)
        if (alpha1 - alpha2) > alpha1 / 2.0 or (1 - alpha2 / alpha1) < 0.96:
            alpha2 = alpha1 / 2.0
        phi_a2 = phi(alpha2)
        if phi_a2 <= phi0 + c1 * alpha2 * derphi0:
            return alpha2, phi_a2
        alpha0, alpha1, phi_a0, phi_a1 = alpha1, alpha2, phi_a1, phi_a2
    return None, phi_a1


def optimize_state(state, ctm_env_init, loss_fn, obs_fn=None, post_proc=None, main_args=cfg.main_args, opt_args=cfg.opt_args,
                   ctm_args=cfg.ctm_args, global_args=cfg.global_args):
    """Minimise `loss_fn(state, env, opt_context) -> (loss, env, history, t_ctm, t_obs[, t_loss])` over the on-site tensors of
    `state`.  Every evaluation starts from the environment the previous one returned (detached); the lowest-loss state is written
    to `<out_prefix>_state.json`, a checkpoint before every step to `<out_prefix>_checkpoint.p`."""
    verbosity = opt_args.verbosity_opt_epoch
    checkpoint_file = main_args.out_prefix + "_checkpoint.p"
    outputstatefile = main_args.out_prefix + "_state.json"
    t_data = dict({"loss": [], "min_loss": 1.0e+16, "loss_ls": [], "min_loss_ls": 1.0e+16})
    current_env = [ctm_env_init]
    context = dict({"ctm_args": ctm_args, "opt_args": opt_args, "loss_history": t_data})
    epoch = 0
    parameters, optimizer = create_optimizer(state, main_args=main_args, opt_args=opt_args, ctm_args=ctm_args, global_args=global_args)
    if main_args.opt_resume is not None:
        load_optimizer_state_(optimizer, state, main_args=main_args, opt_args=opt_args, ctm_args=ctm_args, global_args=global_args)
    calls = [0]

    def _write_best():
        if parallel.world()[0] == 0:        # replicated run: one writer
            state.write_to_file(outputstatefile, normalize=True)

    def _record(loss, linesearching):
        if linesearching:
            t_data["loss_ls"].append(loss)
            if t_data["min_loss_ls"] > loss:
                t_data["min_loss_ls"] = loss
                if t_data["min_loss"] > loss:
                    _write_best()
        else:
            t_data["loss"].append(loss)
            if t_data["min_loss"] > loss:
                t_data["min_loss"] = loss
                _write_best()

    def closure():
        # torch's strong-Wolfe search re-enters the closure: the first call of a step is the epoch's.  Without that search the extra
        # closure calls of an epoch are plain inner iterations (max_iter_per_epoch > 1) and count as losses of their own
        linesearching = calls[0] > 0 and opt_args.line_search == "strong_wolfe"
        calls[0] += 1
        context["line_search"] = linesearching
        optimizer.zero_grad()
        loss, ctm_env, history, *timings = loss_fn(state, current_env[0], context)
        t_ctm, t_check = timings[0], timings[1]
        t0 = time.perf_counter()
        loss.backward()
        parallel.average_grads(parameters)             # no-op in a single process; distributed runs: mean of the ranks' local gradients
        t1 = time.perf_counter()
        current_env[0] = ctm_env.detach()
        _record(loss.item(), linesearching)
        if opt_args.opt_logging:
            flat_grad = torch.cat(tuple(p.grad.reshape(-1) for p in parameters))
            entry = dict({"id": epoch, "loss": t_data["loss_ls" if linesearching else "loss"][-1], "t_ctm": t_ctm, "t_check": t_check,
                          "t_grad": t1 - t0, "grad_mag": [flat_grad.norm().item(), flat_grad.abs().max().item()]})
            if linesearching:
                entry["LS"] = len(t_data["loss_ls"])
            if opt_args.opt_log_grad:
                entry["grad"] = [torch.view_as_real(p.grad).tolist() if p.grad.is_complex() else p.grad.tolist() for p in parameters]
            log.info(json.dumps(entry))
        context['id'] = epoch
        if obs_fn is not None:
            obs_fn(state, current_env[0], context)
        return loss

    @torch.no_grad()
    def closure_linesearch():
        loc_opt_args = copy.deepcopy(opt_args)
        loc_opt_args.opt_ctm_reinit = opt_args.line_search_ctm_reinit
        loc_ctm_args = copy.deepcopy(ctm_args)
        if opt_args.line_search_svd_method != 'DEFAULT':
            loc_ctm_args.projector_svd_method = opt_args.line_search_svd_method
        ls_context = dict({"ctm_args": loc_ctm_args, "opt_args": loc_opt_args, "loss_history": t_data, "line_search": True})
        loss, ctm_env, history, *timings = loss_fn(state, current_env[0], ls_context)
        current_env[0] = ctm_env
        _record(loss.item(), True)
        if obs_fn is not None:
            context["line_search"] = True
            obs_fn(state, current_env[0], context)
        return loss.item()

    def backtracking_step():
        """One L-BFGS iteration whose step length comes from the derivative-free Armijo search: torch's optimiser computes the
        direction with lr = 1 (its step is undone), the search then moves along it."""
        x0 = [p.detach().clone() for p in parameters]
        group = optimizer.param_groups[0]
        lr0 = group["lr"]
        st = optimizer.state[optimizer._params[0]]
        n_iter0 = st.get("n_iter", 0)
        loss = optimizer.step(closure)
        st = optimizer.state[optimizer._params[0]]
        # step() returns early (gradient below tolerance_grad, or no descent along d) WITHOUT a new direction: the state then still
        # holds the previous epoch's d / t / prev_flat_grad, and searching along that stale direction would move the parameters
        if "d" not in st or loss is None or st.get("n_iter", 0) == n_iter0:
            return
        d, t = st["d"], st["t"]
        loss0 = float(loss.detach())
        g = st["prev_flat_grad"]
        gtd = float(torch.real(torch.vdot(g, d))) if g.is_complex() else float(g.dot(d))

        def move(alpha):
            off = 0
            with torch.no_grad():
                for p, x in zip(parameters, x0):
                    n = p.numel() * (2 if p.is_complex() else 1)
                    step = d[off:off + n]
                    step = torch.view_as_complex(step.view(-1, 2)).view_as(p) if p.is_complex() else step.view_as(p)
                    p.copy_(x + alpha * step)
                    off += n

        def phi(alpha):
            move(alpha)
            return closure_linesearch()
        alpha, _ = _armijo(phi, loss0, gtd, alpha0=float(t))
        while alpha is None and t > opt_args.line_search_tol:
            t = t / 2.0
            alpha, _ = _armijo(phi, loss0, gtd, alpha0=float(t))
        if alpha is None:
            raise RuntimeError("minimize_scalar failed")
        log.info(f"LS final step: {alpha}")
        move(alpha)
        st["t"] = alpha
        group["lr"] = lr0

    for epoch in range(main_args.opt_max_iter):
        if epoch > 0 and len(t_data["loss"]) > 0:
            store_checkpoint(checkpoint_file, state, optimizer, epoch, t_data["loss"][-1])
        calls[0] = 0
        if opt_args.line_search == "backtracking":
            backtracking_step()
        else:
            optimizer.step(closure)
        t_data["loss_ls"] = []
        t_data["min_loss_ls"] = 1.0e+16
        if post_proc is not None:
            post_proc(state, current_env[0], context)
        if len(t_data["loss"]) > 1 and abs(t_data["loss"][-1] - t_data["loss"][-2]) < opt_args.tolerance_change:
            break
        flat_grad = torch.cat(tuple(p.grad.reshape(-1) for p in parameters if p.grad is not None))
        if flat_grad.numel() and flat_grad.abs().max() <= opt_args.tolerance_grad:
            break
        st = optimizer.state[optimizer._params[0]]
        if 'd' in st and st['d'].mul(st['t']).abs().max() <= opt_args.tolerance_change:
            break
    if len(t_data["loss"]) > 0:
        store_checkpoint(checkpoint_file, state, optimizer, main_args.opt_max_iter, t_data["loss"][-1])
