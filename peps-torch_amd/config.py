"""Configuration singletons with the reference's names, defaults and CLI convention.

Mirrors config.py:36-129,246-415 of jurajHasik/peps-torch for the fields the CTMRG + RDM path
reads, so that `examples/j1j2/ctmrg_*.py`-style scripts and their `--CTMARGS_<field>` /
`--GLOBALARGS_<field>` flags work unchanged.  Differences: `GLOBALARGS.device` defaults to the
MI355X (``cuda:0``) because the engine is GPU-only, and two engine options are added to CTMARGS
(`jacobi_tol`, `jacobi_max_sweeps`).
"""
import argparse
import logging
import os
# One hardware queue per HIP stream for the concurrent units of a move (units.UnitPool: up to four worker streams + the caller's).
# The ROCm runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES queues (default 4) and kernels of two streams that
# share a queue do not overlap -- with four worker streams two pairs ran serialised.  The runtime reads the variable when it
# initialises, i.e. at the first HIP call of the process, so it is set here, at the import of the first host-layer module.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch


class _Args:
    _DEFAULTS = {}

    def __init__(self):
        for k, v in self._DEFAULTS.items():
            setattr(self, k, list(v) if isinstance(v, list) else v)

    def __str__(self):
        return type(self).__name__ + "\n" + "\n".join(
            f"{k}= {getattr(self, k)}" for k in sorted(vars(self)) if "__" not in k)


class MAINARGS(_Args):
    _DEFAULTS = dict(omp_cores=1, instate=None, instate_noise=0., ipeps_init_type="RANDOM", out_prefix="output",
                     bond_dim=1, chi=20, opt_max_iter=100, opt_resume=None, opt_resume_override_params=False,
                     seed=0, pattern=None)


class GLOBALARGS(_Args):
    _DEFAULTS = dict(tensor_io_format="legacy", dtype="float64", torch_dtype=torch.float64,
                     device="cuda:0" if torch.cuda.is_available() else "cpu", offload_to_gpu="None",
                     cuda_mem_profile=False)


class PEPSARGS(_Args):
    _DEFAULTS = dict(build_dl=True, build_dl_open=False, quasi_gauge_max_iter=10 ** 6, quasi_gauge_tol=1.0e-8)


class CTMARGS(_Args):
    _DEFAULTS = dict(
        ctm_max_iter=50, ctm_warmup_iter=-1, ctm_env_init_type='CTMRG', ctm_conv_tol=1.0e-8,
        ctm_absorb_normalization='inf', dtype_rdm='DEFAULT', conv_check_cpu=False,
        projector_method='4X4', projector_svd_method='DEFAULT', warmup_projector_svd_method='DEFAULT',
        projector_full_matrices=True, projector_warm_start=True, projector_warm_tol=0.0, native_move=True, share_units=True, concurrent_units=True, unit_stagger_ms=float(os.environ.get('CTM_STAGGER_MS', 0.0)), corner_cache=True, absorb_skip_zero_columns=True, absorb_skip_min_n=8192, projector_svd_reltol=1.0e-8, projector_svd_reltol_block=0.0,
        projector_eps_multiplet=1.0e-8, projector_multiplet_abstol=1.0e-14, ad_decomp_reg=1.0e-12,
        ctm_move_sequence=[(0, -1), (-1, 0), (0, 1), (1, 0)], randomize_ctm_move_sequence=False,
        ctm_force_dl=False, ctm_logging=False,
        verbosity_initialization=0, verbosity_ctm_convergence=0, verbosity_projectors=0, verbosity_ctm_move=0,
        verbosity_rdm=0,
        fwd_checkpoint_c2x2=False, fwd_checkpoint_halves=False, fwd_checkpoint_projectors=False,
        fwd_checkpoint_absorb=False, fwd_checkpoint_move=False, fwd_checkpoint_loop_rdm=False,
        fpcm_init_iter=1, fpcm_freq=-1,
        # engine options (not in the reference)
        # projector_warm_tol > 0 (opt-in, like the reference's tolerance-driven partial solvers ctm_projectors.py:229-257, ARP / PROPACK /
        # RSVD): once the previous sweep's singular basis of a unit lies within this distance of the new solve, the truncation is ONE
        # Rayleigh-Ritz half step from that basis, accepted when the residual of its triplets is <= projector_warm_tol * s_0 (csrc/svd_leading.hip:
        # svd_stationary); 0 = every truncation solved to the engine's rounding-level threshold
        jacobi_tol=1.0e-14, jacobi_max_sweeps=40)


class OPTARGS(_Args):
    _DEFAULTS = dict(lr=1.0, momentum=0., dampening=0., tolerance_grad=1e-5, tolerance_change=1e-9, opt_ctm_reinit=True,
                     env_sens_scale=10.0, env_sens_regauge=False, line_search="default", line_search_ctm_reinit=True,
                     line_search_svd_method='DEFAULT', line_search_tol=1.0e-8, fd_eps=1.0e-4, fd_ctm_reinit=True,
                     history_size=100, max_iter_per_epoch=1, verbosity_opt_epoch=1, opt_logging=True, opt_log_grad=False)


main_args = MAINARGS()
global_args = GLOBALARGS()
peps_args = PEPSARGS()
ctm_args = CTMARGS()
opt_args = OPTARGS()


def _torch_version_check(version):
    t = torch.__version__.split('+')[0].split('.')
    v = version.split('.')
    return (int(t[0]), int(t[1])) >= (int(v[0]), int(v[1]))


def get_args_parser():
    """Same flag scheme as the reference (config.py:36-79): plain main args plus one
    ``--<GROUP>_<field>`` flag per config attribute; booleans become store_true /
    ``--<GROUP>_no_<field>``."""
    p = argparse.ArgumentParser(description='', allow_abbrev=False)
    p.add_argument("--omp_cores", type=int, default=1)
    p.add_argument("--pattern", type=str, default=None)
    p.add_argument("--instate", default=None)
    p.add_argument("--instate_noise", type=float, default=0.)
    p.add_argument("--ipeps_init_type", default="RANDOM")
    p.add_argument("--out_prefix", default="output")
    p.add_argument("--bond_dim", type=int, default=1)
    p.add_argument("--chi", type=int, default=20)
    p.add_argument("--opt_max_iter", type=int, default=100)
    p.add_argument("--opt_resume", type=str, default=None)
    p.add_argument("--opt_resume_override_params", action='store_true')
    p.add_argument("--seed", type=int, default=0)
    for c in (global_args, peps_args, ctm_args, opt_args):
        pref = type(c).__name__ + "_"
        for x in sorted(vars(c)):
            v = getattr(c, x)
            if x in ("torch_dtype", "ctm_move_sequence"):
                continue
            if isinstance(v, bool):
                if not v:
                    p.add_argument("--" + pref + x, action='store_true')
                else:
                    p.add_argument("--" + pref + "no_" + x, action='store_false', dest=pref + x)
            else:
                p.add_argument("--" + pref + x, type=type(v) if v is not None else str, default=v)
    return p


def configure(parsed_args):
    """config.py:81-129: copy parsed flags back into the singletons, resolve the torch dtype,
    set up the log file."""
    groups = {type(c).__name__: c for c in (global_args, peps_args, ctm_args, opt_args)}
    for name, val in vars(parsed_args).items():
        for g, c in groups.items():
            if name.startswith(g + "_"):
                setattr(c, name[len(g) + 1:], val)
                break
        else:
            setattr(main_args, name, val)
    if global_args.dtype == "float64":
        global_args.torch_dtype = torch.float64
    elif global_args.dtype == "complex128":
        global_args.torch_dtype = torch.complex128
    else:
        raise NotImplementedError(f"Unsupported dtype {global_args.dtype}")
    logging.basicConfig(filename=main_args.out_prefix + ".log", filemode='w', level=logging.INFO)


def print_config():
    print(main_args); print(global_args); print(peps_args); print(ctm_args); print(opt_args)
