"""Projectors of a directional move (reference ctm/generic/ctm_projectors.py:14-64,142-293)."""
import config as cfg
from backend import get_engine
from ctm.generic.ctm_components import _halves, _halves_t


def _trunc_cfg(eng, ctm_args):
    return eng.cfg(svd_reltol=ctm_args.projector_svd_reltol, eps_multiplet=ctm_args.projector_eps_multiplet,
                   multiplet_abstol=ctm_args.projector_multiplet_abstol, keep_multiplets=True)


def ctm_get_projectors_4x4(direction, coord, state, env, ctm_args=cfg.ctm_args, global_args=cfg.global_args,
                           diagnostics=None):
    if direction not in [(0, -1), (-1, 0), (0, 1), (1, 0)]:
        raise ValueError("Invalid direction: " + str(direction))
    if ctm_args.projector_svd_method not in ['DEFAULT', 'GESDD']:
        raise ValueError(f"Projector svd method \"{ctm_args.projector_svd_method}\" not implemented")
    eng = get_engine()
    if hasattr(eng, "projectors_4x4"):
        # fused native path: corners -> implicit M = R^T Rt -> P, Pt (halves never materialised)
        return eng.projectors_4x4(direction, _halves_t(direction, coord, state, env), env.chi, _trunc_cfg(eng, ctm_args))
    R, Rt = _halves(direction, coord, state, env)
    return ctm_get_projectors_from_matrices(R, Rt, env.chi, ctm_args, global_args, diagnostics=diagnostics)


def ctm_get_projectors_from_matrices(R, Rt, chi, ctm_args=cfg.ctm_args, global_args=cfg.global_args, diagnostics=None):
    """M = R^T Rt ; U S V^T = M (leading chi) ; P = R conj(U) S^-1/2 , Pt = Rt V S^-1/2."""
    assert R.shape == Rt.shape
    assert len(R.shape) == 2
    if ctm_args.projector_svd_method not in ['DEFAULT', 'GESDD']:
        raise ValueError(f"Projector svd method \"{ctm_args.projector_svd_method}\" not implemented")
    eng = get_engine()
    return eng.projectors(R, Rt, chi, _trunc_cfg(eng, ctm_args))
