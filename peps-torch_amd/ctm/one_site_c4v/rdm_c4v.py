"""RDMs of the one-site C4v network (reference ctm/one_site_c4v/rdm_c4v.py:13-136,530-665,1160-1284,
1373-1443,1446-1545)."""
from backend import get_engine
from ctm.generic.rdm import _sym_pos_def_rdm


def _parts(state, env):
    return next(iter(state.sites.values())), env.C[env.keyC], env.T[env.keyT]


def _get_open_C2x2_LU_sl(C, T, a, verbosity=0):
    return get_engine().c2x2_c4v(a, C, T, open_=True)


_get_open_C2x2_LU_dl = _get_open_C2x2_LU_sl


def _rdm(which, who, state, env, sym_pos_def, verbosity):
    raw = get_engine().rdm_c4v(which, *_parts(state, env))
    return _sym_pos_def_rdm(raw, sym_pos_def=sym_pos_def, verbosity=verbosity, who=who)


def rdm2x1_sl(state, env, sym_pos_def=False, force_cpu=False, verbosity=0):
    return _rdm(0, "rdm2x1_sl", state, env, sym_pos_def, verbosity)


def rdm2x2_NN_lowmem_sl(state, env, sym_pos_def=False, force_cpu=False, verbosity=0):
    return _rdm(1, "rdm2x2_NN_lowmem", state, env, sym_pos_def, verbosity)


def rdm2x2_NNN_lowmem_sl(state, env, sym_pos_def=False, force_cpu=False, verbosity=0):
    return _rdm(2, "rdm2x2_NNN_lowmem", state, env, sym_pos_def, verbosity)


def rdm2x2(state, env, sym_pos_def=False, force_cpu=False, verbosity=0):
    return _rdm(3, "rdm2x2", state, env, sym_pos_def, verbosity)


rdm2x1 = rdm2x1_sl
rdm2x2_NN_lowmem = rdm2x2_NN_lowmem_sl
rdm2x2_NNN_lowmem = rdm2x2_NNN_lowmem_sl
