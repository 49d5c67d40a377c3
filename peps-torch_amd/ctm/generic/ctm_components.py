"""Enlarged corners and the two halves of the 4x4 network (reference ctm/generic/ctm_components.py).

The `_t` helpers select the raw tensors exactly like the reference's `c2x2_*_t`; the `_c` /`_sl_c`
closures hand the raw tuple to the native engine (`ctm_c2x2`, `ctm_halves`), which builds
C.T1.T2.a.conj(a) layer by layer on the FP64 matrix cores.
"""
import config as cfg
from config import ctm_args
from backend import get_engine

LU, RU, RD, LD = 0, 1, 2, 3
_CVEC = {LU: ((-1, -1), (0, -1), (-1, 0)), RU: ((1, -1), (1, 0), (0, -1)),
         RD: ((1, 1), (0, 1), (1, 0)), LD: ((-1, 1), (-1, 0), (0, 1))}


def _corner_t(corner, coord, state, env):
    c, t1, t2 = _CVEC[corner]
    s = state.vertexToSite(coord)
    return env.C[(s, c)], env.T[(s, t1)], env.T[(s, t2)], state.site(coord)


def c2x2_LU_t(coord, state, env): return _corner_t(LU, coord, state, env)
def c2x2_RU_t(coord, state, env): return _corner_t(RU, coord, state, env)
def c2x2_RD_t(coord, state, env): return _corner_t(RD, coord, state, env)
def c2x2_LD_t(coord, state, env): return _corner_t(LD, coord, state, env)


def _corner_c(corner, tensors):
    C, T1, T2, a = tensors[:4]
    open_ = bool(tensors[4]) if len(tensors) == 5 and tensors[4] is not None else False
    return get_engine().c2x2(corner, C, T1, T2, a, open_=open_)


def c2x2_LU_sl_c(*tensors): return _corner_c(LU, tensors)
def c2x2_RU_sl_c(*tensors): return _corner_c(RU, tensors)
def c2x2_RD_sl_c(*tensors): return _corner_c(RD, tensors)
def c2x2_LD_sl_c(*tensors): return _corner_c(LD, tensors)
# the engine contracts layer by layer in every mode; the double-layer closures are the same math
c2x2_LU_c, c2x2_RU_c, c2x2_RD_c, c2x2_LD_c = c2x2_LU_sl_c, c2x2_RU_sl_c, c2x2_RD_sl_c, c2x2_LD_sl_c


def _corner(corner, coord, state, env, mode='dl', verbosity=0):
    return get_engine().c2x2(corner, *_corner_t(corner, coord, state, env), open_=mode in ['dl-open', 'sl-open'])


def c2x2_LU(coord, state, env, mode='dl', verbosity=0): return _corner(LU, coord, state, env, mode, verbosity)
def c2x2_RU(coord, state, env, mode='dl', verbosity=0): return _corner(RU, coord, state, env, mode, verbosity)
def c2x2_RD(coord, state, env, mode='dl', verbosity=0): return _corner(RD, coord, state, env, mode, verbosity)
def c2x2_LD(coord, state, env, mode='dl', verbosity=0): return _corner(LD, coord, state, env, mode, verbosity)


# (corner, shift) of the four corners of each move, in the reference's tensor order (ctm_components.py:37-38,
# 105-106,168-169,231-232): first the two corners of R, then the two of Rt.
_HALVES = {
    (0, -1): ((RU, (0, 0)), (RD, (0, 1)), (LU, (-1, 0)), (LD, (-1, 1))),
    (-1, 0): ((LU, (0, 0)), (RU, (1, 0)), (LD, (0, 1)), (RD, (1, 1))),
    (0, 1): ((LD, (0, 0)), (LU, (0, -1)), (RD, (1, 0)), (RU, (1, -1))),
    (1, 0): ((RD, (0, 0)), (LD, (-1, 0)), (RU, (0, -1)), (LU, (-1, -1))),
}


def _halves_t(direction, coord, state, env):
    t = ()
    for corner, sh in _HALVES[direction]:
        t += _corner_t(corner, (coord[0] + sh[0], coord[1] + sh[1]), state, env)
    return t


def _halves(direction, coord, state, env, mode='sl', verbosity=0):
    return get_engine().halves(direction, _halves_t(direction, coord, state, env))


def halves_of_4x4_CTM_MOVE_UP(coord, state, env, mode='sl', verbosity=0): return _halves((0, -1), coord, state, env)
def halves_of_4x4_CTM_MOVE_LEFT(coord, state, env, mode='sl', verbosity=0): return _halves((-1, 0), coord, state, env)
def halves_of_4x4_CTM_MOVE_DOWN(coord, state, env, mode='sl', verbosity=0): return _halves((0, 1), coord, state, env)
def halves_of_4x4_CTM_MOVE_RIGHT(coord, state, env, mode='sl', verbosity=0): return _halves((1, 0), coord, state, env)
def halves_of_4x4_CTM_MOVE_UP_t(coord, state, env): return _halves_t((0, -1), coord, state, env)
def halves_of_4x4_CTM_MOVE_LEFT_t(coord, state, env): return _halves_t((-1, 0), coord, state, env)
def halves_of_4x4_CTM_MOVE_DOWN_t(coord, state, env): return _halves_t((0, 1), coord, state, env)
def halves_of_4x4_CTM_MOVE_RIGHT_t(coord, state, env): return _halves_t((1, 0), coord, state, env)
def halves_of_4x4_CTM_MOVE_UP_c(*t): return get_engine().halves((0, -1), t[:16])
def halves_of_4x4_CTM_MOVE_LEFT_c(*t): return get_engine().halves((-1, 0), t[:16])
def halves_of_4x4_CTM_MOVE_DOWN_c(*t): return get_engine().halves((0, 1), t[:16])
def halves_of_4x4_CTM_MOVE_RIGHT_c(*t): return get_engine().halves((1, 0), t[:16])
