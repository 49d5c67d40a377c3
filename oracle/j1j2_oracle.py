"""numpy restatement of the J1-J2 energy evaluation that consumes the RDMs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates groups/su2.py:5-175 (spin
operators, bipartite rotation) and models/j1j2.py:115-144,223-247,641-679.
"""
import numpy as np
from math import sqrt


def su2_ops(m=2):
    """get_op (groups/su2.py:84-166): basis ordered by descending S^z."""
    I = np.eye(m)
    sz = np.diag([-0.5 * (-(m - 1) + 2 * i) for i in range(m)])
    S = 0.5 * (m - 1)
    sp = np.zeros((m, m)); sm = np.zeros((m, m))
    for i in range(m - 1):
        M = -S + i
        sp[i, i + 1] = sqrt(S * (S + 1) - M * (M + 1))
    for i in range(1, m):
        M = -S + i
        sm[i, i - 1] = sqrt(S * (S + 1) - M * (M - 1))
    return I, sz, sp, sm


def rot_op(m=2):
    """get_rot_op (groups/su2.py:168-172)."""
    r = np.zeros((m, m))
    for i in range(m):
        r[i, m - 1 - i] = (-1) ** i
    return r


def SS(m=2, xyz=(1., 1., 1.)):
    """SU2.SS (groups/su2.py:64-82)."""
    I, sz, sp, sm = su2_ops(m)
    k = 'ij,ab->iajb'
    return xyz[0] * np.einsum(k, sz, sz) + 0.5 * xyz[1] * np.einsum(k, sp, sm) + 0.5 * xyz[2] * np.einsum(k, sm, sp)


def get_hp(j1=1.0, j2=0.0, m=2, hz_stag=0.0, h_uni=(0., 0., 0.), coord=(0, 0)):
    """J1J2.get_hp (models/j1j2.py:97-141), delta_zz = 1: the plaquette operator whose expectation value is E/site."""
    I, sz, sp, sm = su2_ops(m)
    ss = SS(m)
    id2 = np.eye(m * m).reshape(m, m, m, m)
    id3 = np.eye(m ** 3).reshape(m, m, m, m, m, m)
    h = np.einsum('ijab,klcd->ijklabcd', ss, id2)
    hp = 0.5 * j1 * (h + h.transpose(0, 2, 1, 3, 4, 6, 5, 7) + h.transpose(2, 3, 0, 1, 6, 7, 4, 5)
                     + h.transpose(3, 1, 2, 0, 7, 5, 6, 4)) \
        + j2 * (h.transpose(0, 3, 2, 1, 4, 7, 6, 5) + h.transpose(2, 1, 0, 3, 6, 5, 4, 7))
    if hz_stag != 0.0:
        hz = np.einsum('ia,jklbcd->ijklabcd', sz, id3)                                            # :122-123
        hp = hp - 0.25 * hz_stag * ((-1) ** (coord[0] + coord[1])) * (
            hz - hz.transpose(3, 0, 1, 2, 7, 4, 5, 6) - hz.transpose(2, 3, 0, 1, 6, 7, 4, 5) + hz.transpose(1, 2, 3, 0, 5, 6, 7, 4))
    if any(abs(x) > 0 for x in h_uni):
        # S = (S^z, S^x, S^y) (groups/su2.py:49-62): h_uni . S on one site, spread over the four sites of the plaquette (:124-141)
        sx, sy = 0.5 * (sp + sm), -0.5j * (sp - sm)
        h1 = h_uni[0] * sz + h_uni[1] * sx + h_uni[2] * sy
        if np.abs(h1.imag).max() == 0: h1 = h1.real
        hu = np.einsum('ia,jklbcd->ijklabcd', h1, id3)
        hp = hp + 0.25 * (hu + hu.transpose(2, 3, 0, 1, 6, 7, 4, 5) + hu.transpose(3, 0, 1, 2, 7, 4, 5, 6) + hu.transpose(1, 2, 3, 0, 5, 6, 7, 4))
    return hp


def energy_per_site(rdms, j1=1.0, j2=0.0):
    """J1J2.energy_per_site (models/j1j2.py:223-247): mean over sites of tr(rho_2x2 h_p)."""
    hp = get_hp(j1, j2)
    e = sum(np.einsum('ijklabcd,ijklabcd', r, hp) for r in rdms)
    return float(np.real(e)) / len(rdms)


def eval_nnnn_per_site(corrf_1sO1sO, coord=(0, 0)):
    """eval_nnnn_per_site (models/j1j2.py:27-44): third-neighbour S.S along x and y from the distance-2 two-point functions;
    `corrf_1sO1sO(coord, direction, op1, get_op2, dist)` is the correlator routine bound to a state and its environment."""
    I, sz, sp, sm = su2_ops(2)
    f = lambda d, o1, o2: corrf_1sO1sO(coord, d, o1, (lambda r: o2), 2)[1]
    return (f((1, 0), sz, sz) + f((0, 1), sz, sz)
            + 0.5 * (f((1, 0), sp, sm) + f((0, 1), sp, sm) + f((1, 0), sm, sp) + f((0, 1), sm, sp)))


def energy_1x1_lowmem(rdm_nn, rdm_nnn, j1=1.0, j2=0.0):
    """J1J2_C4V_BIPARTITE.energy_1x1_lowmem (models/j1j2.py:641-679), hz_stag=h_uni=j3=0."""
    r = rot_op(2)
    ss = SS(2)
    ss_rot = np.einsum('ki,kjcb,ca->ijab', r, ss, r)          # models/j1j2.py:115
    e = 2.0 * j1 * np.einsum('ijkl,ijkl', rdm_nn, ss_rot)
    if abs(j2) > 0:
        e = e + 2.0 * j2 * np.einsum('ijkl,ijkl', rdm_nnn, ss)
    return float(np.real(e))


def energy_1x1(rdm2x2, j1=1.0, j2=0.0):
    """J1J2_C4V_BIPARTITE.energy_1x1 (models/j1j2.py:591-639): tr(rho_2x2 hp_rot)."""
    r = rot_op(2)
    hp_rot = np.einsum('xj,yk,ixylauvd,ub,vc->ijklabcd', r, r, get_hp(j1, j2), r, r)   # :142-143
    return float(np.real(np.einsum('ijklabcd,ijklabcd', rdm2x2, hp_rot)))


def make_c4v_symm_A1(A):
    """make_c4v_symm_A1 (groups/pg.py:45-54)."""
    A = 0.5 * (A + A.transpose(0, 1, 4, 3, 2))
    A = 0.5 * (A + A.transpose(0, 3, 2, 1, 4))
    A = 0.5 * (A + A.transpose(0, 4, 1, 2, 3))
    A = 0.5 * (A + A.transpose(0, 2, 3, 4, 1))
    return A
