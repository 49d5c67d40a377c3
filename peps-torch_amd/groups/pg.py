"""C4v point-group projections of an on-site tensor a[s,u,l,d,r] (reference groups/pg.py:4-103)."""
import torch

_SIGMA, _UD, _SKEW = (0, 1, 4, 3, 2), (0, 3, 2, 1, 4), (0, 4, 3, 2, 1)
_R, _RINV, _R2 = (0, 4, 1, 2, 3), (0, 2, 3, 4, 1), (0, 3, 4, 1, 2)


def _avg(A, perm, sign=1.0):
    return 0.5 * (A + sign * A.permute(perm))


def make_d2_symm(A):
    return _avg(A, _SIGMA)


def make_d2_antisymm(A):
    return _avg(A, _SIGMA, -1.0)


def make_c4v_symm_A1(A):
    for p in (_SIGMA, _UD, _R, _RINV):
        A = _avg(A, p)
    return A


def make_c4v_symm_A2(A):
    A = _avg(A, _SIGMA, -1.0); A = _avg(A, _SKEW, -1.0); A = _avg(A, _R); A = _avg(A, _R2)
    return A


def make_c4v_symm_B1(A):
    A = _avg(A, _SIGMA); A = _avg(A, _SKEW, -1.0); A = _avg(A, _R, -1.0); A = _avg(A, _R2)
    return A


def make_c4v_symm_B2(A):
    A = _avg(A, _SIGMA, -1.0); A = _avg(A, _SKEW); A = _avg(A, _R); A = _avg(A, _R2, -1.0)
    return A


def make_c4v_symm(A, irreps=["A1"]):
    proj = {"A1": make_c4v_symm_A1, "A2": make_c4v_symm_A2, "B1": make_c4v_symm_B1, "B2": make_c4v_symm_B2}
    irreps = set(irreps)
    assert irreps.issubset(proj.keys()), "Unknown C4v irrep"
    out = torch.zeros(A.size(), device=A.device, dtype=A.dtype)
    for ir in irreps:
        out = out + proj[ir](A)
    return out


def verify_c4v_symm_A1(A, tol=1e-14):
    ds = [(p, torch.dist(A, A.permute(p))) for p in (_SIGMA, _UD, _R, _RINV)]
    mx = max(float(d) for _, d in ds)
    return mx < tol, mx, ds
