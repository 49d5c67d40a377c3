"""CPU: the C++ restatement of one CTM unit (oracle/cpu_unit.cpp, the `cpu_baseline` leg of bench.py) against the numpy oracle on a
committed golden state: same singular values, same absorbed tensors (gauge-invariant norms), for two move directions."""
import os, shutil
import numpy as np
import pytest
from conftest import golden
from helpers_cpu import sites_from


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("direction", ["UP", "LEFT"])
def test_cpp_unit_matches_numpy_oracle(direction):
    from oracle import cpu_unit, ctm_oracle as O
    d = getattr(O, direction)
    g = golden("generic_D3_chi18_f64")
    ost = O.State(sites_from(g))
    oe = O.init_env_ctmrg(ost, 18)
    for _ in range(2):
        O.ctm_sweep(ost, oe)
    r = cpu_unit.run_unit(d, (0, 0), ost, oe, dump=("S", "nC1", "nC2", "nT"))
    assert set(r["times"]) >= {"corners", "halves", "svd", "proj", "absorb", "total"} and r["threads"] >= 1
    R, Rt = O.halves(d, (0, 0), ost, oe)
    P, Pt, S = O.projectors_from_matrices(R, Rt, 18, return_S=True)
    fro, mx, first = r["dumps"]["S"]
    assert abs(fro - np.linalg.norm(S)) < 1e-12 * fro and np.abs(np.array(first) - S[:6]).max() < 1e-12 * S[0]
    nC1, nC2, nT = O.absorb_truncate(d, (0, 0), ost, oe, {c: P for c in ost.sites}, {c: Pt for c in ost.sites})
    for nm, a in (("nC1", nC1), ("nC2", nC2), ("nT", nT)):
        fro, mx, _ = r["dumps"][nm]
        assert abs(fro / np.linalg.norm(a) - 1.0) < 1e-8 and abs(mx / np.abs(a).max() - 1.0) < 1e-8, nm


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_cpp_unit_bounded_svd_sample():
    """svd_nsub: the dgesdd is run on the leading block only (the bounded sample of the n = 16384 baseline); the program still runs
    through projectors and absorb."""
    from oracle import cpu_unit, ctm_oracle as O
    g = golden("generic_D2_chi8_f64")
    ost = O.State(sites_from(g))
    oe = O.init_env_ctmrg(ost, 8)
    O.ctm_sweep(ost, oe)
    r = cpu_unit.run_unit(O.UP, (0, 0), ost, oe, svd_nsub=16, dump=("S",))
    R, Rt = O.halves(O.UP, (0, 0), ost, oe)
    M = R.T @ Rt
    s = np.linalg.svd(M[:16, :16], compute_uv=False)[:8]
    assert np.abs(np.array(r["dumps"]["S"][2]) - s[:6]).max() < 1e-12 * s[0]
