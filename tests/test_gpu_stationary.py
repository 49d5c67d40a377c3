"""GPU: the stationary-environment fast path of the generic truncation (ctm_args.projector_warm_tol > 0; csrc/svd_leading.hip: svd_stationary).

Once a run has converged, the operator of a (direction, site) unit changes by ~1e-10 s_0 from sweep to sweep; with the option on a unit
whose previous singular basis lies that close to its last full solve is truncated by ONE Rayleigh-Ritz half step from that basis,
accepted on the residual of its triplets, instead of a cold block Krylov solve.  The option is off by default (every truncation
solved to the rounding-level threshold, as the reference's full SVD `ctm/generic/ctm_projectors.py:214-229`); on, it plays the role of the
reference's tolerance-driven partial solvers (`:229-257`).  Checked here: the converged corner spectra and the rdm2x2 energy of a run with
the option, AND with / without the whole-move native call (ctm_move), land on the corner spectra and the rdm2x2 energy the REFERENCE's
ctmrg.run produced for the same state and sweep count (tests/golden/fixed_point_*.npz, oracle/gen_golden.py fixed_point) to 1e-10 while the
fast path really is taken; a run with the option agrees with the run without it at D = 4 and D = 6 too; and a basis that is NOT close (the
environment still moves, or the state changed) is refused.  The speed of the stationary sweeps is printed, not asserted (bench.py's
`stationary_environment` block measures it)."""
import time
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sites(D, seed, cplx=False):
    rng = np.random.default_rng(seed)
    out = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, D, D, D, D)) - 0.5                      # signed random tensors: full-rank environment, block Krylov units
            if cplx:
                A = A + 1j * (rng.random((2, D, D, D, D)) - 0.5)
            out[(x, y)] = torch.from_numpy(A / np.abs(A).max()).cuda()
    return out


def _converge(eng, sites, chi, warm_tol, conv_tol, max_sweeps, extra=0):
    """Sweeps until ctmrg_conv_specC (reference ctm/generic/env.py:816-875) says so, plus `extra` sweeps; per-sweep wall times and the
    number of truncations the fast path accepted in each sweep."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env, ctmrg_conv_specC
    from ctm.generic import ctmrg
    import copy
    args = copy.deepcopy(cfg.ctm_args)
    args.projector_warm_tol = warm_tol
    args.ctm_conv_tol = conv_tol
    args.ctm_max_iter = max_sweeps
    st = IPEPS(dict(sites))
    env = ENV(chi, st); init_env(st, env)
    times, accepts, krylov, hist, left = [], [], [], None, None
    for i in range(max_sweeps + extra):
        a0, l0 = eng.stat("warm_accepts"), eng.stat("lz_hits")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for d in args.ctm_move_sequence:
            for _ in range(2):
                ctmrg.ctm_MOVE(d, st, env, ctm_args=args)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        accepts.append(int(eng.stat("warm_accepts") - a0)); krylov.append(int(eng.stat("lz_hits") - l0))
        if left is None:
            conv, hist = ctmrg_conv_specC(st, env, hist, ctm_args=args)
            if conv:
                left = extra
        if left is not None:
            if left == 0:
                break
            left -= 1
    return st, env, times, accepts, krylov, hist


def _energy(st, env):
    from models import j1j2
    return float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))


def _spectra(env):
    return {k: (s / s[0]).cpu().numpy() for k, s in env.get_spectra().items()}


@pytest.mark.parametrize("native_move", [True, False], ids=["ctm_move", "unit-by-unit"])
@pytest.mark.parametrize("warm_tol", [0.0, 1e-9], ids=["solve-every-truncation", "fast-path"])
@pytest.mark.parametrize("name", ["fixed_point_D3_chi48_f64", "fixed_point_D3_chi48_c128"])
def test_fixed_number_of_sweeps_against_the_reference(eng, name, warm_tol, native_move):
    """ctmrg.run for the sweep count of the fixture (the reference's own run, /root/reference/ctm/generic/ctmrg.py:63-110, on a signed
    state whose truncations are block Krylov solves: n = 432, chi = 48) with the stationary fast path off / on and the move as one native
    call / unit by unit: corner spectra and rdm2x2 energy against the REFERENCE's numbers, 1e-10.  With the option on the fast path
    must really have carried truncations (warm_accepts > 0), with it off none."""
    import config as cfg
    import copy
    from conftest import golden
    from helpers import dev, sites_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg, rdm
    from models import j1j2
    g = golden(name)
    chi, nsweeps = int(g["chi"]), int(g["nsweeps"])
    st = IPEPS({k: dev(v) for k, v in sites_from(g).items()})
    env = ENV(chi, st); init_env(st, env)
    args = copy.deepcopy(cfg.ctm_args)
    args.projector_warm_tol = warm_tol
    args.native_move = native_move
    args.ctm_max_iter = nsweeps

    def count(state, env, history, ctm_args=None):
        history = (history or []) + [1]
        return len(history) >= nsweeps, history
    a0, l0 = eng.stat("warm_accepts"), eng.stat("lz_hits")
    try:
        env, hist, *_ = ctmrg.run(st, env, conv_check=count, ctm_args=args)
    finally:
        for e in [eng] + list(eng.workers):
            e.set_option("warm_accept_tol", 0.0); e._warm_tol = 0.0
    assert len(hist) == nsweeps
    accepted, krylov = int(eng.stat("warm_accepts") - a0), int(eng.stat("lz_hits") - l0)
    assert krylov > 0, "the state must reach the block Krylov solver"
    if warm_tol > 0:
        assert accepted > 0, "the fast path was never taken: this test would not pin it"
    else:
        assert accepted == 0
    for k, s_ in env.get_spectra().items():
        ref = g[f"spec_{k[0][0]}_{k[0][1]}_{k[1][0]}_{k[1][1]}"]
        assert np.abs(s_.cpu().numpy() - ref).max() < 1e-10, (k, accepted)
    assert np.abs(rdm.rdm2x2((0, 0), st, env).cpu().numpy() - g["rdm2x2_0_0"]).max() < 1e-10
    e = float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))
    assert abs(e - float(g["energy"])) <= 1e-10 * abs(float(g["energy"])), (e, float(g["energy"]))
    print(f"\n{name} warm_tol={warm_tol} native_move={native_move}: {accepted} of {32 * nsweeps} truncations accepted from the previous basis, {krylov} block Krylov solves")
    env.__dict__.pop("_corner_cache", None)
    eng.trim()


@pytest.mark.parametrize("D,chi,conv_tol,max_sweeps,cplx", [(4, 64, 1e-9, 60, False), pytest.param(6, 128, 1e-8, 40, False, marks=pytest.mark.soak),
                                                            pytest.param(4, 64, 1e-9, 60, True, marks=pytest.mark.soak)],
                         ids=["D4chi64", "D6chi128", "D4chi64-c128"])
def test_converged_run_with_and_without_the_fast_path(eng, D, chi, conv_tol, max_sweeps, cplx):
    sites = _sites(D, 11, cplx)
    try:
        st0, env0, t0, acc0, kr0, h0 = _converge(eng, sites, chi, 0.0, conv_tol, max_sweeps, extra=8)
        assert sum(acc0) == 0 and sum(kr0) > 0, "option off: every truncation is a full solve (and the state must reach the block Krylov solver)"
        st1, env1, t1, acc1, kr1, h1 = _converge(eng, sites, chi, 1e-9, conv_tol, max_sweeps, extra=8)
    finally:
        for e in [eng] + list(eng.workers):
            e.set_option("warm_accept_tol", 0.0); e._warm_tol = 0.0
    assert len(h0['diffs']) == len(h1['diffs']) or abs(len(h0['diffs']) - len(h1['diffs'])) <= 1      # same number of sweeps to convergence
    s0, s1 = _spectra(env0), _spectra(env1)
    for k in s0:
        assert np.abs(s0[k] - s1[k]).max() < 1e-10, k                               # converged corner spectra
    e0, e1 = _energy(st0, env0), _energy(st1, env1)
    assert abs(e0 - e1) <= 1e-10 * abs(e0), (e0, e1)                                # rdm2x2 energy
    # the fast path carried the stationary sweeps: 32 truncations per sweep, all accepted in the last sweeps
    assert acc1[-1] == 32 and acc1[-2] == 32, acc1
    assert kr1[-1] == 0, kr1
    # ... reported, not asserted (a wall-clock ratio on a shared box is not a correctness property): the speed of the stationary sweeps
    cold = min(t0[-4:])
    stat = min(t1[-4:])
    print(f"\nD={D} chi={chi}: {len(t0)} / {len(t1)} sweeps; solve-from-scratch sweep {1e3 * cold:.1f} ms, stationary sweep {1e3 * stat:.1f} ms "
          f"({cold / stat:.2f}x); accepted per sweep {acc1}; E = {e0:.12f} / {e1:.12f}")
    env0.__dict__.pop("_corner_cache", None); env1.__dict__.pop("_corner_cache", None)
    eng.trim()


def test_a_basis_that_is_not_close_is_refused(eng):
    """Converge with the fast path on, then swap in a DIFFERENT state under the same environment object (warm workspaces included): the
    old bases must not be accepted (the units fall back to full solves)."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    import copy
    D, chi = 4, 64
    a, b = _sites(D, 11), _sites(D, 12)
    args = copy.deepcopy(cfg.ctm_args)
    args.projector_warm_tol = 1e-9
    try:
        st, env, *_ = _converge(eng, a, chi, 1e-9, 1e-9, 60, extra=2)
        assert eng.stat("warm_accepts") > 0
        st2 = IPEPS(dict(b))
        a0, l0 = eng.stat("warm_accepts"), eng.stat("lz_hits")
        for _ in range(2):
            for d in args.ctm_move_sequence:
                for _r in range(2):
                    ctmrg.ctm_MOVE(d, st2, env, ctm_args=args)
        # every truncation of the two sweeps on the new state was a full solve: either the fast path was not tried (the singular values
        # of the unit moved) or its residual test refused the old basis
        assert eng.stat("warm_accepts") == a0, "a basis of another state passed"
        assert eng.stat("lz_hits") - l0 == 64
    finally:
        for e in [eng] + list(eng.workers):
            e.set_option("warm_accept_tol", 0.0); e._warm_tol = 0.0
    env.__dict__.pop("_corner_cache", None)
    eng.trim()
