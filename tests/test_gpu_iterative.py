"""GPU: the production configuration of the truncation (n >= 256 (si_min_n): implicit-operator block power iteration, warm started
from the previous sweep, four site-units on concurrent streams) against the numpy oracle (LAPACK gesdd on the explicit
M = R^T Rt) over several full sweeps, on states with a non-trivial spectrum (signed random tensors, f64 and c128)."""
import numpy as np
import pytest
import torch
from helpers import dev, relerr

pytestmark = pytest.mark.gpu


# (most of a case is the numpy oracle on the host: the sweep counts are the smallest that still warm-start, and the complex128 Krylov
# case -- 64 s, also pinned against the reference by tests/test_gpu_stationary.py::test_fixed_number_of_sweeps_against_the_reference -- is `soak`)
@pytest.mark.parametrize("cplx,chi,nsweeps,D", [(False, 32, 2, 4), (True, 32, 2, 3), (False, 64, 1, 3), pytest.param(True, 64, 1, 4, marks=pytest.mark.soak)],
                         ids=["f64", "c128", "f64-krylov", "c128-krylov"])
def test_sweeps_match_oracle_on_iterative_path(eng, cplx, chi, nsweeps, D):
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from models import j1j2
    from oracle import ctm_oracle as O, j1j2_oracle as OJ
    rng = np.random.default_rng(5 + int(cplx))        # (D = 3 where the case allows: n = 288 / 576 >= 256 still takes the iterative route, the oracle is 5x faster)
    sites = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, D, D, D, D)) - 0.5
            if cplx:
                A = A + 1j * (rng.random((2, D, D, D, D)) - 0.5)
            sites[(x, y)] = A / np.abs(A).max()
    st = IPEPS({k: dev(v) for k, v in sites.items()})
    env = ENV(chi, st); init_env(st, env)
    ost = O.State(sites); oe = O.init_env_ctmrg(ost, chi)
    h0, w0, l0 = eng.stat("si_hits"), eng.stat("si_warm_starts"), eng.stat("lz_hits")
    for _ in range(nsweeps):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _r in range(2):
                ctmrg.ctm_MOVE(d, st, env)
        O.ctm_sweep(ost, oe)
    if chi >= 48:
        assert eng.stat("lz_hits") > l0                                      # chi + 1 >= 48 on a hard state: the block Krylov solver ran
    else:
        assert eng.stat("si_hits") > h0 and eng.stat("si_warm_starts") > w0   # the iterative, warm-started path ran
    spec = env.get_spectra(); ospec = O.corner_spectra(oe)
    for k in ospec:
        assert np.abs(spec[k].cpu().numpy() - ospec[k]).max() < 1e-10, k
    # entries carry the 1/sqrt(S) amplification of the smallest kept triplets (S/S0 ~ 1e-8): gauge-invariant |.| to 1e-7,
    # the north-star quantities (spectra above, RDM/energy below) to 1e-10
    for k in oe.C: assert relerr(env.C[k].abs(), np.abs(oe.C[k])) < 1e-7, k
    for k in oe.T: assert relerr(env.T[k].abs(), np.abs(oe.T[k])) < 1e-7, k
    # plaquette RDM of one site (the oracle's open-corner contraction is the slow part of this test)
    from ctm.generic import rdm
    r = rdm.rdm2x2((0, 0), st, env).cpu().numpy()
    ro = O.rdm2x2((0, 0), ost, oe)
    assert np.abs(r - ro).max() < 1e-10, np.abs(r - ro).max()
    eo = OJ.energy_per_site([ro], 1.0, 0.5); e = OJ.energy_per_site([r], 1.0, 0.5)
    assert abs(e - eo) < 1e-10 * max(abs(eo), 1e-3), (e, eo)


def _svd_matrix(n, s, seed):
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.standard_normal((n, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return U, V, (U * s) @ V.T


@pytest.mark.parametrize("factor", [3.0, 40.0])
def test_warm_start_finds_a_new_leading_direction(eng, factor):
    """Between two calls the operator gains a singular direction ORTHOGONAL to the warm basis and larger than the smallest kept
    value.  Every triplet of the warm basis is still an exact singular triplet of the new matrix (residual zero), so a residual
    test alone would accept the old leading chi at the first check; the engine must still return the new direction
    (a full warm block is not accepted before its guard rows have had three half steps)."""
    import torch
    n, chi = 1024, 64
    s = 0.9 ** np.arange(n)
    U, V, M1 = _svd_matrix(n, s, 11)
    s2 = s.copy()
    s2[200] = factor * s[chi]                                   # a direction that was far down the spectrum, now inside the kept range
    M2 = (U * s2) @ V.T
    basis = eng.warm_basis(chi, n, torch.float64)
    cfg = eng.cfg(keep_multiplets=False)
    _, S1, _ = eng.truncated_svd(dev(M1), chi, cfg, basis=basis)
    assert np.abs(S1.cpu().numpy() - s[:chi]).max() < 1e-12
    U2, S2, V2 = (t.cpu().numpy() for t in eng.truncated_svd(dev(M2), chi, cfg, basis=basis))
    exact = np.sort(s2)[::-1][:chi]
    assert np.abs(S2 - exact).max() < 1e-12, "the new singular direction was missed by the warm-started solve"
    assert np.abs(U2.T @ M2 @ V2 - np.diag(S2)).max() < 1e-12
    # third call, unchanged matrix: the warm start is now exact and must reproduce the same values
    _, S3, _ = eng.truncated_svd(dev(M2), chi, cfg, basis=basis)
    assert np.abs(S3.cpu().numpy() - exact).max() < 1e-12


def test_values_below_the_numerical_rank_are_returned_as_zeros(eng):
    """Documented deviation of ctm_truncated_svd from truncated_svd_gesdd: the leading-k solver does not resolve
    singular values below rank_tol = 5e-13 s_0 and returns them as exact zeros, where LAPACK returns numbers of that size
    (themselves only accurate to eps s_0).  Everything above the threshold is exact, and whatever is zeroed is below it."""
    n, chi = 600, 40
    s = np.zeros(n)
    s[:20] = 10.0 ** (-np.arange(20) * 0.5)                    # 1 ... 3e-10
    s[20:30] = 10.0 ** (-13.0 - 0.3 * np.arange(10))           # 1e-13 ... 2e-16: below the numerical rank
    _, _, M = _svd_matrix(n, s, 5)
    S = eng.truncated_svd(dev(M), chi, eng.cfg(keep_multiplets=False))[1].cpu().numpy()
    ex = np.linalg.svd(M, compute_uv=False)[:chi]
    assert np.abs(S[:20] - ex[:20]).max() < 1e-14
    assert (S[S < 5e-13] == 0).all() and ex[S == 0].max() < 5e-13
    assert np.abs(S - ex).max() < 5e-13


LZ_BLOCK_DEFAULT, CROSS_ONLY_DEFAULT = 0, 1          # csrc/ctm_common.h


def test_krylov_solver_variants_agree(eng):
    """The block Krylov truncation of a full-rank unit (signed 2x2 state, D = 4, chi = 64: n = 1024, k = 65) through its variants:
    the sync-free recurrence against the synchronous one, 32- and 64-row blocks, full and cross-pair rounds in the Ritz extraction,
    all four units in flight against serial units.  Same singular values to 1e-12 s0
    and the same environment after two sweeps to 1e-10 (every variant is residual-verified by the solver itself)."""
    import numpy as np, torch
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    rng = np.random.default_rng(11)
    sites = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, 4, 4, 4, 4)) - 0.5
            sites[(x, y)] = torch.from_numpy(A / np.abs(A).max()).cuda()

    def run(opts, concurrent=True):
        for k_, v_ in opts.items(): eng.set_option(k_, v_)
        old = cfg.ctm_args.concurrent_units
        cfg.ctm_args.concurrent_units = concurrent
        try:
            st = IPEPS(dict(sites))
            env = ENV(64, st); init_env(st, env)
            lz0 = eng.stat("lz_hits")
            for _ in range(2):
                for d in cfg.ctm_args.ctm_move_sequence:
                    for _r in range(2):
                        ctmrg.ctm_MOVE(d, st, env)
            assert eng.stat("lz_hits") > lz0, "the state did not reach the block Krylov solver"
            return {k: (s_ / s_[0]).cpu().numpy() for k, s_ in env.get_spectra().items()}
        finally:
            cfg.ctm_args.concurrent_units = old
            for k_ in opts: eng.set_option(k_, {"lz_async": 1, "lz_local_project": 1, "lz_block": LZ_BLOCK_DEFAULT, "jacobi_cross_only": CROSS_ONLY_DEFAULT}[k_])
    ref = run({})
    for name, opts, conc in (("synchronous recurrence", {"lz_async": 0}, True), ("serial units", {}, False),
                             ("two full projection passes", {"lz_local_project": 0}, True),
                             ("32-row blocks", {"lz_block": 32}, True), ("64-row blocks", {"lz_block": 64}, True),
                             ("32-row blocks, synchronous recurrence", {"lz_block": 32, "lz_async": 0}, True),
                             ("cross-pair rounds in the Ritz extraction", {"jacobi_cross_only": 1}, True), ("full rounds", {"jacobi_cross_only": 0}, True)):
        got = run(opts, conc)
        for k in ref:
            assert np.abs(got[k] - ref[k]).max() < 1e-10, (name, k)


@pytest.mark.parametrize("opts", [{"jacobi_cross_only": 0}, {"jacobi_cross_only": 1}], ids=["full-rounds", "cross-pair-rounds"])
def test_many_panel_block_jacobi_variants_against_lapack(eng, opts):
    """The dense one-sided block Jacobi SVD on many 32-row panels (the Ritz extraction of the block Krylov solver; here called directly
    through the full decomposition of an explicit matrix) with full or cross-pair rounds -- singular values against LAPACK to 1e-13 s0,
    orthonormal factors, reconstruction."""
    n = 640
    rng = np.random.default_rng(3)
    U, _ = np.linalg.qr(rng.standard_normal((n, n))); V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.exp(-16.0 * (np.arange(n) / n) ** 0.5)              # steep head, slowly decaying dense tail
    M = (U * s) @ V.T
    keep = {"si_enable": 1, "jacobi_cross_only": CROSS_ONLY_DEFAULT}
    try:
        eng.set_option("si_enable", 0)                         # the dense path, not the leading-k iteration
        for k_, v_ in opts.items(): eng.set_option(k_, v_)
        Ug, Sg, Vg = (t.cpu().numpy() for t in eng.truncated_svd(dev(M), n, eng.cfg(keep_multiplets=False)))
    finally:
        for k_, v_ in keep.items(): eng.set_option(k_, v_)
    ref = np.linalg.svd(M, compute_uv=False)
    big = ref > 1e-11 * ref[0]
    assert np.abs(Sg - ref)[big].max() < 1e-13 * ref[0]
    kk = int(big.sum())
    assert np.abs(Ug[:, :kk].T @ Ug[:, :kk] - np.eye(kk)).max() < 1e-12
    assert np.abs(Vg[:, :kk].T @ Vg[:, :kk] - np.eye(kk)).max() < 1e-12
    assert np.abs((Ug * Sg) @ Vg.T - M).max() < 1e-13 * ref[0] * n


def test_complex_krylov_solver_variants_agree(eng):
    """complex128 twin of test_krylov_solver_variants_agree (signed complex 2x2 state, D = 4, chi = 64, n = 1024, k = 65): 32-row blocks of the
    complex recurrence, and the corner passes as two real products on the stacked [re; im] rows (xgemm_stack_rows) or as four."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    rng = np.random.default_rng(12)
    sites = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, 4, 4, 4, 4)) - 0.5 + 1j * (rng.random((2, 4, 4, 4, 4)) - 0.5)
            sites[(x, y)] = torch.from_numpy(A / np.abs(A).max()).cuda()

    def run(opts):
        for k_, v_ in opts.items(): eng.set_option(k_, v_)
        try:
            st = IPEPS(dict(sites))
            env = ENV(64, st); init_env(st, env)
            lz0 = eng.stat("lz_hits")
            for _ in range(2):
                for d in cfg.ctm_args.ctm_move_sequence:
                    for _r in range(2):
                        ctmrg.ctm_MOVE(d, st, env)
            assert eng.stat("lz_hits") > lz0, "the state did not reach the complex block Krylov solver"
            return {k: (s_ / s_[0]).cpu().numpy() for k, s_ in env.get_spectra().items()}
        finally:
            for k_ in opts: eng.set_option(k_, {"lz_block_c": LZ_BLOCK_C_DEFAULT, "xgemm_stack_rows": 1}[k_])
    ref = run({})
    for name, opts in (("32 complex rows per block", {"lz_block_c": 32}), ("64 complex rows per block", {"lz_block_c": 64}),
                       ("four real products per corner pass", {"xgemm_stack_rows": 0}), ("32 rows, four products", {"lz_block_c": 32, "xgemm_stack_rows": 0})):
        got = run(opts)
        for k in ref:
            assert np.abs(got[k] - ref[k]).max() < 1e-10, (name, k)


LZ_BLOCK_C_DEFAULT = 32          # csrc/ctm_common.h


def test_warm_started_ritz_extraction_and_two_pass_recurrence_give_the_same_sweeps(eng):
    """Round 6 routes of the block Krylov truncation (csrc/svd_leading.hip): the dense SVD of the Ritz matrix started from the rotations of the
    unit's previous extraction (`ritz_warm`; needs the orientation of the returned vectors to follow the previous decomposition, `sign_follow`,
    or the operator of the next sweep is a different matrix) and two instead of three Cholesky-QR passes per block where the unit's previous
    solve needed no third (`lz_two_pass`).  Six sweeps of a signed D = 4 chi = 64 state with the routes on and off: same corner spectra and
    rdm2x2 (1e-10; the gauge of the environment tensors differs by signs), the warm start really is taken, and it needs fewer Jacobi sweeps."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg, rdm
    rng = np.random.default_rng(9)
    D, chi, nsweeps = 4, 64, 6
    sites = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, D, D, D, D)) - 0.5
            sites[(x, y)] = dev(A / np.abs(A).max())
    engines = [eng] + list(eng.workers)
    out = {}
    try:
        for on in (1, 0):
            for e in engines:
                for key in ("ritz_warm", "sign_follow", "lz_two_pass"):
                    e.set_option(key, on)
            st = IPEPS(dict(sites)); env = ENV(chi, st); init_env(st, env)
            per = []
            for _ in range(nsweeps):
                x0, w0, s0 = eng.stat("lz_extractions"), eng.stat("ritz_warm_starts"), eng.stat("ritz_sweeps")
                for d in cfg.ctm_args.ctm_move_sequence:
                    for _r in range(2):
                        ctmrg.ctm_MOVE(d, st, env)
                nx = eng.stat("lz_extractions") - x0
                per.append((int(nx), int(eng.stat("ritz_warm_starts") - w0), (eng.stat("ritz_sweeps") - s0) / max(nx, 1)))
            out[on] = (per, {k: (s_ / s_[0]).cpu().numpy() for k, s_ in env.get_spectra().items()}, rdm.rdm2x2((0, 0), st, env).cpu().numpy())
            env.__dict__.pop("_corner_cache", None)
    finally:
        for e in engines:
            for key in ("ritz_warm", "sign_follow", "lz_two_pass"):
                e.set_option(key, 1)
    (p1, s1, r1), (p0, s0_, r0) = out[1], out[0]
    assert all(nx >= 32 for nx, _, _ in p1[1:]) and all(w == 0 for _, w, _ in p0)
    assert sum(w for _, w, _ in p1) > 32, p1                                    # the warm start is taken ...
    assert p1[-1][2] < 0.75 * p0[-1][2], (p1, p0)                                # ... and the late sweeps need fewer Jacobi sweeps per extraction
    for k in s1:
        assert np.abs(s1[k] - s0_[k]).max() < 1e-10, k
    assert np.abs(r1 - r0).max() < 1e-10
    print(f"\nJacobi sweeps per Ritz extraction, sweep by sweep: routes on {[round(x[2], 2) for x in p1]} (warm started {[x[1] for x in p1]}), off {[round(x[2], 2) for x in p0]}")
    eng.trim()


def test_round6_routes_on_a_state_with_exact_multiplets(eng):
    """The same three routes on an SU(2)-symmetric state (RVB D = 3 of the reference's test-input, tiled on the 2 x 2 cell, chi = 80: exactly
    degenerate singular values, exactly dependent rows inside multiplets -- where a first Cholesky-QR pass shifts and a two-pass solve has to be
    repeated with three, where vectors inside a multiplet have no orientation to follow, and where the Ritz matrix has exactly degenerate
    values): five sweeps with the routes on and off give the same corner spectra."""
    import os
    import config as cfg
    from ipeps.ipeps_c4v import read_ipeps_c4v
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    A = read_ipeps_c4v(os.path.join(root, "tests", "golden", "test-input", "RVB_1x1.in")).site().cuda()
    engines = [eng] + list(eng.workers)
    res = {}
    try:
        for on in (1, 0):
            for e in engines:
                for key in ("ritz_warm", "sign_follow", "lz_two_pass"):
                    e.set_option(key, on)
            st = IPEPS({(x, y): A.clone() for x in range(2) for y in range(2)})
            env = ENV(80, st); init_env(st, env)
            l0, f0, a0 = eng.stat("lz_hits"), eng.stat("si_fallbacks"), eng.stat("lz_async_fallbacks")
            for _ in range(5):
                for d in cfg.ctm_args.ctm_move_sequence:
                    for _r in range(2):
                        ctmrg.ctm_MOVE(d, st, env)
            res[on] = ({k: (v / v[0]).cpu().numpy() for k, v in env.get_spectra().items()},
                       int(eng.stat("lz_hits") - l0), int(eng.stat("si_fallbacks") - f0), int(eng.stat("lz_async_fallbacks") - a0))
            env.__dict__.pop("_corner_cache", None)
    finally:
        for e in engines:
            for key in ("ritz_warm", "sign_follow", "lz_two_pass"):
                e.set_option(key, 1)
    print(f"\nRVB chi = 80: routes on: {res[1][1]} block Krylov solves, {res[1][2]} dense fallbacks, {res[1][3]} repeats on the synchronous path; "
          f"off: {res[0][1]}, {res[0][2]}, {res[0][3]}")
    for k in res[1][0]:
        assert np.abs(res[1][0][k] - res[0][0][k]).max() < 1e-10, k
    eng.trim()
