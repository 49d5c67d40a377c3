"""GPU: shape/edge coverage of the native kernels against the oracle -- every padded variant of the fused
two-layer kernel (D = 2..8 -> KT = 1..4), non-square GEMM edges, rank-deficient and flat spectra (fallback from the
block power iteration to the full Jacobi), chi >= n."""
import numpy as np
import pytest
import torch
from helpers import dev, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cplx", [False, True], ids=["f64", "c128"])
@pytest.mark.parametrize("D,chi", [(2, 5), (3, 7), (4, 6), (5, 9), (6, 8), (7, 5), (8, 6), (9, 4), (10, 3)])
def test_corners_and_absorb_all_bond_dims(eng, D, chi, cplx):
    """c2x2 (4 corners, closed + open), the C4v corner and one absorb per direction for every D (fused kernel
    variants KT=1..4 incl. zero padding 25->32, 36->48, 49->64) vs the oracle.  D = 9, 10: beyond the fused two-layer kernel
    (csrc/layer2.hip: layer2_roles refuses legs > 8) -- the executor's pairwise permute + GEMM route (csrc/contract.hip)."""
    from oracle import ctm_oracle as O, c4v_oracle as O4
    rng = np.random.default_rng(100 * D + chi)
    rnd = (lambda *s_: rng.standard_normal(s_) + 1j * rng.standard_normal(s_)) if cplx else (lambda *s_: rng.standard_normal(s_))
    a = rnd(2, D, D, D, D)
    C = rnd(chi, chi)
    Ts = {(0, -1): rnd(chi, D * D, chi), (-1, 0): rnd(chi, chi, D * D), (0, 1): rnd(D * D, chi, chi), (1, 0): rnd(chi, D * D, chi)}
    for cid in range(4):
        sp = O._CORNER[cid]
        T1, T2 = Ts[sp['T1']], Ts[sp['T2']]
        ref = O.c2x2_sl(cid, C, T1, T2, a)
        out = eng.c2x2(cid, dev(C), dev(T1), dev(T2), dev(a))
        assert relerr(out, ref) < 1e-12, (D, cid)
        if D <= 4:
            refo = O.c2x2_sl(cid, C, T1, T2, a, open_=True)
            assert relerr(eng.c2x2(cid, dev(C), dev(T1), dev(T2), dev(a), open_=True), refo) < 1e-12
    Tc = rnd(chi, chi, D * D)
    assert relerr(eng.c2x2_c4v(dev(a), dev(C), dev(Tc)), O4.c2x2_sl(a, C, Tc)) < 1e-12
    # absorb: random projectors
    sites = {(0, 0): a}
    ost = O.State(sites, lX=1, lY=1)
    env = O.Env(chi)
    for v in [(-1, -1), (1, -1), (1, 1), (-1, 1)]: env.C[((0, 0), v)] = rnd(chi, chi)
    for v, t in Ts.items(): env.T[((0, 0), v)] = t
    n = chi * D * D
    P = {(0, 0): rnd(n, chi)}; Pt = {(0, 0): rnd(n, chi)}
    from ctm.generic import ctmrg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    st = IPEPS({(0, 0): dev(a)}, lX=1, lY=1)
    denv = ENV(chi, st)
    denv.C = {k: dev(v) for k, v in env.C.items()}; denv.T = {k: dev(v) for k, v in env.T.items()}
    dP = {(0, 0): dev(P[(0, 0)])}; dPt = {(0, 0): dev(Pt[(0, 0)])}
    for d in O.DIRECTIONS:
        ref = O.absorb_truncate(d, (0, 0), ost, env, P, Pt)
        out = ctmrg._absorb(d, (0, 0), st, denv, dP, dPt, None, normalize=False)
        for r, o in zip(ref, out):
            assert relerr(o, r) < 1e-12, (D, d)


def test_gemm_fast_and_generic_agree(eng):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((256, 384)); B = rng.standard_normal((384, 128))
    ref = A @ B
    eng.set_option("gemm_fast", 1); f = eng.gemm(dev(A), dev(B)).cpu().numpy()
    eng.set_option("gemm_fast", 0); g = eng.gemm(dev(A), dev(B)).cpu().numpy()
    eng.set_option("gemm_fast", 1)
    assert np.abs(f - ref).max() < 1e-11 and np.abs(g - ref).max() < 1e-11
    for tA, tB in ((True, False), (False, True), (True, True)):
        a = A.T.copy() if tA else A; b = B.T.copy() if tB else B
        assert np.abs(eng.gemm(dev(a), dev(b), tA, tB).cpu().numpy() - ref).max() < 1e-11


def test_flat_spectrum_falls_back_to_full_jacobi(eng):
    """A dense random matrix has a flat spectrum: the block power iteration cannot converge and the engine must fall
    back to the full decomposition -- same answer as LAPACK."""
    rng = np.random.default_rng(2)
    n, chi = 640, 40
    M = rng.random((n, n)) - 0.5
    eng.set_option("si_max_iter", 6)
    fb0 = eng.stat("si_fallbacks")
    U, S, V = (t.cpu().numpy() for t in eng.truncated_svd(dev(M), chi))
    eng.set_option("si_max_iter", 40)
    assert eng.stat("si_fallbacks") == fb0 + 1
    Sr = np.linalg.svd(M, compute_uv=False)[:chi]
    assert np.abs(S - Sr).max() < 1e-12 * Sr[0]
    assert np.abs(U.T @ M @ V - np.diag(S)).max() < 1e-11 * Sr[0]


def test_rank_deficient_and_small_chi_ge_n(eng):
    rng = np.random.default_rng(3)
    n = 600
    X = rng.standard_normal((n, 7)); Y = rng.standard_normal((7, n))
    M = X @ Y                                   # rank 7
    U, S, V = (t.cpu().numpy() for t in eng.truncated_svd(dev(M), 32))
    Sr = np.linalg.svd(M, compute_uv=False)[:32]
    assert np.abs(S[:7] - Sr[:7]).max() < 1e-12 * Sr[0]
    assert (S[7:] < 1e-10 * Sr[0]).all()
    assert np.abs((U[:, :7] * S[:7]) @ V[:, :7].T - M).max() < 1e-10 * Sr[0]
    # chi >= n : no truncation
    m = rng.standard_normal((20, 20))
    U, S, V = (t.cpu().numpy() for t in eng.truncated_svd(dev(m), 32))
    assert S.shape == (20,) and np.abs(S - np.linalg.svd(m, compute_uv=False)).max() < 1e-13 * S[0]
    assert np.abs((U * S) @ V.T - m).max() < 1e-12


def test_projectors_fused_equals_explicit(eng):
    """ctm_projectors_4x4 (implicit M) == ctm_halves + ctm_projectors (explicit R, Rt) on gauge invariants."""
    from conftest import golden
    from helpers import sites_from, env_from, device_state_env, DIRS
    from ctm.generic.ctm_components import _halves_t
    g = golden("generic_D3_chi18_f64")
    C, T = env_from(g, "warm_")
    st, env = device_state_env(sites_from(g), C, T, 18)
    for dn, d in DIRS.items():
        t16 = _halves_t(d, (0, 0), st, env)
        R, Rt = eng.halves(d, t16)
        P, Pt, S = eng.projectors(R, Rt, 18, return_S=True)
        P2, Pt2, S2 = eng.projectors_4x4(d, t16, 18, return_S=True)
        assert relerr(S2, S) < 1e-12
        assert relerr(P2 @ Pt2.t(), P @ Pt.t()) < 1e-7


def test_concurrent_units_and_warm_start_do_not_change_the_environment(eng):
    """Two sweeps with (concurrent streams + warm-started truncation) == two sweeps with both switched off,
    on gauge invariants; n = 512 so the iterative truncation (the only consumer of the warm basis) is active."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    rng = np.random.default_rng(21)
    D, chi = 4, 32
    sites = {(x, y): rng.random((2, D, D, D, D)) for y in range(2) for x in range(2)}
    sites = {k: v / np.abs(v).max() for k, v in sites.items()}
    res = []
    for flag in (True, False):
        cfg.ctm_args.concurrent_units = flag
        cfg.ctm_args.projector_warm_start = flag
        st = IPEPS({k: dev(v) for k, v in sites.items()})
        env = ENV(chi, st); init_env(st, env)
        w0 = eng.stat("si_warm_starts")
        for _ in range(2):
            for d in cfg.ctm_args.ctm_move_sequence:
                for _r in range(2):
                    ctmrg.ctm_MOVE(d, st, env)
        res.append(env)
        if flag:
            assert eng.stat("si_warm_starts") > w0 and len(eng.workers) >= 2
        else:
            assert eng.stat("si_warm_starts") == w0
    cfg.ctm_args.concurrent_units = True
    cfg.ctm_args.projector_warm_start = True
    for k in res[0].C: assert relerr(res[0].C[k].abs(), res[1].C[k].abs()) < 1e-8, k
    for k in res[0].T: assert relerr(res[0].T[k].abs(), res[1].T[k].abs()) < 1e-8, k
    for k, s in res[0].get_spectra().items():
        assert relerr(s, res[1].get_spectra()[k]) < 1e-9


@pytest.mark.parametrize("name,chi", [("generic_D2_chi8_f64", 8), ("generic_D2_chi8_c128", 8)])
def test_two_norm_normalisation_and_force_dl_flag(eng, name, chi):
    """ctm_absorb_normalization = '2' (vector 2-norm, ctmrg.py:212-214) and the ctm_force_dl flag (same mathematics)."""
    import config as cfg
    from conftest import golden
    from helpers import sites_from, env_from, device_state_env, oracle_state_env
    from ctm.generic import ctmrg
    from oracle import ctm_oracle as O
    g = golden(name)
    C, T = env_from(g, "warm_")
    st, env = device_state_env(sites_from(g), C, T, chi)
    ost, oe = oracle_state_env(sites_from(g), C, T, chi)
    cfg.ctm_args.ctm_absorb_normalization = '2'
    cfg.ctm_args.ctm_force_dl = True
    try:
        for d in [(0, -1), (-1, 0), (0, 1), (1, 0)]:
            ctmrg.ctm_MOVE(d, st, env)
            O.ctm_move(d, ost, oe, norm_type='2')
    finally:
        cfg.ctm_args.ctm_absorb_normalization = 'inf'
        cfg.ctm_args.ctm_force_dl = False
    for k in oe.C: assert relerr(env.C[k].abs(), np.abs(oe.C[k])) < 1e-7, k
    for k in oe.T: assert relerr(env.T[k].abs(), np.abs(oe.T[k])) < 1e-7, k
    for k, t in env.T.items(): assert abs(float(torch.linalg.vector_norm(t)) - 1.0) < 1e-12


@pytest.mark.parametrize("name,chi", [("generic_D2_chi8_f64", 8), ("generic_D2_chi8_c128", 8)])
def test_projector_method_4x2(eng, name, chi):
    """projector_method = '4X2' (ctm_projectors.py:66-136): two sweeps vs the oracle (itself checked against the reference
    on these states when this test was written)."""
    import config as cfg
    from conftest import golden
    from helpers import sites_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from oracle import ctm_oracle as O
    g = golden(name)
    sites = sites_from(g)
    st = IPEPS({k: dev(v) for k, v in sites.items()})
    env = ENV(chi, st); init_env(st, env)
    ost = O.State(sites); oe = O.init_env_ctmrg(ost, chi)
    cfg.ctm_args.projector_method = '4X2'
    try:
        for _ in range(2):
            for d in cfg.ctm_args.ctm_move_sequence:
                for _r in range(2):
                    ctmrg.ctm_MOVE(d, st, env)
                    O.ctm_move(d, ost, oe, projector_method='4X2')
    finally:
        cfg.ctm_args.projector_method = '4X4'
    for k in oe.C: assert relerr(env.C[k].abs(), np.abs(oe.C[k])) < 1e-7, k
    for k in oe.T: assert relerr(env.T[k].abs(), np.abs(oe.T[k])) < 1e-7, k
    with pytest.raises(ValueError):
        cfg.ctm_args.projector_method = '4X3'
        try:
            ctmrg.ctm_MOVE((0, -1), st, env)
        finally:
            cfg.ctm_args.projector_method = '4X4'


@pytest.mark.parametrize("cplx", [False, True], ids=["f64", "c128"])
def test_slowly_decaying_tail_uses_block_krylov(eng, cplx):
    """A spectrum with a fast head and a long, slowly decaying tail (what entangled states give): the subspace iteration
    hands over to the block Golub-Kahan-Lanczos solver; same triplets as LAPACK."""
    rng = np.random.default_rng(31)
    n, chi = 1536, 64
    rnd = (lambda: rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) if cplx else (lambda: rng.standard_normal((n, n)))
    Q1, _ = np.linalg.qr(rnd()); Q2, _ = np.linalg.qr(rnd())
    i = np.arange(n)
    sv = np.sort(np.where(i < 30, 0.6 ** i, 1e-6 / (1.0 + 0.02 * (i - 30))))[::-1]
    M = (Q1 * sv) @ Q2.conj().T
    h0 = eng.stat("lz_hits")
    U, S, V = (t.cpu().numpy() for t in eng.truncated_svd(dev(M), chi, eng.cfg(keep_multiplets=False)))
    assert eng.stat("lz_hits") == h0 + 1
    assert np.abs(S - sv[:chi]).max() < 1e-13
    assert np.abs(M @ V - U * S).max() < 1e-12 and np.abs(U.conj().T @ M - S[:, None] * V.conj().T).max() < 1e-12
    assert np.abs(U.conj().T @ U - np.eye(chi)).max() < 1e-12 and np.abs(V.conj().T @ V - np.eye(chi)).max() < 1e-12


@pytest.mark.parametrize("Dv,Dh,chi", [(3, 2, 6), (2, 3, 7), (4, 3, 48)])
def test_direction_dependent_bond_dimensions(eng, Dv, Dh, chi):
    """Vertical and horizontal bonds of different dimension (rectangular enlarged corners, square halves): one full sweep,
    fused and explicit projector routes, vs the oracle.  (4,3,48): n = 768 / 432, the implicit-operator iteration."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from ctm.generic.ctm_components import _halves_t
    from oracle import ctm_oracle as O
    rng = np.random.default_rng(7 * Dv + Dh)
    sites = {(x, y): rng.random((2, Dv, Dh, Dv, Dh)) - 0.3 for y in range(2) for x in range(2)}
    sites = {k: v / np.abs(v).max() for k, v in sites.items()}
    st = IPEPS({k: dev(v) for k, v in sites.items()})
    env = ENV(chi, st); init_env(st, env)
    ost = O.State(sites); oe = O.init_env_ctmrg(ost, chi)
    for k in oe.C: assert relerr(env.C[k], oe.C[k]) < 1e-12
    for k in oe.T: assert relerr(env.T[k], oe.T[k]) < 1e-12
    for d in cfg.ctm_args.ctm_move_sequence:
        # same environment on both sides at the start of every direction (moves fix the gauge only up to signs)
        env.C = {k: dev(v) for k, v in oe.C.items()}; env.T = {k: dev(v) for k, v in oe.T.items()}
        env.__dict__.pop("_warm", None)
        t16 = _halves_t(d, (0, 0), st, env)
        R, Rt = eng.halves(d, t16)
        Ro, Rto = O.halves(d, (0, 0), ost, oe)
        assert relerr(R, Ro) < 1e-11 and relerr(Rt, Rto) < 1e-11
        P, Pt, S = eng.projectors(R, Rt, chi, return_S=True)
        P2, Pt2, S2 = eng.projectors_4x4(d, t16, chi, return_S=True)
        Po, Pto, So = O.projectors_from_matrices(Ro, Rto, chi, return_S=True)
        assert relerr(S, So) < 1e-11 and relerr(S2, So) < 1e-11
        assert relerr(P2 @ Pt2.t(), Po @ Pto.T) < 1e-6 and relerr(P @ Pt.t(), Po @ Pto.T) < 1e-6
        for _r in range(2):
            ctmrg.ctm_MOVE(d, st, env)
            O.ctm_move(d, ost, oe)
    for k in oe.C: assert relerr(env.C[k].abs(), np.abs(oe.C[k])) < 1e-7, k
    for k in oe.T: assert relerr(env.T[k].abs(), np.abs(oe.T[k])) < 1e-7, k
    spec = env.get_spectra(); ospec = O.corner_spectra(oe)
    for k in ospec: assert np.abs(spec[k].cpu().numpy() - ospec[k]).max() < 1e-9, k


@pytest.mark.parametrize("name,chi", [("generic_D2_chi8_f64", 8), ("generic_D2_chi8_c128", 8)])
@pytest.mark.parametrize("chi_out", [5, 12])
def test_absorb_with_different_environment_and_projector_dimensions(eng, name, chi, chi_out):
    """ctm_absorb_x: incoming tensors of dimension chi_in, projectors with chi_out columns (a growing or shrinking
    environment), every direction, against the oracle's absorb on the same operands."""
    from helpers import sites_from, env_from, device_state_env, oracle_state_env, DIRS
    from conftest import golden
    from ctm.generic import ctmrg
    from oracle import ctm_oracle as O
    g = golden(name)
    sites = sites_from(g); C, T = env_from(g, "warm_")
    st, env = device_state_env(sites, C, T, chi)
    ost, oe = oracle_state_env(sites, C, T, chi)
    rng = np.random.default_rng(11)
    cplx = next(iter(sites.values())).dtype.kind == 'c'
    for d in DIRS.values():
        P, Pt = {}, {}
        for c, a in sites.items():
            leg = {(0, -1): 2, (-1, 0): 3, (0, 1): 4, (1, 0): 1}[d]
            n = chi * a.shape[leg] ** 2
            mk = lambda: (rng.standard_normal((n, chi_out)) + (1j * rng.standard_normal((n, chi_out)) if cplx else 0))
            P[c], Pt[c] = mk(), mk()
        dP, dPt = {c: dev(v) for c, v in P.items()}, {c: dev(v) for c, v in Pt.items()}
        for c in sites:
            got = ctmrg._absorb(d, c, st, env, dP, dPt, None, normalize=False)
            want = O.absorb_truncate(d, c, ost, oe, P, Pt)
            for x, y in zip(got, want):
                assert tuple(x.shape) == y.shape
                assert relerr(x, y) < 1e-12


def test_one_move_beyond_the_fused_kernel(eng):
    """D = 9 (n = chi D^2 = 486): one whole directional move of a 1x1 cell -- corners by the pairwise route, fused and explicit
    projector paths, absorb -- against the oracle."""
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from ctm.generic.ctm_components import _halves_t
    from oracle import ctm_oracle as O
    D, chi = 9, 6
    rng = np.random.default_rng(99)
    a = rng.random((2, D, D, D, D)) - 0.4
    sites = {(0, 0): a / np.abs(a).max()}
    st = IPEPS({k: dev(v) for k, v in sites.items()}, lX=1, lY=1)
    env = ENV(chi, st); init_env(st, env)
    ost = O.State(sites, lX=1, lY=1); oe = O.init_env_ctmrg(ost, chi)
    for k in oe.C: assert relerr(env.C[k], oe.C[k]) < 1e-12
    for k in oe.T: assert relerr(env.T[k], oe.T[k]) < 1e-12
    d = (0, -1)
    t16 = _halves_t(d, (0, 0), st, env)
    R, Rt = eng.halves(d, t16)
    Ro, Rto = O.halves(d, (0, 0), ost, oe)
    assert relerr(R, Ro) < 1e-11 and relerr(Rt, Rto) < 1e-11
    P2, Pt2, S2 = eng.projectors_4x4(d, t16, chi, return_S=True)
    Po, Pto, So = O.projectors_from_matrices(Ro, Rto, chi, return_S=True)
    assert relerr(S2, So) < 1e-11
    assert relerr(P2 @ Pt2.t(), Po @ Pto.T) < 1e-6
    ctmrg.ctm_MOVE(d, st, env)
    O.ctm_move(d, ost, oe)
    spec = env.get_spectra(); ospec = O.corner_spectra(oe)
    for k in ospec: assert np.abs(spec[k].cpu().numpy() - ospec[k]).max() < 1e-9, k
    for k in oe.T: assert relerr(env.T[k].abs(), np.abs(oe.T[k])) < 1e-7, k


@pytest.mark.parametrize("name", ["rect_cut_chi5_f64", "rect_cut_chi5_c128"])
def test_bond_dimensions_that_differ_along_one_cut(eng, name):
    """The reference only asserts R.shape == Rt.shape (ctm/generic/ctm_projectors.py:209): horizontal bonds of dimension 2 in the upper row
    and 3 in the lower row make the halves of an UP / DOWN move rectangular (a = chi 2^2 rows -- the truncated bond -- by b = chi 3^2).  The
    fused implicit-operator entry truncates square halves and says so by name (ValueError from its own shape check, before any kernel);
    the host layer sends such a unit the explicit way (ctm_halves -> ctm_projectors_rect: M = R^T Rt is b x b, P = R conj(U) S^-1/2 is
    a x chi).  One move per direction and whole sweeps against the REFERENCE's run of the same state
    (tests/golden/rect_cut_*.npz, oracle/gen_golden.py rect_cut)."""
    import config as cfg
    from conftest import golden
    from helpers import sites_from, env_from, DIRS
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from ctm.generic.ctm_projectors import rectangular_unit
    from ctm.generic.ctm_components import _halves_t
    g = golden(name)
    chi, nsweeps = int(g["chi"]), int(g["nsweeps"])
    st = IPEPS({k: dev(v) for k, v in sites_from(g).items()})
    env = ENV(chi, st); init_env(st, env)
    assert rectangular_unit((0, -1), (0, 0), st, env) and rectangular_unit((0, 1), (0, 0), st, env)
    assert not rectangular_unit((-1, 0), (0, 0), st, env) and not rectangular_unit((1, 0), (0, 0), st, env)
    with pytest.raises(ValueError, match="differ along one cut"):
        eng.projectors_4x4((0, -1), _halves_t((0, -1), (0, 0), st, env), chi)
    for dn, d in DIRS.items():
        env = ENV(chi, st); init_env(st, env)
        ctmrg.ctm_MOVE(d, st, env)
        C1, T1 = env_from(g, f"move_{dn}_")
        for k in C1: assert relerr(env.C[k].abs(), np.abs(C1[k])) < 1e-7, (dn, k)
        for k in T1: assert relerr(env.T[k].abs(), np.abs(T1[k])) < 1e-7, (dn, k)
    env = ENV(chi, st); init_env(st, env)
    for _ in range(nsweeps):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _r in range(2):
                ctmrg.ctm_MOVE(d, st, env)
    for k, s_ in env.get_spectra().items():
        assert np.abs(s_.cpu().numpy() - g[f"spec_{k[0][0]}_{k[0][1]}_{k[1][0]}_{k[1][1]}"]).max() < 1e-10, k
    C1, T1 = env_from(g, "end_")
    for k in C1: assert relerr(env.C[k].abs(), np.abs(C1[k])) < 1e-7, k
    for k in T1: assert relerr(env.T[k].abs(), np.abs(T1[k])) < 1e-7, k
