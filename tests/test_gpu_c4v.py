"""GPU parity of the one-site C4v move, its RDMs and the reference's published known-answer test."""
import numpy as np
import pytest
import torch
from conftest import golden
from helpers import dev, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["c4v_D2_chi8", "c4v_D3_chi18", "c4v_D2_chi8_c128", "c4v_D3_chi18_c128"])
def case(request):
    return golden(request.param)


def _state_env(g, Ckey="warm_C", Tkey="warm_T"):
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V
    st = IPEPS_C4V(dev(g["site"]))
    env = ENV_C4V(g[Ckey].shape[0], st)
    env.C[env.keyC] = dev(g[Ckey]); env.T[env.keyT] = dev(g[Tkey])
    return st, env


def test_c4v_init(case, eng):
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    st = IPEPS_C4V(dev(case["site"]))
    env = ENV_C4V(case["init_C"].shape[0], st)
    init_env(st, env)
    assert relerr(torch.diagonal(env.get_C()).abs(), np.abs(np.diag(case["init_C"]))) < 1e-12
    assert relerr(env.get_T().abs(), np.abs(case["init_T"])) < 1e-9      # eigenvector signs are a gauge


def test_c4v_corner_move_rdms(case, eng):
    from ctm.one_site_c4v import ctm_components_c4v as cc4, ctmrg_c4v, rdm_c4v
    from models import j1j2
    st, env = _state_env(case)
    assert relerr(cc4.c2x2_sl(st.site(), env.get_C(), env.get_T()), case["c2x2"]) < 1e-12
    for nm, f in (("rdm2x1", rdm_c4v.rdm2x1_sl), ("rdmNN", rdm_c4v.rdm2x2_NN_lowmem_sl), ("rdmNNN", rdm_c4v.rdm2x2_NNN_lowmem_sl),
                  ("rdm2x2", rdm_c4v.rdm2x2)):
        assert relerr(f(st, env, sym_pos_def=True), case[nm]) < 1e-10, nm
    m = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.5)
    assert abs(float(m.energy_1x1_lowmem(st, env)) - float(case["e_lowmem"])) < 1e-11
    assert abs(float(m.energy_1x1(st, env)) - float(case["e_2x2"])) < 1e-11
    ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
    assert relerr(torch.diagonal(env.get_C()), np.diag(case["move_C"])) < 1e-10
    assert relerr(env.get_T().abs(), np.abs(case["move_T"])) < 1e-8


def test_rvb_known_answer(eng):
    """examples/j1j2/ctmrg_j1j2_c4v.py:218-260 (TestRVB): RVB D=3, chi=16, j2=0.5 -> E = -0.47684229 +- 1e-8,
    and the multiplet back-off leaves 3 exact zeros in the corner spectrum."""
    import config as cfg
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    from ctm.one_site_c4v import ctmrg_c4v
    from models import j1j2
    g = golden("rvb_c4v")
    st = IPEPS_C4V(dev(g["site"]))
    env = ENV_C4V(16, st)
    init_env(st, env)
    m = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.5)
    cfg.ctm_args.ctm_max_iter = int(g["nsweeps"])
    ctmrg_c4v.run(st, env)
    e = float(m.energy_1x1_lowmem(st, env))
    assert abs(e - (-0.47684229)) < 1e-8
    assert abs(e - float(g["energy"])) < 1e-10
    spec = torch.diagonal(env.get_C()).abs().cpu().numpy()
    assert (spec == 0).sum() == 3
    assert np.abs(np.sort(spec)[::-1] - np.sort(g["spec"])[::-1]).max() < 1e-10


def test_rvb_chi32_with_and_without_the_orthogonal_iteration(eng):
    """RVB (D = 3) at chi = 32, n = 288: SU(2) multiplets and +-lambda pairs everywhere, and the 128-row block of the orthogonal
    iteration is numerically rank deficient in some moves (its Cholesky steps then return garbage: the iteration has to notice and
    leave -- it used to hand NaNs to the small eigensolver, which failed the whole move).  Gauge-independent results of 14 moves agree
    between the two routes."""
    import config as cfg
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    from ctm.one_site_c4v import ctmrg_c4v
    from models import j1j2
    g = golden("rvb_c4v")
    m = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.5)
    out = {}
    save = cfg.ctm_args.ctm_max_iter
    try:
        for orth in (1, 0):
            eng.set_option("eigh_orth_iter", orth)
            st = IPEPS_C4V(dev(g["site"]))
            env = ENV_C4V(32, st)
            init_env(st, env)
            cfg.ctm_args.ctm_max_iter = 14
            eng.timers(reset=True)
            ctmrg_c4v.run(st, env)
            spec = np.sort(torch.diagonal(env.get_C()).abs().cpu().numpy())[::-1]
            out[orth] = (float(m.energy_1x1_lowmem(st, env)), spec, eng.stat("eigh_orth_hits"))
    finally:
        cfg.ctm_args.ctm_max_iter = save
        eng.set_option("eigh_orth_iter", 1)
    assert out[1][2] >= 5 and out[0][2] == 0
    assert np.isfinite(out[1][0]) and abs(out[1][0] - out[0][0]) < 1e-10
    assert np.abs(out[1][1] - out[0][1]).max() < 1e-10 * out[0][1][0]


@pytest.mark.parametrize("D,chi,signed", [(3, 40, False), (3, 40, True), (4, 32, True), (5, 30, False), (4, 64, True)])
def test_c4v_runs_with_and_without_the_orthogonal_iteration_agree(eng, D, chi, signed):
    """Random C4v-symmetric states at sizes where the moving sweeps take the orthogonal iteration (n = chi D^2 >= 2 x its block):
    corner spectra after 12 moves agree with the regular route to 1e-10 (gauge-independent), positive and signed tensors."""
    import config as cfg
    from groups.pg import make_c4v_symm
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    from ctm.one_site_c4v import ctmrg_c4v
    rng = np.random.default_rng(100 * D + chi + int(signed))
    A = make_c4v_symm(torch.from_numpy(rng.random((2, D, D, D, D)) - (0.5 if signed else 0.0)))
    A = A / A.abs().max()
    out = {}
    save = cfg.ctm_args.ctm_max_iter
    try:
        for orth in (1, 0):
            eng.set_option("eigh_orth_iter", orth)
            st = IPEPS_C4V(A.clone().cuda())
            env = ENV_C4V(chi, st)
            init_env(st, env)
            cfg.ctm_args.ctm_max_iter = 12
            eng.timers(reset=True)
            ctmrg_c4v.run(st, env)
            spec = np.sort(torch.diagonal(env.get_C()).abs().cpu().numpy())[::-1]
            out[orth] = (spec, eng.stat("eigh_orth_hits"), eng.stat("eigh_orth_fails"))
    finally:
        cfg.ctm_args.ctm_max_iter = save
        eng.set_option("eigh_orth_iter", 1)
    assert out[0][1] == 0
    assert np.all(np.isfinite(out[1][0])) and np.abs(out[1][0] - out[0][0]).max() < 1e-10 * out[0][0][0]


def test_c4v_move_with_2norm_normalisation(case, eng):
    """ctm_absorb_normalization = '2' (_move_normalize_c, ctmrg_c4v.py:182-197): T divided by its vector 2-norm, C by |C[0,0]|."""
    import config as cfg
    from ctm.one_site_c4v import ctmrg_c4v
    st, env = _state_env(case)
    old = cfg.ctm_args.ctm_absorb_normalization
    cfg.ctm_args.ctm_absorb_normalization = '2'
    try:
        ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
    finally:
        cfg.ctm_args.ctm_absorb_normalization = old
    assert relerr(torch.diagonal(env.get_C()), np.diag(case["move2_C"])) < 1e-10
    assert relerr(env.get_T().abs(), np.abs(case["move2_T"])) < 1e-8
    assert abs(float(torch.linalg.vector_norm(env.get_T())) - 1.0) < 1e-13


@pytest.mark.parametrize("n,chi", [(96, 24), (300, 300), (640, 40)])
def test_truncated_eigh_complex_hermitian(eng, n, chi):
    """ctm_truncated_eigh on a complex128 Hermitian matrix (eig_sym.py:25-34 is dtype generic): eigenvalues by |lambda| descending,
    signs kept, unitary eigenvectors -- full Jacobi path (n <= 300) and the iterative leading-subspace path (n = 640)."""
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    lam = np.exp(-0.05 * np.arange(n)) * np.where(rng.random(n) < 0.35, -1.0, 1.0)
    H = (Q * lam) @ Q.conj().T
    H = 0.5 * (H + H.conj().T)
    D, U = (t.cpu().numpy() for t in eng.truncated_eigh(dev(H), chi, eng.cfg(keep_multiplets=False)))
    w = np.linalg.eigvalsh(H)
    w = w[np.argsort(-np.abs(w))][:chi]
    assert D.dtype == np.float64 and np.abs(D - w).max() < 1e-13
    assert np.abs(U.conj().T @ U - np.eye(chi)).max() < 1e-12
    assert np.abs(H @ U - U * D[None, :]).max() < 1e-12


# ---- warm restart of the symmetric truncation (ctm_truncated_eigh_ws): accepted only when residual-verified AND the deflated probe
# ---- finds nothing above the smallest accepted |lambda| -------------------------------------------------------------------------
def _sym_with_spectrum(n, lam, seed):
    g = torch.Generator().manual_seed(seed)
    Q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
    return (Q * lam) @ Q.T, Q


def test_warm_restart_is_accepted_on_a_stationary_matrix_and_changes_nothing(eng):
    n, chi = 768, 48
    lam = (0.8 ** torch.arange(n, dtype=torch.float64)) * torch.where(torch.arange(n) % 3 == 1, -1.0, 1.0)     # signs + - + + - + ...
    A, _ = _sym_with_spectrum(n, lam.double(), 3)
    A = A.cuda()
    D0, U0 = eng.truncated_eigh(A, chi)
    basis = eng.warm_basis_c4v(chi, n)
    eng.timers(reset=True)
    outs = [eng.truncated_eigh(A, chi, basis=basis) for _ in range(4)]
    assert eng.stat("eigh_warm_hits") >= 2                  # first call cold, second from the cold call's subspace, then restarts
    for D, U in outs:
        assert float((D - D0).abs().max()) < 1e-13
        assert float(((A @ U) - U * D).abs().max()) < 1e-12
        assert float((U.T @ U - torch.eye(chi, device=U.device, dtype=U.dtype)).abs().max()) < 1e-12
    # same gauge from the regular and the restart path: the columns do not change sign between consecutive calls
    for (_, Ua), (_, Ub) in zip(outs[:-1], outs[1:]):
        assert float((Ua - Ub).abs().max()) < 1e-7


def test_warm_restart_with_a_missing_leading_direction_is_rejected_by_the_probe(eng):
    """The warm subspace is exactly invariant (every residual passes) but lacks the 5th eigenvector: only the probe of the deflated
    matrix can see that; the call must fall back to the regular iteration and return the true leading pairs."""
    n, chi = 768, 48
    lam = (0.8 ** torch.arange(n, dtype=torch.float64))
    A, Q = _sym_with_spectrum(n, lam, 5)
    kk = chi + 1 + 8
    cols = [i for i in range(kk + 1) if i != 4]
    basis = eng.warm_basis_c4v(chi, n)
    assert tuple(basis.shape) == (kk + 1, n)                    # kk vectors + the header row (adaptive state of the sequence)
    basis[:kk].copy_(Q[:, cols].T.contiguous().cuda())
    eng.timers(reset=True)
    D, U = eng.truncated_eigh(A.cuda(), chi, basis=basis)
    assert eng.stat("eigh_warm_rejects") == 1 and eng.stat("eigh_warm_hits") == 0
    assert float((D.cpu() - lam[:chi]).abs().max()) < 1e-12
    assert float((U.cpu().T @ Q[:, :chi]).abs().diagonal()[:30].min()) > 1 - 1e-8      # eigenvalues down to 0.8^29: gaps well above the rounding level


def test_warm_restart_on_a_moved_matrix_falls_back_and_stays_exact(eng):
    n, chi = 512, 32
    lam = (0.8 ** torch.arange(n, dtype=torch.float64)) * torch.where(torch.arange(n) % 3 == 0, -1.0, 1.0)
    A, _ = _sym_with_spectrum(n, lam, 7)
    g = torch.Generator().manual_seed(11)
    E = torch.randn(n, n, generator=g, dtype=torch.float64); E = 1e-5 * (E + E.T)
    basis = eng.warm_basis_c4v(chi, n)
    eng.truncated_eigh(A.cuda(), chi, basis=basis)
    eng.timers(reset=True)
    D, U = eng.truncated_eigh((A + E).cuda(), chi, basis=basis)
    assert eng.stat("eigh_warm_hits") == 0
    w = torch.linalg.eigvalsh(A + E)
    w = w[torch.argsort(w.abs(), descending=True)][:chi]
    assert float((D.cpu() - w).abs().max()) < 1e-12
    assert float((((A + E).cuda() @ U) - U * D).abs().max()) < 1e-12


def test_moved_matrix_takes_the_orthogonal_iteration_and_equals_the_regular_route(eng):
    """A refused warm restart (the matrix moved) goes to the symmetric orthogonal iteration (Cholesky-QR steps, one Rayleigh-Ritz):
    same eigenpairs as the regular block iteration (option off) and as LAPACK, columns in the gauge of the previous call."""
    n, chi = 1024, 64
    lam = (0.85 ** torch.arange(n, dtype=torch.float64)) * torch.where(torch.arange(n) % 4 == 2, -1.0, 1.0)
    A, _ = _sym_with_spectrum(n, lam, 21)
    g = torch.Generator().manual_seed(22)
    E = torch.randn(n, n, generator=g, dtype=torch.float64); E = E + E.T
    E = E / torch.linalg.matrix_norm(E, 2)              # |E|_2 = 1: eps below is the size of the move (gap at the 64th eigenvalue: 5e-6)
    res = {}
    for orth in (1, 0):
        eng.set_option("eigh_orth_iter", orth)
        basis = eng.warm_basis_c4v(chi, n)
        D0, U0 = eng.truncated_eigh(A.cuda(), chi, basis=basis)
        eng.timers(reset=True)
        outs = [(D0, U0)]
        for eps in (3e-7, 1e-8, 1e-10):
            outs.append(eng.truncated_eigh((A + eps * E).cuda(), chi, basis=basis))
        res[orth] = (outs, eng.stat("eigh_orth_hits"), eng.stat("eigh_warm_hits"))
    eng.set_option("eigh_orth_iter", 1)
    assert res[1][1] == 3 and res[0][1] == 0 and res[1][2] == 0
    for k, eps in enumerate((3e-7, 1e-8, 1e-10)):
        M = A + eps * E
        w = torch.linalg.eigvalsh(M)
        w = w[torch.argsort(w.abs(), descending=True)][:chi]
        (D1, U1), (D2, U2) = res[1][0][k + 1], res[0][0][k + 1]
        assert float((D1.cpu() - w).abs().max()) < 1e-12 and float((D1 - D2).abs().max()) < 1e-12
        assert float(((M.cuda() @ U1) - U1 * D1).abs().max()) < 1e-12
        assert float((U1.T @ U1 - torch.eye(chi, device=U1.device, dtype=U1.dtype)).abs().max()) < 1e-12
        # same vectors as the regular route (well separated eigenvalues: 0.85 ratio) and no sign flips against the previous call
        assert float((U1 - U2).abs().max()) < 1e-8
        assert float((U1 * res[1][0][k][1]).sum(0).min()) > 0.99


def test_orthogonal_iteration_then_stationary_restart(eng):
    """The workspace the orthogonal iteration leaves is the one the warm restart accepts once the matrix stops moving."""
    n, chi = 768, 48
    lam = (0.8 ** torch.arange(n, dtype=torch.float64)) * torch.where(torch.arange(n) % 3 == 1, -1.0, 1.0)
    A, _ = _sym_with_spectrum(n, lam, 31)
    g = torch.Generator().manual_seed(32)
    E = torch.randn(n, n, generator=g, dtype=torch.float64); E = E + E.T
    E = 1e-8 * E / torch.linalg.matrix_norm(E, 2)          # (the 57 kept |lambda| go down to 3e-6)
    basis = eng.warm_basis_c4v(chi, n)
    eng.truncated_eigh(A.cuda(), chi, basis=basis)
    eng.timers(reset=True)
    D1, U1 = eng.truncated_eigh((A + E).cuda(), chi, basis=basis)
    assert eng.stat("eigh_orth_hits") == 1 and eng.stat("eigh_warm_hits") == 0
    outs = [eng.truncated_eigh((A + E).cuda(), chi, basis=basis) for _ in range(3)]
    assert eng.stat("eigh_warm_hits") >= 2 and eng.stat("eigh_orth_hits") <= 2
    for D, U in outs:
        assert float((D - D1).abs().max()) < 1e-13 and float((U - U1).abs().max()) < 1e-7


def test_orthogonal_iteration_with_a_slowly_contracting_block(eng):
    """|lambda_129 / lambda_57| = 0.05: eight applications for a move of 1e-4 -- the looks are placed from the measured move and
    contraction, the iteration stays on its route (no fallback to the regular one) and returns the exact pairs."""
    n, chi = 768, 48
    lam = torch.cat([torch.linspace(1.0, 0.2, 60), 0.1 * 0.97 ** torch.arange(n - 60, dtype=torch.float64)]).double()
    lam = lam * torch.where(torch.arange(n) % 5 == 3, -1.0, 1.0)
    A, _ = _sym_with_spectrum(n, lam, 51)
    g = torch.Generator().manual_seed(52)
    E = torch.randn(n, n, generator=g, dtype=torch.float64); E = E + E.T
    E = 1e-4 * E / torch.linalg.matrix_norm(E, 2)
    basis = eng.warm_basis_c4v(chi, n)
    eng.truncated_eigh(A.cuda(), chi, basis=basis)
    eng.timers(reset=True)
    D, U = eng.truncated_eigh((A + E).cuda(), chi, basis=basis)
    assert eng.stat("eigh_orth_hits") == 1 and eng.stat("eigh_orth_fails") == 0
    w = torch.linalg.eigvalsh(A + E)
    w = w[torch.argsort(w.abs(), descending=True)][:chi]
    assert float((D.cpu() - w).abs().max()) < 1e-12
    assert float((((A + E).cuda() @ U) - U * D).abs().max()) < 1e-12
    assert float((U.T @ U - torch.eye(chi, device=U.device, dtype=U.dtype)).abs().max()) < 1e-12


def test_orthogonal_iteration_leaves_a_flat_spectrum_to_the_regular_route(eng):
    """|lambda_129 / lambda_57| = 0.6: 36 applications for a move of 1e-6, more than the iteration allows itself.  The first look measures
    the contraction, the iteration leaves (and stays away for the next calls); the regular route (63 half steps) returns the pairs."""
    n, chi = 768, 48
    lam = torch.cat([torch.linspace(1.0, 0.5, 57), 0.48 * 0.9935 ** torch.arange(n - 57, dtype=torch.float64)]).double()
    A, _ = _sym_with_spectrum(n, lam, 61)
    g = torch.Generator().manual_seed(62)
    E = torch.randn(n, n, generator=g, dtype=torch.float64); E = E + E.T
    E = 1e-6 * E / torch.linalg.matrix_norm(E, 2)
    basis = eng.warm_basis_c4v(chi, n)
    eng.truncated_eigh(A.cuda(), chi, basis=basis)
    eng.timers(reset=True)
    for k in (1, 2, 3):
        M = A + k * E
        D, U = eng.truncated_eigh(M.cuda(), chi, basis=basis)
        w = torch.linalg.eigvalsh(M)
        w = w[torch.argsort(w.abs(), descending=True)][:chi]
        assert float((D.cpu() - w).abs().max()) < 1e-12
        assert float(((M.cuda() @ U) - U * D).abs().max()) < 1e-12
    assert eng.stat("eigh_orth_hits") == 0 and eng.stat("eigh_orth_fails") == 1      # the second and third call skipped it
    eng.set_option("eigh_orth_iter", 1)


def test_orthogonal_iteration_declines_a_numerically_low_rank_block(eng):
    """Fewer significant eigenvalues than the iteration's block: it steps aside (no acceptance) and the regular route returns the pairs."""
    n, chi = 768, 48
    lam = torch.zeros(n, dtype=torch.float64); lam[:40] = 0.7 ** torch.arange(40, dtype=torch.float64)
    A, _ = _sym_with_spectrum(n, lam, 41)
    g = torch.Generator().manual_seed(42)
    v = torch.randn(n, 1, generator=g, dtype=torch.float64); E = 1e-4 * (v @ v.T) / float(v.T @ v)
    basis = eng.warm_basis_c4v(chi, n)
    eng.truncated_eigh(A.cuda(), chi, basis=basis)
    eng.timers(reset=True)
    D, U = eng.truncated_eigh((A + E).cuda(), chi, basis=basis)
    assert eng.stat("eigh_orth_hits") == 0
    w = torch.linalg.eigvalsh(A + E)
    w = w[torch.argsort(w.abs(), descending=True)][:chi]
    assert float((D.cpu()[:41] - w[:41]).abs().max()) < 1e-12


def test_c4v_run_with_and_without_the_warm_restart_agree(eng):
    """20 moves of a random C4v state (converges after ~7): the restart path takes over once the enlarged corner is stationary; corner
    spectrum and energy equal those of the run with the restart switched off."""
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    from ctm.one_site_c4v import ctmrg_c4v
    from groups.pg import make_c4v_symm
    from models import j1j2
    rng = np.random.default_rng(4)
    a = make_c4v_symm(torch.from_numpy(rng.random((2, 4, 4, 4, 4)))).cuda()
    res = []
    for flag in (1, 0):
        eng.set_option("eigh_warm", flag)
        try:
            st = IPEPS_C4V(a.clone())
            env = ENV_C4V(20, st); init_env(st, env)          # n = 320: above the size below which the dense path is used
            eng.timers(reset=True)
            for _ in range(20):
                ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
            hits = eng.stat("eigh_warm_hits")
            e = float(j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.3).energy_1x1_lowmem(st, env))
            res.append((hits, torch.diagonal(env.get_C()).clone(), e))
        finally:
            eng.set_option("eigh_warm", 1)
    assert res[0][0] > 0 and res[1][0] == 0
    assert float((res[0][1] - res[1][1]).abs().max()) < 1e-11
    assert abs(res[0][2] - res[1][2]) < 1e-11


@pytest.mark.parametrize("signs", ["pos", "alt"])
def test_iterative_solver_converges_on_a_flat_leading_spectrum(eng, signs):
    """60 comparable leading eigenvalues (gaps of 1.4 %) over a slowly decaying tail.  The Rayleigh-Ritz of the subspace iteration used
    to measure EVERY pair with a row below the k-th norm against that norm; the unit-normalised guard rows then stayed non-orthogonal
    to the leading rows at tol * tau^2 / (s_i s_j), a residual floor of ~60 x the Jacobi tolerance above the acceptance threshold, and
    every call ran to the iteration cap and fell back to the dense path.  Only guard-guard pairs are relaxed now."""
    n, chi = 768, 48
    sg = torch.tensor([1.0, -1.0]).repeat(30) if signs == "alt" else torch.ones(60)
    lam = torch.cat([torch.linspace(1.0, 0.2, 60) * sg, 0.1 * 0.9 ** torch.arange(n - 60, dtype=torch.float64)]).double()
    A, _ = _sym_with_spectrum(n, lam, 3)
    A = A.cuda()
    eng.timers(reset=True)
    D, U = eng.truncated_eigh(A, chi)
    assert eng.stat("si_hits") == 1 and eng.stat("si_fallbacks") == 0
    assert float((D.cpu() - lam[:chi]).abs().max()) < 1e-12
    assert float(((A @ U) - U * D).abs().max()) < 1e-12


def test_warm_restart_is_accepted_when_the_kept_subspace_cuts_a_degenerate_pair(eng):
    """The chi + 1 + 8 kept pairs end inside an exactly degenerate pair of eigenvalues (position kk and kk + 1): the probe finds a
    Ritz value EQUAL to the smallest accepted |lambda| -- a tie, not a missed direction; the restart must still be accepted."""
    n, chi = 768, 48
    kk = chi + 1 + 8
    lam = (0.8 ** torch.arange(n, dtype=torch.float64))
    lam[kk] = lam[kk - 1]
    A, _ = _sym_with_spectrum(n, lam, 9)
    A = A.cuda()
    basis = eng.warm_basis_c4v(chi, n)
    eng.timers(reset=True)
    outs = [eng.truncated_eigh(A, chi, basis=basis) for _ in range(3)]
    assert eng.stat("eigh_warm_hits") >= 1 and eng.stat("eigh_warm_rejects") == 0
    for D, U in outs:
        assert float((D.cpu() - lam[:chi]).abs().max()) < 1e-13


# ---- complex Hermitian restart (real embedding; only the keep-the-vectors route) --------------------------------------------------
def _herm_with_spectrum(n, lam, seed):
    g = torch.Generator().manual_seed(seed)
    Z = torch.randn(n, n, generator=g, dtype=torch.float64) + 1j * torch.randn(n, n, generator=g, dtype=torch.float64)
    Q, _ = torch.linalg.qr(Z)
    return (Q * lam.to(torch.complex128)) @ Q.conj().T, Q


def test_complex_warm_restart_on_a_stationary_hermitian_matrix(eng):
    """Spectrum with pairs of equal modulus and opposite sign (as the enlarged corner of the A1 + i A2 ansatz has): the workspace keeps
    the eigenvectors after the small Rayleigh-Ritz, not the singular vectors of the iteration (mixtures inside such a pair)."""
    n, chi = 768, 48
    mod = 0.8 ** (torch.arange(n, dtype=torch.float64) // 2)
    lam = mod * torch.where(torch.arange(n) % 2 == 0, 1.0, -1.0)           # +m0, -m0, +m1, -m1, ...
    A, _ = _herm_with_spectrum(n, lam, 21)
    A = A.cuda()
    D0, U0 = eng.truncated_eigh(A, chi)
    basis = eng.warm_basis_c4v(chi, n, A.dtype)
    eng.timers(reset=True)
    outs = [eng.truncated_eigh(A, chi, basis=basis) for _ in range(4)]
    assert eng.stat("eigh_warm_hits") >= 2
    for D, U in outs:
        assert float((D.abs() - D0.abs()).abs().max()) < 1e-13
        assert float(((A @ U) - U * D.to(U.dtype)).abs().max()) < 1e-12
        assert float((U.conj().T @ U - torch.eye(chi, device=U.device, dtype=U.dtype)).abs().max()) < 1e-12
    for (_, Ua), (_, Ub) in zip(outs[:-1], outs[1:]):          # canonical phases: the columns do not change between calls
        assert float((Ua - Ub).abs().max()) < 1e-7


def test_complex_warm_restart_with_a_missing_leading_direction_is_rejected(eng):
    n, chi = 768, 48
    lam = (0.8 ** torch.arange(n, dtype=torch.float64))
    A, Q = _herm_with_spectrum(n, lam, 23)
    kk = chi + 1 + 8
    cols = [i for i in range(kk + 1) if i != 4]
    basis = eng.warm_basis_c4v(chi, n, A.dtype)
    Vh = Q[:, cols].conj().T.contiguous()                      # rows v_i^H
    basis[:2 * kk].copy_(torch.cat([Vh.real, Vh.imag]).cuda())
    eng.timers(reset=True)
    D, U = eng.truncated_eigh(A.cuda(), chi, basis=basis)
    assert eng.stat("eigh_warm_rejects") == 1 and eng.stat("eigh_warm_hits") == 0
    assert float((D.cpu() - lam[:chi]).abs().max()) < 1e-12


def test_complex_c4v_run_with_and_without_the_warm_restart_agree(eng):
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    from ctm.one_site_c4v import ctmrg_c4v
    from groups.pg import make_c4v_symm
    from models import j1j2
    rng = np.random.default_rng(14)
    a = make_c4v_symm(torch.from_numpy(rng.random((2, 4, 4, 4, 4)))) \
        + 1j * make_c4v_symm(torch.from_numpy(rng.random((2, 4, 4, 4, 4)) - 0.5), irreps=["A2"])
    a = (a / a.abs().max()).cuda()
    res = []
    for flag in (1, 0):
        eng.set_option("eigh_warm", flag)
        try:
            st = IPEPS_C4V(a.clone())
            env = ENV_C4V(20, st); init_env(st, env)
            eng.timers(reset=True)
            for _ in range(30):
                ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
            hits = eng.stat("eigh_warm_hits")
            e = float(j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.3).energy_1x1_lowmem(st, env))
            res.append((hits, torch.diagonal(env.get_C()).clone(), e))
        finally:
            eng.set_option("eigh_warm", 1)
    assert res[0][0] > 0 and res[1][0] == 0
    assert float((res[0][1] - res[1][1]).abs().max()) < 1e-11
    assert abs(res[0][2] - res[1][2]) < 1e-11


@pytest.mark.parametrize("tag,chi", [("c4v_f64_D2_chi3", 3), ("c4v_f64_D3_chi12", 12), ("c4v_c128_D2_chi6", 6)])
def test_c4v_env_init_prod_and_obc(eng, tag, chi):
    """ctm_env_init_type PROD / CTMRG_OBC of the C4v environment (env_c4v.py:215-246, 315-355) against the reference's output."""
    import config as cfg
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    g = golden("envinit")
    st = IPEPS_C4V(dev(g[f"{tag}_site"]))
    for kind in ("PROD", "CTMRG_OBC"):
        env = ENV_C4V(chi, st)
        old = cfg.ctm_args.ctm_env_init_type
        cfg.ctm_args.ctm_env_init_type = kind
        try:
            init_env(st, env)
        finally:
            cfg.ctm_args.ctm_env_init_type = old
        assert relerr(env.get_C(), g[f"{tag}_{kind}_C"]) < 1e-13
        T, Tr = env.get_T().cpu().numpy(), g[f"{tag}_{kind}_T"]
        if kind == "PROD":
            ph = np.vdot(T[0, 0, :], Tr[0, 0, :]); ph = ph / abs(ph)
            T = T * ph
        assert float(np.abs(T - Tr).max()) < 1e-12


@pytest.mark.parametrize("base", ["c4v_D2_chi8", "c4v_D3_chi18", "c4v_D2_chi8_c128"])
def test_c4v_rdm3x1_and_the_j3_term(eng, base):
    """rdm3x1_sl and energy_1x1_lowmem with j3 != 0 (reference rdm_c4v.py:829-994, models/j1j2.py:672-676)."""
    from ctm.one_site_c4v import rdm_c4v
    from models import j1j2
    g, j = golden(base), golden("c4v_j3")
    st, env = _state_env(g)
    assert relerr(rdm_c4v.rdm3x1_sl(st, env, sym_pos_def=True), j[f"{base}_rdm3x1"]) < 1e-10
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.3, j3=0.2)
    assert abs(float(model.energy_1x1_lowmem(st, env)) - float(j[f"{base}_e_j3"])) < 1e-11
    assert abs(float(model.energy_1x1(st, env)) - float(j[f"{base}_e1x1_j3"])) < 1e-11
    vals, labels = model.eval_obs(st, env)                                 # same labels, same order, same numbers as the reference
    assert ",".join(labels) == str(j[f"{base}_obs_labels"])
    assert float(np.abs(np.array([complex(v) for v in vals]) - j[f"{base}_obs"]).max()) < 1e-10


@pytest.mark.parametrize("base", ["c4v_D2_chi8", "c4v_D3_chi18", "c4v_D2_chi8_c128"])
def test_c4v_correlators_and_transfer_spectrum(eng, base):
    """What the reference script prints after FINAL (examples/j1j2/ctmrg_j1j2_c4v.py:153-183): rho_1x1, <S(r).S(0)> plain and
    canonical (corrf_c4v.corrf_1sO1sO), leading eigenvalues of the width-1 transfer operator (transferops_c4v.get_Top_spec_c4v)."""
    from ctm.one_site_c4v import rdm_c4v, transferops_c4v
    from models import j1j2
    g, j = golden(base), golden("c4v_j3")
    st, env = _state_env(g)
    assert relerr(rdm_c4v.rdm1x1(st, env), j[f"{base}_rdm1x1"]) < 1e-11
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.3)
    for canon in (False, True):
        c = model.eval_corrf_SS(st, env, 4, canonical=canon)
        for k, v in c.items():
            assert float(np.abs(v.cpu().numpy() - j[f"{base}_corr{'_canon' if canon else ''}_{k}"]).max()) < 1e-10, (canon, k)
    assert float(np.abs(model.eval_corrf_DD_H(st, env, 3)["dd"].cpu().numpy() - j[f"{base}_corr_dd"]).max()) < 1e-10
    top = transferops_c4v.get_Top_spec_c4v(3, st, env).cpu().numpy()
    ref = j[f"{base}_top"]
    assert float(np.abs(np.hypot(top[:, 0], top[:, 1]) - np.hypot(ref[:, 0], ref[:, 1])).max()) < 1e-8     # moduli (conjugate pairs may swap)
    # width-2 channel: vertical dimer-dimer correlator (corrf_c4v.corrf_2sOV2sOV_E2) and transfer operator (get_Top2_spec_c4v)
    assert float(np.abs(model.eval_corrf_DD_V(st, env, 2)["dd"].cpu().numpy() - j[f"{base}_corr_dd_v"]).max()) < 1e-10
    eh, refe = transferops_c4v.get_EH_spec_Ttensor(2, 3, st, env).cpu().numpy(), j[f"{base}_eh3"]      # exp(EH) of a 3-leg cylinder
    assert float(np.abs(np.hypot(eh[:, 0], eh[:, 1]) - np.hypot(refe[:, 0], refe[:, 1])).max()) < 1e-8
    if f"{base}_top2" in j:
        top2 = transferops_c4v.get_Top2_spec_c4v(2, st, env).cpu().numpy()
        ref2 = j[f"{base}_top2"]
        assert float(np.abs(np.hypot(top2[:, 0], top2[:, 1]) - np.hypot(ref2[:, 0], ref2[:, 1])).max()) < 1e-8


@pytest.mark.parametrize("cplx", [False, True])
def test_full_decomposition_warm_start_changes_the_work_not_the_result(eng, cplx):
    """k = n (the differentiable route's SYMEIG node) with a workspace of n rows: the second call on a slightly changed matrix starts
    the Jacobi sweeps from the previous eigenvector rows -- fewer sweeps, same eigenpairs as a cold call (clusters and exact
    degeneracies included: any orthonormal start is valid)."""
    n = 384
    g = torch.Generator().manual_seed(17)
    lam = torch.cat([torch.linspace(1.0, 0.2, n - 8, dtype=torch.float64) * torch.where(torch.arange(n - 8) % 4 == 2, -1.0, 1.0),
                     torch.tensor([0.15, 0.15, 0.15, -0.15, 1e-9, 1e-9, 0.0, 0.0], dtype=torch.float64)])
    if cplx:
        Q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.complex128))
        P = torch.randn(n, n, generator=g, dtype=torch.complex128); P = 0.5 * (P + P.conj().T)
    else:
        Q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
        P = torch.randn(n, n, generator=g, dtype=torch.float64); P = 0.5 * (P + P.T)
    A0 = ((Q * lam) @ Q.conj().T).cuda()
    A1 = (A0 + 1e-4 * P.cuda() / n ** 0.5)
    A1 = 0.5 * (A1 + A1.conj().T)
    cfgT = eng.cfg(keep_multiplets=False)
    basis = eng.warm_basis_c4v(n, n, A0.dtype)
    assert tuple(basis.shape) == ((2 if cplx else 1) * n + 1, n)
    eng.timers(reset=True)
    eng.truncated_eigh(A0, n, cfgT, basis=basis)                      # cold: fills the workspace
    cold_sweeps = eng.stat("total_sweeps")
    assert eng.stat("eigh_warm_hits") == 0
    eng.timers(reset=True)
    D1, U1 = eng.truncated_eigh(A1, n, cfgT, basis=basis)
    assert eng.stat("eigh_warm_hits") == 1 and eng.stat("total_sweeps") < cold_sweeps
    Dc, Uc = eng.truncated_eigh(A1, n, cfgT)
    assert float((D1 - Dc).abs().max()) < 1e-13
    I = torch.eye(n, device=U1.device, dtype=U1.dtype)
    assert float((U1.conj().T @ U1 - I).abs().max()) < 1e-12
    assert float(((A1 @ U1) - U1 * D1.to(U1.dtype)).abs().max()) < 1e-12
    w = torch.linalg.eigvalsh(A1.cpu())
    w = w[torch.argsort(-w.abs())]
    assert float((D1.cpu() - w).abs().max()) < 1e-13
    # a stale workspace (zeros, or rows that are not orthonormal) is ignored
    eng.timers(reset=True)
    D2, _ = eng.truncated_eigh(A1, n, cfgT, basis=torch.zeros_like(basis))
    assert eng.stat("eigh_warm_hits") == 0 and float((D2 - Dc).abs().max()) < 1e-13


def test_orthogonal_iteration_with_two_applications_per_cholesky_step(eng):
    """Option eigh_orth_double (2 by default since round 5; 0 = a Cholesky-QR step after every application): once a look of this workspace has measured |lambda_kk / lambda_0| above the gate, the
    iteration orthonormalises after every SECOND application (1), with the shift Q (A^2 - c^2/2) once an unshifted solve has
    measured the contraction (2).  Same acceptance test, so the same exact pairs as LAPACK; the doubled steps are counted."""
    n, chi = 768, 48
    lam = torch.cat([torch.linspace(1.0, 0.2, 60), 0.1 * 0.97 ** torch.arange(n - 60, dtype=torch.float64)]).double()
    lam = lam * torch.where(torch.arange(n) % 5 == 3, -1.0, 1.0)
    A, _ = _sym_with_spectrum(n, lam, 61)
    g = torch.Generator().manual_seed(62)
    E = torch.randn(n, n, generator=g, dtype=torch.float64); E = E + E.T
    E = E / torch.linalg.matrix_norm(E, 2)
    try:
        eng.set_option("eigh_orth_double", 2)          # (the adaptive state lives in the header row of the fresh workspace below: born zero)
        basis = eng.warm_basis_c4v(chi, n)
        eng.truncated_eigh(A.cuda(), chi, basis=basis)
        eng.timers(reset=True)
        doubled = []
        for k, eps in enumerate((1e-4, 2e-4, 3e-4, 4e-4)):          # (first move: measures the ratio; second: doubled; from the third: shifted)
            M = A + eps * E
            D, U = eng.truncated_eigh(M.cuda(), chi, basis=basis)
            doubled.append(eng.stat("eigh_orth_doubled"))
            w = torch.linalg.eigvalsh(M)
            w = w[torch.argsort(w.abs(), descending=True)][:chi]
            assert float((D.cpu() - w).abs().max()) < 1e-12, k
            assert float(((M.cuda() @ U) - U * D).abs().max()) < 1e-12, k
            assert float((U.T @ U - torch.eye(chi, device=U.device, dtype=U.dtype)).abs().max()) < 1e-12, k
        assert eng.stat("eigh_orth_hits") == 4 and eng.stat("eigh_orth_fails") == 0
        assert doubled[0] == 0 and doubled[3] > doubled[2] > doubled[1] > 0, doubled
    finally:
        eng.set_option("eigh_orth_double", 2)
