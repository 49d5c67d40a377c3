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
