"""GPU parity of the generic directional CTM move and RDMs: native HIP path (through the C-ABI and the
host layer that mirrors the reference API) vs the numpy oracle and the committed golden vectors.

Tolerances (float64): contractions 1e-12 relative; projector-dependent tensors 1e-8 on gauge invariants
(|.|, P Pt^T) because singular triplets with S/S0 ~ 1e-7 amplify rounding by 1/S; converged spectra and
rdm2x2 energies 1e-10 relative (the north-star bar)."""
import numpy as np
import pytest
import torch
from conftest import golden
from helpers import DIRS, dev, sites_from, env_from, device_state_env, oracle_state_env, relerr

pytestmark = pytest.mark.gpu
CASES = [("generic_D2_chi8_f64", 8), ("generic_D3_chi18_f64", 18), ("generic_D2_chi8_c128", 8)]


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def case(request):
    name, chi = request.param
    g = golden(name)
    sites = sites_from(g)
    C, T = env_from(g, "warm_")
    return dict(g=g, chi=chi, sites=sites, C=C, T=T)


def test_init_env(case, eng):
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    st = IPEPS({k: dev(v) for k, v in case["sites"].items()})
    env = ENV(case["chi"], st)
    init_env(st, env)
    C0, T0 = env_from(case["g"], "init_")
    for k in C0: assert relerr(env.C[k], C0[k]) < 1e-13, k
    for k in T0: assert relerr(env.T[k], T0[k]) < 1e-13, k


def test_corners(case, eng):
    from ctm.generic import ctm_components as cc
    from oracle import ctm_oracle as O
    st, env = device_state_env(case["sites"], case["C"], case["T"], case["chi"])
    ost, oe = oracle_state_env(case["sites"], case["C"], case["T"], case["chi"])
    fs = {0: cc.c2x2_LU, 1: cc.c2x2_RU, 2: cc.c2x2_RD, 3: cc.c2x2_LD}
    for cid, f in fs.items():
        assert relerr(f((0, 0), st, env, mode='sl'), case["g"][f"c2x2_{cid}"]) < 1e-12          # golden (reference output)
        for coord in case["sites"]:
            assert relerr(f(coord, st, env, mode='sl'), O.c2x2(cid, coord, ost, oe)) < 1e-12
            assert relerr(f(coord, st, env, mode='sl-open'), O.c2x2(cid, coord, ost, oe, open_=True)) < 1e-12


def test_halves_projectors_absorb(case, eng):
    from ctm.generic import ctm_components as cc, ctm_projectors as cp, ctmrg
    from oracle import ctm_oracle as O
    g, chi = case["g"], case["chi"]
    st, env = device_state_env(case["sites"], case["C"], case["T"], chi)
    ost, oe = oracle_state_env(case["sites"], case["C"], case["T"], chi)
    hf = {'UP': cc.halves_of_4x4_CTM_MOVE_UP, 'LEFT': cc.halves_of_4x4_CTM_MOVE_LEFT,
          'DOWN': cc.halves_of_4x4_CTM_MOVE_DOWN, 'RIGHT': cc.halves_of_4x4_CTM_MOVE_RIGHT}
    af = {'UP': ctmrg.absorb_truncate_CTM_MOVE_UP, 'LEFT': ctmrg.absorb_truncate_CTM_MOVE_LEFT,
          'DOWN': ctmrg.absorb_truncate_CTM_MOVE_DOWN, 'RIGHT': ctmrg.absorb_truncate_CTM_MOVE_RIGHT}
    for dn, d in DIRS.items():
        R, Rt = hf[dn]((0, 0), st, env)
        assert relerr(R, g[f"R_{dn}"]) < 1e-12 and relerr(Rt, g[f"Rt_{dn}"]) < 1e-12
        P, Pt, S = eng.projectors(R, Rt, chi, return_S=True)
        assert relerr(S, g[f"S_{dn}"]) < 1e-12
        Pr, Ptr = g[f"P_{dn}"], g[f"Pt_{dn}"]
        assert relerr(P.abs(), np.abs(Pr)) < 1e-6 and relerr(Pt.abs(), np.abs(Ptr)) < 1e-6
        assert relerr(P @ Pt.t(), Pr @ Ptr.T) < 1e-6
        # biorthogonality of the kept subspace: Pt^T P = 1 on the non-zero block
        k = int((S.cpu().numpy() / S[0].item() > 1e-8).sum())
        G = (Pt.t() @ P).cpu().numpy()[:k, :k]
        assert np.abs(G - np.eye(k)).max() < 1e-7
        # absorb with the REFERENCE's projectors -> entrywise comparable with the golden outputs
        Pd = {c: dev(g[f"Pall_{dn}_{c[0]}_{c[1]}"]) for c in case["sites"]}
        Ptd = {c: dev(g[f"Ptall_{dn}_{c[0]}_{c[1]}"]) for c in case["sites"]}
        for c in case["sites"]:
            out = af[dn](c, st, env, Pd, Ptd)
            for t, nm in zip(out, ("nC1", "nC2", "nT")):
                assert relerr(t, g[f"abs_{dn}_{c[0]}_{c[1]}_{nm}"]) < 1e-11, (dn, c, nm)


def test_one_move_each_direction(case, eng):
    from ctm.generic import ctmrg
    g, chi = case["g"], case["chi"]
    for dn, d in DIRS.items():
        st, env = device_state_env(case["sites"], case["C"], case["T"], chi)
        ctmrg.ctm_MOVE(d, st, env)
        C2, T2 = env_from(g, f"move_{dn}_")
        for k in C2: assert relerr(env.C[k].abs(), np.abs(C2[k])) < 1e-7, (dn, k)
        for k in T2: assert relerr(env.T[k].abs(), np.abs(T2[k])) < 1e-7, (dn, k)


def test_rdms_and_energy(case, eng):
    from ctm.generic import rdm
    from models import j1j2
    g, chi = case["g"], case["chi"]
    st, env = device_state_env(case["sites"], case["C"], case["T"], chi)
    for c in case["sites"]:
        assert relerr(rdm.rdm2x2(c, st, env), g[f"rdm2x2_{c[0]}_{c[1]}"]) < 1e-11
    assert relerr(rdm.rdm1x1((0, 0), st, env), g["rdm1x1"]) < 1e-11
    assert relerr(rdm.rdm2x1((0, 0), st, env), g["rdm2x1"]) < 1e-11
    assert relerr(rdm.rdm1x2((0, 0), st, env), g["rdm1x2"]) < 1e-11
    e = float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))
    assert abs(e - float(g["energy_j2_0.5"])) < 1e-12 * abs(e)


@pytest.mark.parametrize("native_move", [True, False], ids=["ctm_move", "unit-by-unit"])
@pytest.mark.parametrize("warm_tol", [0.0, 1e-9], ids=["warm_tol0", "warm_tol1e-9"])
def test_converged_run(case, eng, warm_tol, native_move):
    """ctmrg.run with ctmrg_conv_specC from the CTMRG init: same number of sweeps as the reference,
    corner spectra and rdm2x2 energy within 1e-10 -- on every route of the move: the whole move as one native call (ctm_move, the
    default) or unit by unit, with and without ctm_args.projector_warm_tol (at these sizes, n < 256, every truncation is a dense solve
    and the option must change nothing; tests/test_gpu_stationary.py pins it where the fast path is taken)."""
    import config as cfg
    import copy
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env, ctmrg_conv_specC
    from ctm.generic import ctmrg
    from models import j1j2
    g, chi = case["g"], case["chi"]
    st = IPEPS({k: dev(v) for k, v in case["sites"].items()})
    env = ENV(chi, st)
    init_env(st, env)
    args = copy.deepcopy(cfg.ctm_args)
    args.ctm_max_iter = 60
    args.projector_warm_tol = warm_tol
    args.native_move = native_move
    try:
        env, hist, t_ctm, t_obs = ctmrg.run(st, env, conv_check=ctmrg_conv_specC, ctm_args=args)
    finally:
        for e_ in [eng] + list(eng.workers):
            e_.set_option("warm_accept_tol", 0.0); e_._warm_tol = 0.0
    assert len(hist['conv_crit']) == int(g["conv_nsweeps"])
    for k, s in env.get_spectra().items():
        ref = g[f"conv_spec_{k[0][0]}_{k[0][1]}_{k[1][0]}_{k[1][1]}"]
        assert np.abs(s.cpu().numpy() - ref).max() < 1e-10, k
    e = float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))
    assert abs(e - float(g["conv_energy"])) < 1e-10 * abs(e)


@pytest.mark.parametrize("name", ["chi_ramp_D3_chi16_36_f64", "chi_ramp_D2_chi6_12_c128"])
def test_chi_ramped_run_against_the_reference(eng, name):
    """ENV.extend inside a run (reference ctm/generic/env.py:164-202; how its scripts ramp the environment dimension): n0 sweeps at
    chi0 from the CTMRG init, extend(chi1), n1 more sweeps -- corner spectra and rdm2x2 energy of the end against the REFERENCE's run
    (1e-10), |C|, |T| entrywise (gauge: 1e-7), and the extended environment itself against the oracle's env_extend (exact padding)."""
    import config as cfg
    import copy
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from models import j1j2
    from oracle import ctm_oracle as O
    g = golden(name)
    chi0, chi1, n0, n1 = (int(g[k]) for k in ("chi0", "chi1", "n0", "n1"))
    sites = sites_from(g)
    st = IPEPS({k: dev(v) for k, v in sites.items()})
    env = ENV(chi0, st); init_env(st, env)

    def fixed(n):
        def count(state, env, history, ctm_args=None):
            history = (history or []) + [1]
            return len(history) >= n, history
        return count
    args = copy.deepcopy(cfg.ctm_args)
    args.ctm_max_iter = n0
    env, *_ = ctmrg.run(st, env, conv_check=fixed(n0), ctm_args=args)
    before = {k: v.cpu().numpy() for k, v in {**{("C",) + k: v for k, v in env.C.items()}, **{("T",) + k: v for k, v in env.T.items()}}.items()}
    env = env.extend(chi1)
    assert env.chi == chi1
    # the extension is the oracle's (zero padding around the leading block), checked on this run's own tensors
    oe = O.Env(chi0)
    oe.C = {k[1:]: v for k, v in before.items() if k[0] == "C"}; oe.T = {k[1:]: v for k, v in before.items() if k[0] == "T"}
    ox = O.env_extend(oe, chi1)
    for k in ox.C: assert np.array_equal(env.C[k].cpu().numpy(), ox.C[k]), k
    for k in ox.T: assert np.array_equal(env.T[k].cpu().numpy(), ox.T[k]), k
    args.ctm_max_iter = n1
    env, *_ = ctmrg.run(st, env, conv_check=fixed(n1), ctm_args=args)
    for k, s_ in env.get_spectra().items():
        assert np.abs(s_.cpu().numpy() - g[f"spec_{k[0][0]}_{k[0][1]}_{k[1][0]}_{k[1][1]}"]).max() < 1e-10, k
    C1, T1 = env_from(g, "end_")
    for k in C1: assert relerr(env.C[k].abs(), np.abs(C1[k])) < 1e-7, k
    for k in T1: assert relerr(env.T[k].abs(), np.abs(T1[k])) < 1e-7, k
    e = float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))
    assert abs(e - float(g["energy"])) <= 1e-10 * abs(float(g["energy"])), (e, float(g["energy"]))


def test_rdm2x2_partially_open(case, eng):
    """open_sites subsets = partial traces of the reference's full plaquette RDM (rdm.py:1306-1360)."""
    from ctm.generic import rdm
    g, chi = case["g"], case["chi"]
    st, env = device_state_env(case["sites"], case["C"], case["T"], chi)
    full = g["rdm2x2_0_0"]
    for os_, expr in (([0, 1], "abijefij->abef"), ([0, 3], "aijdeijh->adeh"), ([1, 2, 3], "ibcdifgh->bcdfgh"), ([2], "ijckijgk->cg")):
        ref = np.einsum(expr, full)
        ref = ref / np.trace(ref.reshape(int(np.sqrt(ref.size)), -1))
        assert relerr(rdm.rdm2x2((0, 0), st, env, open_sites=os_), ref) < 1e-11, os_


def test_transfer_matrix_correlators(case, eng):
    """ctm/generic/corrf.py: edges, one transfer step with an operator, <Sz Sz>(r) and <S+ S->(r) in all four directions
    on the native einsum vs the oracle (itself pinned against the reference by `gen_golden.py variants`)."""
    from ctm.generic import corrf
    from oracle import ctm_oracle as O, j1j2_oracle as OJ
    st, env = device_state_env(case["sites"], case["C"], case["T"], case["chi"])
    ost, oe = oracle_state_env(case["sites"], case["C"], case["T"], case["chi"])
    I2, sz, sp, sm = OJ.su2_ops(2)
    dt = next(iter(case["sites"].values())).dtype
    tt = lambda a: dev(a.astype(dt))
    for dn, d in DIRS.items():
        for c in [(0, 0), (1, 1)]:
            Eo = O.get_edge(c, d, ost, oe)
            assert relerr(corrf.get_edge(c, d, st, env), Eo) < 1e-13
            assert relerr(corrf.apply_TM_1sO(c, d, st, env, dev(Eo), op=tt(sp)), O.apply_TM_1sO(c, d, ost, oe, Eo, op=sp.astype(dt))) < 1e-12
            assert relerr(corrf.apply_TM_1sO(c, d, st, env, dev(Eo)), O.apply_TM_1sO(c, d, ost, oe, Eo)) < 1e-12
            for o1, o2 in ((sz, sz), (sp, sm)):
                cr = corrf.corrf_1sO1sO(c, d, st, env, tt(o1), lambda r: tt(o2), 3).cpu().numpy()
                cro = O.corrf_1sO1sO(c, d, ost, oe, o1.astype(dt), lambda r: o2.astype(dt), 3)
                assert np.abs(cr - cro).max() < 1e-11, (dn, c)


@pytest.mark.parametrize("name", ["aklt_S2_2x1", "aklt_S2_2x2"])
def test_aklt_S2_known_answer(eng, name):
    """examples/akltS2/ctmrg_akltS2.py:166-221,224-279 of the reference: AKLT S=2 (p=5, D=2), chi=32: E/site < 1e-12 and
    every on-site magnetisation < 1e-12, through ctmrg.run with the corner-spectrum convergence check."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env, ctmrg_conv_specC
    from ctm.generic import ctmrg
    from models import akltS2
    g = golden(name)
    sites = sites_from(g)
    v2s = (lambda c: ((((c[0] + abs(c[0]) * 2) % 2) + abs(c[1])) % 2, 0)) if name.endswith("2x1") else None
    st = IPEPS({k: dev(v) for k, v in sites.items()}, vertexToSite=v2s, lX=int(g["lX"]), lY=int(g["lY"]))
    env = ENV(32, st); init_env(st, env)
    cfg.ctm_args.ctm_max_iter = 30
    env, hist, *_ = ctmrg.run(st, env, conv_check=ctmrg_conv_specC)
    cfg.ctm_args.ctm_max_iter = 50
    assert len(hist['conv_crit']) == int(g["nsweeps"])
    m = akltS2.AKLTS2()
    assert abs(float(m.energy_2x1_1x2(st, env))) < 1e-12
    vals, labels = m.eval_obs(st, env)
    obs = dict(zip(labels, vals))
    for c in st.sites:
        assert abs(obs[f"m{c}"]) < 1e-12


def test_transfer_operator_spectrum(case, eng):
    """transferops.get_Top_spec: ARPACK on the host over native transfer steps vs the oracle (moduli: complex pairs may swap)."""
    from ctm.generic import transferops
    from oracle import ctm_oracle as O
    st, env = device_state_env(case["sites"], case["C"], case["T"], case["chi"])
    ost, oe = oracle_state_env(case["sites"], case["C"], case["T"], case["chi"])
    for d in [(1, 0), (0, -1)]:
        L = transferops.get_Top_spec(4, (0, 0), d, st, env).cpu().numpy()
        Lo = O.get_Top_spec(4, (0, 0), d, ost, oe)
        assert np.abs(np.abs(L[:, 0] + 1j * L[:, 1]) - np.abs(Lo)).max() < 1e-9
        assert abs(L[0, 0] - 1.0) < 1e-12


def test_corner_cache_gives_identical_sweeps_and_hits(eng):
    """Enlarged corners kept with the environment (half of them survive every move): the environment after two sweeps is
    bit-identical to the one computed with the cache off, in-place modification of an environment tensor invalidates."""
    import copy
    import config as cfg
    from helpers import sites_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    g = golden("generic_D3_chi18_f64")
    sites = sites_from(g)
    envs = []
    for use in (True, False):
        st = IPEPS({k: dev(v) for k, v in sites.items()})
        env = ENV(18, st); init_env(st, env)
        args = copy.deepcopy(cfg.ctm_args); args.corner_cache = use; args.projector_warm_start = False
        h0 = eng.stat("corner_cache_hits")
        for _ in range(2):
            for d in args.ctm_move_sequence:
                for _r in range(2):
                    ctmrg.ctm_MOVE(d, st, env, ctm_args=args)
        hits = eng.stat("corner_cache_hits") - h0
        # 16 moves x 4 units x 4 corners = 256 corner uses; all but the first move find half of them in the cache
        assert hits == (15 * 8 if use else 0), hits
        envs.append(env)
    for k in envs[0].C: assert torch.equal(envs[0].C[k], envs[1].C[k]), k
    for k in envs[0].T: assert torch.equal(envs[0].T[k], envs[1].T[k]), k
    # in-place change of one T tensor: the two corner types that contain it are rebuilt at that site
    env = envs[0]
    st = IPEPS({k: dev(v) for k, v in sites.items()})
    args = copy.deepcopy(cfg.ctm_args); args.projector_warm_start = False
    k0 = next(iter(env.T))
    env.T[k0].mul_(1.0)
    h0 = eng.stat("corner_cache_hits")
    ctmrg.ctm_MOVE((0, -1), st, env, ctm_args=args)
    assert eng.stat("corner_cache_hits") - h0 < 8


def test_absorb_of_nonzero_projector_prefix_is_identical(eng):
    """Masked projector columns (S/S[0] <= projector_svd_reltol) are exact zeros: absorbing only the non-zero prefix and
    padding gives the same environment, bit for bit, as multiplying the zeros through like the reference does."""
    import copy
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    rng = np.random.default_rng(12)
    sites = {(x, y): rng.random((2, 3, 3, 3, 3)) for x in range(2) for y in range(2)}      # positive tensors: low-rank environment
    envs, used = [], []
    for use in (True, False):
        st = IPEPS({k: dev(v / np.abs(v).max()) for k, v in sites.items()})
        env = ENV(48, st); init_env(st, env)
        args = copy.deepcopy(cfg.ctm_args); args.absorb_skip_zero_columns = use; args.absorb_skip_min_n = 0
        for _ in range(3):
            for d in args.ctm_move_sequence:
                for _r in range(2):
                    ctmrg.ctm_MOVE(d, st, env, ctm_args=args)
        envs.append(env)
        nc = env.__dict__.get("_ncol")
        used.append(bool(nc) and max(nc.values()) <= 24)
    assert used == [True, False]                     # the compact path really ran (<= chi/2 significant columns)
    for k in envs[0].C: assert torch.equal(envs[0].C[k], envs[1].C[k]), k
    for k in envs[0].T: assert torch.equal(envs[0].T[k], envs[1].T[k]), k


@pytest.mark.parametrize("cplx", [False, True], ids=["f64", "c128"])
@pytest.mark.parametrize("positive", [True, False], ids=["masked-columns", "full-rank"])
def test_whole_move_in_one_native_call_equals_the_unit_by_unit_move(eng, cplx, positive):
    """ctm_move (include/ctm_hip.h; reference seam ctm_MOVE_c, ctm/generic/ctmrg.py:233-283): both phases of a move and the threads
    that overlap their units inside the library -- against the host-orchestrated move (one native call per unit), three sweeps,
    bit for bit; with concurrent worker contexts and serially; with the masked-column absorb (positive tensors) and without."""
    import copy
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    rng = np.random.default_rng(31)
    mk = (lambda: rng.random((2, 3, 3, 3, 3)) - (0.0 if positive else 0.5))
    sites = {(x, y): (mk() + (1j * mk() if cplx else 0)) for x in range(2) for y in range(2)}
    envs = []
    for native, conc in ((True, True), (True, False), (False, True)):
        st = IPEPS({k: dev(v / np.abs(v).max()) for k, v in sites.items()})
        env = ENV(40, st); init_env(st, env)
        args = copy.deepcopy(cfg.ctm_args); args.native_move = native; args.concurrent_units = conc; args.absorb_skip_min_n = 0
        for _ in range(3):
            for d in args.ctm_move_sequence:
                for _r in range(2):
                    ctmrg.ctm_MOVE(d, st, env, ctm_args=args)
        envs.append(env)
    nc = envs[0].__dict__.get("_ncol")
    assert (bool(nc) and max(nc.values()) <= 20) == positive                 # the compact absorb ran exactly in the low-rank case
    for other in envs[1:]:
        for k in envs[0].C: assert torch.equal(envs[0].C[k], other.C[k]), k
        for k in envs[0].T: assert torch.equal(envs[0].T[k], other.T[k]), k


def test_a_move_that_runs_out_of_workspace_is_repeated_with_fewer_units_in_flight(eng, monkeypatch):
    """The number of units a move keeps in flight comes from an ESTIMATE of a unit's workspace (ctmrg.ctm_MOVE: one measured high-water
    mark scaled by n^2).  When the library reports CTM_ERR_NOMEM with worker contexts in use, the move -- which has written nothing
    yet -- is repeated with half the width after the arenas were given back, and the environment remembers the width that worked.
    The out-of-memory condition is injected at the C-ABI wrapper (the real one needs > 144 GB of workspace)."""
    import copy
    import config as cfg
    import _native
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    rng = np.random.default_rng(32)
    sites = {(x, y): rng.random((2, 3, 3, 3, 3)) - 0.5 for x in range(2) for y in range(2)}
    args = copy.deepcopy(cfg.ctm_args)

    def sweeps(env, st, n=2):
        for _ in range(n):
            for d in args.ctm_move_sequence:
                for _r in range(2):
                    ctmrg.ctm_MOVE(d, st, env, ctm_args=args)

    st = IPEPS({k: dev(v / np.abs(v).max()) for k, v in sites.items()})
    ref = ENV(40, st); init_env(st, ref)
    sweeps(ref, st)
    real_move, widths = type(eng).move, []

    def starved(self, direction, units, chi, cfgt, normalize=1, skip_zero_columns=False, workers=()):
        widths.append(len(workers))
        if len(workers) > 1 and len(widths) >= 3:                  # from the third move on: no room for more than one unit
            err = _native.NativeError("ctm_move: out of memory: (injected)"); err.status = _native.CTM_ERR_NOMEM
            raise err
        return real_move(self, direction, units, chi, cfgt, normalize=normalize, skip_zero_columns=skip_zero_columns, workers=workers)

    monkeypatch.setattr(type(eng), "move", starved)
    env = ENV(40, st); init_env(st, env)
    sweeps(env, st)
    assert widths[:2] == [4, 4] and widths[2:5] == [4, 2, 0], widths        # 4 in flight, refused, 2 refused, then serially: accepted
    assert set(widths[5:]) == {0} and env.__dict__["_units_cap"] == 1, widths   # ... and later moves do not try again
    for k in ref.C: assert torch.equal(ref.C[k], env.C[k]), k
    for k in ref.T: assert torch.equal(ref.T[k], env.T[k]), k
    # an error that is not a shortage of memory, or one without worker contexts to give up, is the caller's
    def broken(self, *a, **kw):
        err = _native.NativeError("ctm_move: HIP error: (injected)"); err.status = 4
        raise err
    monkeypatch.setattr(type(eng), "move", broken)
    with pytest.raises(_native.NativeError, match="injected"):
        ctmrg.ctm_MOVE((0, -1), st, env, ctm_args=args)


def test_rdm2x2_from_parts_equals_the_whole(case, eng):
    """ctm_rdm2x2_part: the plaquette contraction split over ranges of lower-half slices (what a rank group shares, and what one
    GPU loops over when the open halves do not fit) reassembles to ctm_rdm2x2 exactly; and the host layer's chunked path gives
    the same normalised RDM as the reference-pinned golden one."""
    from ctm.generic import rdm
    from ctm.generic.ctm_components import _corner_t, LU, RU, RD, LD
    st, env = device_state_env(case["sites"], case["C"], case["T"], case["chi"])
    for c in ((0, 0), (1, 1)):
        x, y = c
        t = _corner_t(LU, (x, y), st, env) + _corner_t(RU, (x + 1, y), st, env) + _corner_t(RD, (x + 1, y + 1), st, env) \
            + _corner_t(LD, (x, y + 1), st, env)
        whole = eng.rdm2x2(t)
        p = whole.shape[0]
        for bounds in ([0, 16], [0, 5, 16], [0, 1, 2, 4, 8, 16]):
            R = torch.cat([eng.rdm2x2_part(t, a, b) for a, b in zip(bounds[:-1], bounds[1:])], dim=1)
            assert relerr(eng.rdm2x2_from_parts(R, p), whole) < 1e-14, (c, bounds)
    # forced chunking through the host layer
    import unittest.mock as mock
    with mock.patch("torch.cuda.mem_get_info", return_value=(1, 1)), mock.patch.object(type(eng), "stat", lambda self, k: 0):
        r = rdm.rdm2x2((0, 0), st, env)
    assert relerr(r, rdm.rdm2x2((0, 0), st, env)) < 1e-14


@pytest.mark.parametrize("tag,chi", [("f64_D2_chi3", 3), ("f64_D2_chi6", 6), ("c128_D3_chi7", 7)])
@pytest.mark.parametrize("kind", ["PROD", "CTMRG_OBC"])
def test_env_init_variants_on_the_engine(eng, tag, chi, kind):
    """ctm_env_init_type PROD / CTMRG_OBC (reference ctm/generic/env.py:274-365, 538-716) against the reference's own output."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from helpers_cpu import env_from
    g = golden("envinit")
    sites = {tuple(int(v) for v in k.split('_')[-2:]): g[k] for k in g.files if k.startswith(tag + "_site_")}
    st = IPEPS({k: dev(v) for k, v in sites.items()}, lX=2, lY=2)
    env = ENV(chi, st)
    old = cfg.ctm_args.ctm_env_init_type
    cfg.ctm_args.ctm_env_init_type = kind
    try:
        init_env(st, env)
    finally:
        cfg.ctm_args.ctm_env_init_type = old
    C, T = env_from(g, f"{tag}_{kind}_")
    for k in C: assert relerr(env.C[k], C[k]) < 1e-13 or float(np.abs(C[k]).max()) == 0
    for k in T: assert tuple(env.T[k].shape) == T[k].shape and relerr(env.T[k], T[k]) < 1e-13


@pytest.mark.parametrize("base", ["generic_D2_chi8_f64", "generic_D2_chi8_c128"])
def test_model_correlators_against_the_reference(eng, base):
    """J1J2.eval_corrf_SS (with and without the sublattice rotation) and eval_corrf_SpSm in both directions (models/j1j2.py:477-527)."""
    from models import j1j2
    g, c = golden(base), golden("generic_corr")
    C, T = env_from(g, "warm_")
    st, env = device_state_env(sites_from(g), C, T, next(iter(C.values())).shape[0])
    model = j1j2.J1J2(j1=1.0, j2=0.5)
    for d in ((1, 0), (0, 1)):
        for conj_ in (False, True):
            r = model.eval_corrf_SS((0, 0), d, st, env, 2, conjugate=conj_)
            for k, v in r.items():
                assert float(np.abs(v.cpu().numpy() - c[f"{base}_ss_{d[0]}{d[1]}_{int(conj_)}_{k}"]).max()) < 1e-10, (d, conj_, k)
        r = model.eval_corrf_SpSm((1, 0), d, st, env, 2)
        for k, v in r.items():
            assert float(np.abs(v.cpu().numpy() - c[f"{base}_spsm_{d[0]}{d[1]}_{k}"]).max()) < 1e-10, (d, k)
