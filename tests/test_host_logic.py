"""CPU: host-side logic of the drop-in layer (state container, JSON I/O, config flags, tilings, the move
orchestration on an oracle-backed engine double)."""
import json, os
import numpy as np
import pytest
import torch
import backend
from fake_engine import FakeEngine
from helpers_cpu import sites_from, env_from
from conftest import golden


@pytest.fixture()
def cpu_cfg():
    import config as cfg
    old = cfg.global_args.device
    cfg.global_args.device = 'cpu'
    backend.set_engine(FakeEngine())
    yield cfg
    backend.set_engine(None)
    cfg.global_args.device = old


def test_vertex_to_site_default_and_pattern(cpu_cfg):
    from ipeps.ipeps import IPEPS
    t = lambda: torch.rand(2, 2, 2, 2, 2, dtype=torch.float64)
    st = IPEPS({(0, 0): t(), (1, 0): t(), (0, 1): t(), (1, 1): t()})
    assert (st.lX, st.lY) == (2, 2)
    assert st.vertexToSite((-1, -1)) == (1, 1) and st.vertexToSite((2, 3)) == (0, 1) and st.vertexToSite((-3, 0)) == (1, 0)
    st2 = IPEPS({(0, 0): t(), (1, 0): t()}, pattern=[[0, 1], [1, 0]])
    assert st2.vertexToSite((0, 1)) == (1, 0) and st2.vertexToSite((1, 1)) == (0, 0) and st2.vertexToSite((-1, 0)) == (1, 0)
    assert len(st.get_aux_bond_dims()) == 16


def test_script_tilings():
    import importlib.util, sys
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("ctmrg_j1j2_script", os.path.join(PKG, "examples", "j1j2", "ctmrg_j1j2.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    bp = m.TILINGS["BIPARTITE"]
    assert [bp((x, y)) for y in range(2) for x in range(2)] == [(0, 0), (1, 0), (1, 0), (0, 0)]
    assert m.TILINGS["4SITE"]((-1, 3)) == (1, 1)
    e8 = m.TILINGS["8SITE"]
    assert e8((0, 0)) == (0, 0) and e8((0, 2)) == (2, 0) and e8((3, 1)) == (3, 1)


def test_json_roundtrip(tmp_path, cpu_cfg):
    from ipeps.ipeps import IPEPS, read_ipeps, write_ipeps
    rng = np.random.default_rng(0)
    st = IPEPS({(0, 0): torch.from_numpy(rng.random((2, 2, 3, 2, 3))), (1, 0): torch.from_numpy(rng.random((2, 2, 3, 2, 3)))})
    f = str(tmp_path / "s.json")
    write_ipeps(st, f)
    st2 = read_ipeps(f)
    for c in st.sites:
        assert torch.equal(st.sites[c], st2.sites[c].cpu())
    assert (st2.lX, st2.lY) == (2, 1)
    # aux_seq: file stores [left, up, right, down]
    write_ipeps(st, f, aux_seq=[1, 0, 3, 2])
    st3 = read_ipeps(f, aux_seq=[1, 0, 3, 2])
    assert torch.equal(st.sites[(0, 0)], st3.sites[(0, 0)].cpu())
    # complex128 tensors: "s u l d r re im" with plain floats (a numpy scalar's repr is not one)
    stc = IPEPS({(0, 0): torch.from_numpy(rng.random((2, 2, 2, 2, 2)) + 1j * rng.random((2, 2, 2, 2, 2)))})
    write_ipeps(stc, f)
    assert "np." not in open(f).read()
    assert torch.equal(stc.sites[(0, 0)], read_ipeps(f).sites[(0, 0)].cpu())
    # legacy sparse entries "s u l d r re" with physDim/auxDim
    js = {"lX": 1, "lY": 1, "sites": [{"siteId": "A0", "physDim": 2, "auxDim": 2, "entries": ["0 0 0 0 0 1.5", "1 1 0 1 0 -2.0 0.0"]}],
          "map": [{"siteId": "A0", "x": 0, "y": 0}]}
    json.dump(js, open(f, "w"))
    s4 = read_ipeps(f)
    a = s4.sites[(0, 0)]
    assert a.shape == (2, 2, 2, 2, 2) and a[0, 0, 0, 0, 0] == 1.5 and a[1, 1, 0, 1, 0] == -2.0 and a.abs().sum() == 3.5


def test_config_flags_roundtrip(cpu_cfg):
    cfg = cpu_cfg
    p = cfg.get_args_parser()
    args = p.parse_args(["--chi", "32", "--bond_dim", "3", "--CTMARGS_ctm_max_iter", "7", "--CTMARGS_projector_svd_reltol", "1e-9",
                         "--GLOBALARGS_dtype", "float64", "--out_prefix", "/tmp/_ctm_test"])
    old = cfg.ctm_args.ctm_max_iter
    cfg.configure(args)
    assert cfg.ctm_args.ctm_max_iter == 7 and cfg.ctm_args.projector_svd_reltol == 1e-9 and cfg.main_args.chi == 32
    assert cfg.global_args.torch_dtype == torch.float64
    cfg.ctm_args.ctm_max_iter = old; cfg.ctm_args.projector_svd_reltol = 1e-8
    # defaults of the fields the hot path reads (reference config.py:370-409)
    c = cfg.CTMARGS()
    assert (c.ctm_max_iter, c.ctm_conv_tol, c.ctm_env_init_type, c.projector_method, c.projector_svd_reltol,
            c.projector_eps_multiplet, c.projector_multiplet_abstol) == (50, 1e-8, 'CTMRG', '4X4', 1e-8, 1e-8, 1e-14)
    assert c.ctm_move_sequence == [(0, -1), (-1, 0), (0, 1), (1, 0)]


def test_move_orchestration_matches_oracle(cpu_cfg):
    """ctm_MOVE / run / conv_specC / energy on the host layer (engine double) == oracle end to end."""
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env, ctmrg_conv_specC
    from ctm.generic import ctmrg
    from models import j1j2
    from oracle import ctm_oracle as O, j1j2_oracle as OJ
    cfg = cpu_cfg
    g = golden("generic_D2_chi8_f64")
    sites = sites_from(g)
    st = IPEPS({k: torch.from_numpy(v.copy()) for k, v in sites.items()})
    env = ENV(8, st)
    init_env(st, env)
    C0, T0 = env_from(g, "init_")
    for k in C0: assert np.abs(env.C[k].numpy() - C0[k]).max() < 1e-13
    cfg.ctm_args.ctm_max_iter = 60
    env, hist, t_ctm, t_obs = ctmrg.run(st, env, conv_check=ctmrg_conv_specC)
    assert len(hist['conv_crit']) == int(g["conv_nsweeps"])
    e = float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))
    assert abs(e - float(g["conv_energy"])) < 1e-10 * abs(e)
    with pytest.raises(ValueError):
        ctmrg.ctm_MOVE((1, 1), st, env)
    cfg.ctm_args.ctm_max_iter = 50


def test_error_conventions(cpu_cfg):
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    cfg = cpu_cfg
    st = IPEPS({(0, 0): torch.rand(2, 2, 2, 2, 2, dtype=torch.float64)})
    env = ENV(4, st)
    old = cfg.ctm_args.ctm_env_init_type
    cfg.ctm_args.ctm_env_init_type = "BOGUS"
    with pytest.raises(ValueError):
        init_env(st, env)
    cfg.ctm_args.ctm_env_init_type = old
    e2 = env.extend(6)
    assert e2.C[((0, 0), (-1, -1))].shape == (6, 6) and e2.T[((0, 0), (0, 1))].shape == (4, 6, 6)


@pytest.mark.parametrize("name", ["generic_D2_chi8_f64", "generic_D2_chi8_c128"])
def test_move_variants_on_host_layer(cpu_cfg, name):
    """projector_method '4X2', ctm_absorb_normalization '2', the ctm_force_dl flag and partially open rdm2x2 through the
    host layer (engine double) vs the oracle; invalid settings raise the reference's exception types."""
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg, rdm
    from oracle import ctm_oracle as O
    cfg = cpu_cfg
    g = golden(name)
    sites = sites_from(g)
    for method, norm in (("4X2", "inf"), ("4X4", "2")):
        st = IPEPS({k: torch.from_numpy(v.copy()) for k, v in sites.items()})
        env = ENV(8, st); init_env(st, env)
        ost = O.State(sites); oe = O.init_env_ctmrg(ost, 8)
        cfg.ctm_args.projector_method, cfg.ctm_args.ctm_absorb_normalization, cfg.ctm_args.ctm_force_dl = method, norm, True
        try:
            for d in cfg.ctm_args.ctm_move_sequence:
                ctmrg.ctm_MOVE(d, st, env)
                O.ctm_move(d, ost, oe, norm_type=norm, projector_method=method)
        finally:
            cfg.ctm_args.projector_method, cfg.ctm_args.ctm_absorb_normalization, cfg.ctm_args.ctm_force_dl = "4X4", "inf", False
        for k in oe.C: assert np.abs(np.abs(env.C[k].numpy()) - np.abs(oe.C[k])).max() < 1e-9, (method, norm, k)
        for k in oe.T: assert np.abs(np.abs(env.T[k].numpy()) - np.abs(oe.T[k])).max() < 1e-9, (method, norm, k)
    full = rdm.rdm2x2((0, 0), st, env).numpy()
    part = rdm.rdm2x2((0, 0), st, env, open_sites=[0, 3]).numpy()
    ref = np.einsum("aijdeijh->adeh", full)
    assert np.abs(part - ref / np.trace(ref.reshape(4, 4))).max() < 1e-13
    with pytest.raises(ValueError):
        rdm.rdm2x2((0, 0), st, env, open_sites=[0, 5])
    cfg.ctm_args.projector_method = "4X3"
    try:
        with pytest.raises(ValueError):
            ctmrg.ctm_MOVE((0, -1), st, env)
    finally:
        cfg.ctm_args.projector_method = "4X4"


REF_INPUTS = ["RVB_1x1.in", "RVB_2x1_AB.in", "RVB_2x2_ABCD.in", "VBS_1x2_AB_D2.in", "AKLT-S2_2x1_biLat.in", "AKLT-S2_2x2_ABCD.in",
              "gesdd-D2-chi50-j20.55-run0-iRND2x1_state.json", "BIPARTITE_j2_0_j3_1250_h_39000_D_3_chi_32_seed_100_state.json"]


@pytest.mark.parametrize("fname", REF_INPUTS)
def test_read_ipeps_parses_the_reference_data_files(fname, cpu_cfg):
    """The reference's own test-input files (legacy "entries" format, "1D" format, cells with a pattern) are committed as data
    under tests/golden/test-input; read_ipeps must reproduce the arrays, cell size, tiling and aux_seq handling that the
    REFERENCE's parser gives on them (oracle/gen_golden.py inputs; reference ipeps/ipeps.py:339-441, tensor_io.py:52-87)."""
    from ipeps.ipeps import read_ipeps
    from conftest import REPO
    g = golden("test_input_parsed")
    path = os.path.join(REPO, "tests", "golden", "test-input", fname)
    tag = fname.replace('.', '_').replace('-', '_')
    st = read_ipeps(path)
    assert [st.lX, st.lY] == list(g[f"{tag}__lXlY"])
    keys = [k for k in g.files if k.startswith(f"{tag}__site_")]
    assert len(keys) == len(st.sites)
    for k in keys:
        c = tuple(int(v) for v in k.split("__site_")[1].split("_"))
        ref = g[k]
        assert tuple(st.sites[c].shape) == ref.shape and np.array_equal(st.sites[c].cpu().numpy(), ref), (fname, c)
    win = [(x, y) for y in range(-3, 4) for x in range(-3, 4)]
    assert np.array_equal(np.array([st.vertexToSite(v) for v in win]), g[f"{tag}__v2s"])
    for asq in ([0, 1, 2, 3], [3, 0, 1, 2]):
        st2 = read_ipeps(path, aux_seq=asq)
        assert np.array_equal(next(iter(st2.sites.values())).cpu().numpy(), g[f"{tag}__aux{''.join(map(str, asq))}"])


def test_read_ipeps_c4v_on_the_reference_rvb_file(cpu_cfg):
    """BASELINE configs[0] reads test-input/RVB_1x1.in: D = 3 single-site state in the legacy format."""
    from ipeps.ipeps_c4v import read_ipeps_c4v
    from conftest import REPO
    g = golden("test_input_parsed")
    st = read_ipeps_c4v(os.path.join(REPO, "tests", "golden", "test-input", "RVB_1x1.in"))
    assert tuple(st.site().shape) == (2, 3, 3, 3, 3)
    assert np.array_equal(st.site().cpu().numpy(), g["RVB_1x1_in__c4v_site"])
    # the fixture of the published RVB anchor was generated from this very file
    assert np.array_equal(st.site().cpu().numpy(), golden("rvb_c4v")["site"])


def test_energy_1site_BP_includes_j3_and_chiral_terms(cpu_cfg):
    """energy_2x2_1site_BP = tr(rho_2x2 hp_rot) [+ lambda chiral term] + j3 * eval_nnnn_per_site (reference models/j1j2.py:212-221):
    the j3 and chiral contributions against the oracle's pieces on a golden one-site... (4-site golden state used as a 1x1 tiling of
    site (0,0))."""
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg, rdm
    from models import j1j2
    from oracle import ctm_oracle as O, j1j2_oracle as OJ
    g = golden("generic_D2_chi8_f64")
    A = sites_from(g)[(0, 0)]
    st = IPEPS({(0, 0): torch.from_numpy(A.copy())})
    env = ENV(8, st); init_env(st, env)
    for d in cpu_cfg.ctm_args.ctm_move_sequence:
        ctmrg.ctm_MOVE(d, st, env)
    ost = O.State({(0, 0): A}); oe = O.Env(8)
    oe.C = {k: v.numpy() for k, v in env.C.items()}; oe.T = {k: v.numpy() for k, v in env.T.items()}
    e0 = float(j1j2.J1J2(j1=1.0, j2=0.3).energy_2x2_1site_BP(st, env))
    e3 = float(j1j2.J1J2(j1=1.0, j2=0.3, j3=0.7).energy_2x2_1site_BP(st, env))
    nnnn = OJ.eval_nnnn_per_site(lambda cc, d, o1, g2, dist: O.corrf_1sO1sO(cc, d, ost, oe, o1, g2, dist), (0, 0))
    assert abs((e3 - e0) - 0.7 * float(np.real(nnnn))) < 1e-12
    # rotated plaquette term itself against the oracle
    r = O.rdm2x2((0, 0), ost, oe)
    assert abs(e0 - OJ.energy_1x1(r, 1.0, 0.3)) < 1e-12
    # the chiral term needs a complex dtype (models/j1j2.py:97-98)
    with pytest.raises(AssertionError):
        j1j2.J1J2(j1=1.0, lmbd=0.5)


def test_energy_per_site_includes_the_chiral_term(cpu_cfg):
    """energy_per_site adds lmbd * tr(rho_2x2 chiral_term) per plaquette with the UN-rotated term (reference models/j1j2.py:236-247);
    the expected number was produced with the reference's own model tensors (oracle/gen_golden.py chiral_case)."""
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    from models import j1j2
    g, gc = golden("generic_D2_chi8_c128"), golden("chiral")
    old = cpu_cfg.global_args.torch_dtype
    cpu_cfg.global_args.torch_dtype = torch.complex128
    try:
        st = IPEPS({k: torch.from_numpy(v.copy()) for k, v in sites_from(g).items()})
        env = ENV(8, st)
        C, T = env_from(g, "warm_")
        env.C = {k: torch.from_numpy(v.copy()) for k, v in C.items()}; env.T = {k: torch.from_numpy(v.copy()) for k, v in T.items()}
        model = j1j2.J1J2(j1=1.0, j2=0.5, lmbd=float(gc["lmbd"]))
        assert np.abs(model.chiral_term.numpy() - gc["chiral_term"]).max() < 1e-15
        e = float(model.energy_per_site(st, env))
        e0 = float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))
    finally:
        cpu_cfg.global_args.torch_dtype = old
    assert abs(e - float(gc["energy_j2_05_lmbd_03"])) < 1e-11
    assert abs(e0 - float(g["energy_j2_0.5"])) < 1e-11 and abs(e - e0) > 1e-6       # the term is there and is not negligible


def test_config0_script_prints_the_published_final_line(cpu_cfg, tmp_path, capsys, monkeypatch):
    """BASELINE configs[0] through the host layer (script main() in process, engine double): `--bond_dim 2 --chi 16 --instate
    test-input/RVB_1x1.in` -> `FINAL -0.5901859430133278, ..., -0.29471077912392146` (the reference's own output)."""
    import importlib.util, conftest
    monkeypatch.chdir(tmp_path)
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test-input", "RVB_1x1.in")
    spec = importlib.util.spec_from_file_location("ctmrg_j1j2_c4v_script", os.path.join(conftest.PKG, "examples", "j1j2", "ctmrg_j1j2_c4v.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    old = (cpu_cfg.ctm_args.ctm_max_iter,)
    try:
        mod.main(["--bond_dim", "2", "--chi", "16", "--instate", f, "--GLOBALARGS_device", "cpu", "--out_prefix", str(tmp_path / "o")])
    finally:
        cpu_cfg.ctm_args.ctm_max_iter, = old
        cpu_cfg.global_args.device = 'cpu'
    final = [l for l in capsys.readouterr().out.splitlines() if l.startswith("FINAL")]
    assert len(final) == 1
    vals = [float(v) for v in final[0][6:].split(",")]
    assert abs(vals[0] - (-0.5901859430133278)) < 1e-11 and abs(vals[5] - (-0.29471077912392146)) < 1e-11


def test_corrf_with_user_supplied_boundary_edges(cpu_cfg):
    """corrf_1sO1sO(..., rl_0=(right, left)) (reference corrf.py:980-1067): with the corner-T-corner edges passed explicitly the
    result equals the default; with rescaled edges it is unchanged (the ratio E12/E00 is scale invariant)."""
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg, corrf
    from oracle import j1j2_oracle as OJ
    g = golden("generic_D2_chi8_f64")
    st = IPEPS({k: torch.from_numpy(v.copy()) for k, v in sites_from(g).items()})
    env = ENV(8, st); init_env(st, env)
    for d in cpu_cfg.ctm_args.ctm_move_sequence:
        ctmrg.ctm_MOVE(d, st, env)
    I, sz, sp, sm = (torch.from_numpy(x) for x in OJ.su2_ops(2))
    for d in ((1, 0), (0, 1)):
        rev = (-d[0], -d[1])
        ref = corrf.corrf_1sO1sO((0, 0), d, st, env, sz, lambda r: sz, 3)
        rl = (lambda c: corrf.get_edge(c, rev, st, env), lambda c: 2.5 * corrf.get_edge(c, d, st, env))
        got = corrf.corrf_1sO1sO((0, 0), d, st, env, sz, lambda r: sz, 3, rl_0=rl)
        assert float((got - ref).abs().max()) < 1e-13


# ---- environment initialisations PROD / CTMRG_OBC (reference ctm/generic/env.py:274-365, 538-716) ---------------------------------
ENVINIT_CASES = [("f64_D2_chi3", 3), ("f64_D2_chi6", 6), ("c128_D3_chi7", 7)]


def _envinit_sites(g, tag):
    return {tuple(int(v) for v in k.split('_')[-2:]): g[k] for k in g.files if k.startswith(tag + "_site_")}


@pytest.mark.parametrize("tag,chi", ENVINIT_CASES)
@pytest.mark.parametrize("kind", ["PROD", "CTMRG_OBC"])
def test_env_init_variants_oracle_vs_reference(tag, chi, kind):
    from oracle import ctm_oracle as O
    from helpers_cpu import env_from
    g = golden("envinit")
    ost = O.State(_envinit_sites(g, tag))
    oe = (O.init_env_prod if kind == "PROD" else O.init_env_obc)(ost, chi)
    C, T = env_from(g, f"{tag}_{kind}_")
    assert len(C) == 16 and len(T) == 16
    for k in C: assert abs(oe.C[k] - C[k]).max() < 1e-13
    for k in T: assert oe.T[k].shape == T[k].shape and abs(oe.T[k] - T[k]).max() < 1e-13


@pytest.mark.parametrize("tag,chi", ENVINIT_CASES)
@pytest.mark.parametrize("kind", ["PROD", "CTMRG_OBC"])
def test_env_init_variants_host_layer(cpu_cfg, tag, chi, kind):
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from helpers_cpu import env_from
    g = golden("envinit")
    st = IPEPS({k: torch.from_numpy(v.copy()) for k, v in _envinit_sites(g, tag).items()}, lX=2, lY=2)
    env = ENV(chi, st)
    old = cpu_cfg.ctm_args.ctm_env_init_type
    cpu_cfg.ctm_args.ctm_env_init_type = kind
    try:
        init_env(st, env)
    finally:
        cpu_cfg.ctm_args.ctm_env_init_type = old
    C, T = env_from(g, f"{tag}_{kind}_")
    for k in C: assert float((env.C[k] - torch.from_numpy(C[k])).abs().max()) < 1e-13
    for k in T: assert tuple(env.T[k].shape) == T[k].shape and float((env.T[k] - torch.from_numpy(T[k])).abs().max()) < 1e-13


def test_bench_metric_line_fits_the_driver_record():
    """bench.py prints ONE stdout line; the driver keeps its last 8081 characters.  The line built from a committed detail of the default
    command (profiles/r04_bench_final_detail.json) stays well below that and carries the contract keys plus both states' scalars in `roofline`."""
    import importlib.util, json, os, sys
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    d = json.loads(open(os.path.join(REPO, "profiles", "r04_bench_final_detail.json")).read())
    line = b.metric_line(d)
    assert len(json.dumps(line)) < 7400
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "full_rank_value", "full_rank_ms_per_step", "full_rank_frac", "full_rank_frac_union",
              "executed_flop_per_sweep", "sweep_mfma_frac", "full_rank_sweep_mfma_frac"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert set(line["other_configs"]) >= {"c4v_D4_chi64", "generic_D6_chi128_signed", "generic_D8_chi384_c128_signed"}
