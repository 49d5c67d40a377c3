"""CPU: the oracle reproduces the committed golden vectors (which were produced by the REAL reference in
oracle/gen_golden.py) -- this is what pins the oracle on machines where /root/reference does not exist."""
import numpy as np
import pytest
from conftest import golden
from helpers_cpu import sites_from, env_from
from oracle import ctm_oracle as O, c4v_oracle as O4, j1j2_oracle as OJ

DIRS = {'UP': (0, -1), 'LEFT': (-1, 0), 'DOWN': (0, 1), 'RIGHT': (1, 0)}


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("name,chi", [("generic_D2_chi8_f64", 8), ("generic_D3_chi18_f64", 18), ("generic_D2_chi8_c128", 8)])
def test_generic_oracle_vs_reference_vectors(name, chi):
    g = golden(name)
    sites = sites_from(g)
    ost = O.State(sites)
    e0 = O.init_env_ctmrg(ost, chi)
    C0, T0 = env_from(g, "init_")
    for k in C0: assert rel(e0.C[k], C0[k]) < 1e-13
    for k in T0: assert rel(e0.T[k], T0[k]) < 1e-13
    C, T = env_from(g, "warm_")
    oe = O.Env(chi); oe.C = C; oe.T = T
    for cid in range(4):
        assert rel(O.c2x2(cid, (0, 0), ost, oe), g[f"c2x2_{cid}"]) < 1e-12
    for dn, d in DIRS.items():
        R, Rt = O.halves(d, (0, 0), ost, oe)
        assert rel(R, g[f"R_{dn}"]) < 1e-12 and rel(Rt, g[f"Rt_{dn}"]) < 1e-12
        P, Pt, S = O.projectors_from_matrices(R, Rt, chi, return_S=True)
        assert rel(S, g[f"S_{dn}"]) < 1e-12
        assert rel(np.abs(P), np.abs(g[f"P_{dn}"])) < 1e-6
        assert rel(P @ Pt.T, g[f"P_{dn}"] @ g[f"Pt_{dn}"].T) < 1e-6
        Pd = {c: g[f"Pall_{dn}_{c[0]}_{c[1]}"] for c in sites}
        Ptd = {c: g[f"Ptall_{dn}_{c[0]}_{c[1]}"] for c in sites}
        for c in sites:
            for t, nm in zip(O.absorb_truncate(d, c, ost, oe, Pd, Ptd), ("nC1", "nC2", "nT")):
                assert rel(t, g[f"abs_{dn}_{c[0]}_{c[1]}_{nm}"]) < 1e-12
    rd = [O.rdm2x2(c, ost, oe) for c in sites]
    for c, r in zip(sites, rd):
        assert rel(r, g[f"rdm2x2_{c[0]}_{c[1]}"]) < 1e-10
    assert abs(OJ.energy_per_site(rd, 1.0, 0.5) - float(g["energy_j2_0.5"])) < 1e-12
    assert rel(O.rdm1x1((0, 0), ost, oe), g["rdm1x1"]) < 1e-10
    assert rel(O.rdm2x1((0, 0), ost, oe), g["rdm2x1"]) < 1e-10
    assert rel(O.rdm1x2((0, 0), ost, oe), g["rdm1x2"]) < 1e-10


def test_generic_oracle_converged_run():
    g = golden("generic_D2_chi8_f64")
    sites = sites_from(g)
    ost = O.State(sites)
    oe = O.init_env_ctmrg(ost, 8)
    hist = None
    for _ in range(60):
        O.ctm_sweep(ost, oe)
        done, hist = O.conv_specC(oe, hist, tol=1e-8, max_iter=60)
        if done:
            break
    assert len(hist['conv_crit']) == int(g["conv_nsweeps"])
    e = OJ.energy_per_site([O.rdm2x2(c, ost, oe) for c in sites], 1.0, 0.5)
    assert abs(e - float(g["conv_energy"])) < 1e-10 * abs(e)


def test_decompositions():
    g = golden("decomp")
    for nm in "abc":
        U, S, V = O.truncated_svd_gesdd(g[f"svd_{nm}_M"], 8, keep_multiplets=True, eps_multiplet=1e-8)
        assert np.abs(S - g[f"svd_{nm}_S"]).max() < 1e-13
        assert ((S == 0) == (g[f"svd_{nm}_S"] == 0)).all()
    assert (g["svd_b_S"] == 0).sum() == 2          # multiplet back-off dropped the cut 4-fold multiplet
    assert (g["svd_c_S"] == 0).sum() == 3          # rank 5 < chi 8
    for nm, ch in (("a", 4), ("b", 6)):
        D, U = O.truncated_eig_sym(g["eig_H"], ch, keep_multiplets=True)
        assert np.abs(D - g[f"eig_{nm}_D"]).max() < 1e-13


@pytest.mark.parametrize("name", ["c4v_D2_chi8", "c4v_D3_chi18"])
def test_c4v_oracle(name):
    g = golden(name)
    A, C, T = g["site"], g["warm_C"], g["warm_T"]
    assert rel(O4.c2x2_sl(A, C, T), g["c2x2"]) < 1e-12
    nC, nT = O4.ctm_move_sl(A, C, T)
    assert rel(np.diag(nC), np.diag(g["move_C"])) < 1e-10
    assert rel(np.abs(nT), np.abs(g["move_T"])) < 1e-8
    for nm, f in (("rdm2x1", O4.rdm2x1_sl), ("rdmNN", O4.rdm2x2_NN_lowmem_sl), ("rdmNNN", O4.rdm2x2_NNN_lowmem_sl), ("rdm2x2", O4.rdm2x2)):
        assert rel(f(A, C, T, sym_pos_def=True), g[nm]) < 1e-10
    assert abs(OJ.energy_1x1_lowmem(g["rdmNN"], g["rdmNNN"], 1.0, 0.5) - float(g["e_lowmem"])) < 1e-12
    assert abs(OJ.energy_1x1(g["rdm2x2"], 1.0, 0.5) - float(g["e_2x2"])) < 1e-12


def test_rvb_known_answer_oracle():
    """The reference's own known-answer test (examples/j1j2/ctmrg_j1j2_c4v.py:254-256)."""
    g = golden("rvb_c4v")
    A = g["site"]
    C, T = O4.init_env_ctmrg(A, 16)
    for _ in range(int(g["nsweeps"])):
        C, T = O4.ctm_move_sl(A, C, T)
    e = OJ.energy_1x1_lowmem(O4.rdm2x2_NN_lowmem_sl(A, C, T, True), O4.rdm2x2_NNN_lowmem_sl(A, C, T, True), 1.0, 0.5)
    assert abs(e - (-0.47684229)) < 1e-8
    assert (np.diag(C) == 0).sum() == 3


def test_two_site_golden_state_energy():
    """examples/j1j2/ctmrg_j1j2.py:258-266: 2SITE D=2 chi=32 j2=0.55 -> E = -0.4434603770143078 (tol 1e-6)."""
    g = golden("twosite_D2_chi32")
    sites = sites_from(g)
    v2s = lambda c: ((c[0] + abs(c[0]) * 2) % 2, 0)
    ost = O.State(sites, lX=int(g["lX"]), lY=int(g["lY"]), vertexToSite=v2s)
    oe = O.init_env_ctmrg(ost, 32)
    for _ in range(int(g["nsweeps"])):
        O.ctm_sweep(ost, oe)
    e = OJ.energy_per_site([O.rdm2x2(c, ost, oe) for c in ost.sites], 1.0, 0.55)
    assert abs(e - (-0.4434603770143078)) < 1e-6
    assert abs(e - float(g["energy"])) < 1e-10


def test_bipartite_golden_state_energy_with_j3_and_field():
    """examples/j1j2/ctmrg_j1j2.py:248-257: BIPARTITE D=3 chi=32, j3=0.125, h_uni=[3.9,0,0] -> E = -1.3896897615463615
    (tol 1e-6); the j3 term comes from the distance-2 transfer-matrix correlators (models/j1j2.py:27-44)."""
    g = golden("bipartite_D3_chi32")
    sites = sites_from(g)
    v2s = lambda c: ((((c[0] + abs(c[0]) * 2) % 2) + abs(c[1])) % 2, 0)
    ost = O.State(sites, lX=int(g["lX"]), lY=int(g["lY"]), vertexToSite=v2s)
    oe = O.init_env_ctmrg(ost, 32)
    for _ in range(int(g["nsweeps"])):
        O.ctm_sweep(ost, oe)
    h_uni = tuple(float(x) for x in g["h_uni"]); j3 = float(g["j3"])
    corr = lambda c, d, o1, g2, dist: O.corrf_1sO1sO(c, d, ost, oe, o1, g2, dist)
    e = 0.0
    for c in ost.sites:
        e += np.einsum('ijklabcd,ijklabcd', O.rdm2x2(c, ost, oe), OJ.get_hp(1.0, 0.0, h_uni=h_uni, coord=c))
        e += j3 * OJ.eval_nnnn_per_site(corr, (0, 0))
    e = float(np.real(e)) / len(ost.sites)
    assert abs(e - (-1.3896897615463615)) < 1e-6
    assert abs(e - float(g["energy"])) < 1e-10


@pytest.mark.parametrize("name", ["aklt_S2_2x1", "aklt_S2_2x2"])
def test_aklt_S2_energy_is_zero(name):
    """examples/akltS2/ctmrg_akltS2.py:166-221,224-279: the AKLT S=2 state is the exact ground state of the projector
    Hamiltonian, E/site < 1e-12 (physical dimension 5; energy from rdm2x1 / rdm1x2)."""
    g = golden(name)
    sites = sites_from(g)
    v2s = (lambda c: ((((c[0] + abs(c[0]) * 2) % 2) + abs(c[1])) % 2, 0)) if name.endswith("2x1") else None
    ost = O.State(sites, lX=int(g["lX"]), lY=int(g["lY"]), vertexToSite=v2s)
    oe = O.init_env_ctmrg(ost, 32)
    for _ in range(int(g["nsweeps"])):
        O.ctm_sweep(ost, oe)
    h = g["h"]
    e = sum(np.einsum('ijab,ijab', O.rdm2x1(c, ost, oe), h) + np.einsum('ijab,ijab', O.rdm1x2(c, ost, oe), h) for c in ost.sites) / len(ost.sites)
    assert abs(e) < 1e-12


def test_oracle_svd_symeig_identities():
    """truncated_svd_symeig restates linalg/svd_symeig.py:12-34 + custom_svd.py:143-208; the reference function needs torch.symeig
    (removed from torch), so the restatement is pinned by the reference's own assertion (svd_symeig.py:88: |M - U S V^T| <
    S0 m^2 1e-14) and by its relation to truncated_eig_sym, which IS pinned against the reference (decomp.npz)."""
    from oracle import ctm_oracle as O
    g = golden("decomp")
    H = g["eig_H"]
    m = H.shape[0]
    U, S, V = O.truncated_svd_symeig(H, m)
    assert np.linalg.norm(H - (U * S) @ V.T) < S[0] * m * m * 1e-14
    assert (np.diff(S) <= 0).all()
    D, Ue = O.truncated_eig_sym(H, m)
    assert np.array_equal(S, np.abs(D)) and np.array_equal(U, Ue) and np.array_equal(V, Ue * np.sign(D)[None, :])
    for nm, ch in (("a", 4), ("b", 6)):             # the multiplet back-off cases of the eig golden
        Ut, St, Vt = O.truncated_svd_symeig(H, ch, keep_multiplets=True, eps_multiplet=1e-12)
        assert np.abs(St - np.abs(g[f"eig_{nm}_D"])).max() < 1e-13 and ((St == 0) == (g[f"eig_{nm}_D"] == 0)).all()


def test_oracle_backward_matches_reference_golden():
    """oracle.svd_backward / eigh_backward vs the reference's SVDGESDD.backward / SYMEIG.backward outputs (tests/golden/backward.npz,
    generated by oracle/gen_golden.py backward)."""
    from oracle import ctm_oracle as O
    g = golden("backward")
    for tag in ("sq_f64", "thin_f64", "sq_c128", "thin_c128"):
        a = {nm: g[f"svd_{tag}_{nm}"] for nm in ("U", "S", "V", "gU", "gS", "gV", "dA")}
        dA = O.svd_backward(a["U"], a["S"], a["V"], a["gU"], a["gS"], a["gV"], 1e-12)
        assert np.abs(dA - a["dA"]).max() < 1e-12 * np.abs(a["dA"]).max(), tag
    for tag in ("f64", "c128"):
        a = {nm: g[f"eig_{tag}_{nm}"] for nm in ("D", "U", "gD", "gU", "dA")}
        assert np.abs(O.eigh_backward(a["D"], a["U"], a["gD"], a["gU"], 1e-12) - a["dA"]).max() < 1e-12 * np.abs(a["dA"]).max(), tag


@pytest.mark.parametrize("tag,chi", [("c4v_f64_D2_chi3", 3), ("c4v_f64_D3_chi12", 12), ("c4v_c128_D2_chi6", 6)])
def test_c4v_env_init_variants_oracle_vs_reference(tag, chi):
    """init_prod / init_from_ipeps_obc of the one-site C4v environment (env_c4v.py:215-246, 315-355)."""
    from oracle import c4v_oracle as O4
    g = golden("envinit")
    A = g[f"{tag}_site"]
    C, T = O4.init_env_obc(A, chi)
    assert abs(C - g[f"{tag}_CTMRG_OBC_C"]).max() < 1e-13 and abs(T - g[f"{tag}_CTMRG_OBC_T"]).max() < 1e-13
    C, T = O4.init_env_prod(A, chi)
    Tr = g[f"{tag}_PROD_T"]
    ph = np.vdot(T[0, 0, :], Tr[0, 0, :]); ph = ph / abs(ph)                   # eigenvector phase
    assert abs(C - g[f"{tag}_PROD_C"]).max() < 1e-13 and abs(T * ph - Tr).max() < 1e-12


@pytest.mark.parametrize("base", ["c4v_D2_chi8", "c4v_D3_chi18", "c4v_D2_chi8_c128"])
def test_c4v_rdm3x1_oracle_vs_reference(base):
    g, j = golden(base), golden("c4v_j3")
    r = O4.rdm3x1_sl(g["site"], g["warm_C"], g["warm_T"], sym_pos_def=True)
    assert abs(r - j[f"{base}_rdm3x1"]).max() < 1e-10


@pytest.mark.parametrize("base", ["c4v_D2_chi8", "c4v_D3_chi18", "c4v_D2_chi8_c128"])
def test_c4v_rdm1x1_and_row_correlator_oracle_vs_reference(base):
    g, j = golden(base), golden("c4v_j3")
    A, C, T = g["site"], g["warm_C"], g["warm_T"]
    assert abs(O4.rdm1x1(A, C, T) - j[f"{base}_rdm1x1"]).max() < 1e-12
    sz = np.diag([0.5, -0.5]).astype(A.dtype)
    assert abs(O4.corrf_1sO1sO(A, C, T, sz, lambda r: sz, 3) - j[f"{base}_corr_szsz_plain"]).max() < 1e-11


@pytest.mark.parametrize("name", ["chi_ramp_D2_chi6_12_c128", "chi_ramp_D3_chi16_36_f64"])
def test_oracle_chi_ramped_run_vs_reference(name):
    """oracle.env_extend (reference ENV.extend, ctm/generic/env.py:164-202) inside a run: n0 sweeps at chi0, extend(chi1), n1 sweeps --
    corner spectra and rdm2x2 energy of the REFERENCE's run (tests/golden/chi_ramp_*.npz, oracle/gen_golden.py chi_ramp)."""
    g = golden(name)
    chi0, chi1, n0, n1 = (int(g[k]) for k in ("chi0", "chi1", "n0", "n1"))
    sites = sites_from(g)
    ost = O.State(sites)
    oe = O.init_env_ctmrg(ost, chi0)
    for _ in range(n0): O.ctm_sweep(ost, oe)
    oe = O.env_extend(oe, chi1)
    assert oe.chi == chi1 and all(c.shape == (chi1, chi1) for c in oe.C.values())
    for _ in range(n1): O.ctm_sweep(ost, oe)
    so = O.corner_spectra(oe)
    for k, s in so.items():
        assert np.abs(s - g[f"spec_{k[0][0]}_{k[0][1]}_{k[1][0]}_{k[1][1]}"]).max() < 1e-10, k
    e = OJ.energy_per_site([O.rdm2x2(c, ost, oe) for c in sites], 1.0, 0.5)
    assert abs(e - float(g["energy"])) < 1e-10 * abs(float(g["energy"]))


@pytest.mark.parametrize("name", ["rect_cut_chi5_f64", "rect_cut_chi5_c128"])
def test_oracle_with_bond_dimensions_that_differ_along_one_cut(name):
    """Rectangular halves (the reference only asserts R.shape == Rt.shape, ctm_projectors.py:209): the oracle's moves and sweeps against the
    REFERENCE's (tests/golden/rect_cut_*.npz, oracle/gen_golden.py rect_cut)."""
    g = golden(name)
    chi, nsweeps = int(g["chi"]), int(g["nsweeps"])
    ost = O.State(sites_from(g))
    for dn, d in DIRS.items():
        oe = O.init_env_ctmrg(ost, chi)
        O.ctm_move(d, ost, oe)
        C1, T1 = env_from(g, f"move_{dn}_")
        for k in C1: assert rel(np.abs(oe.C[k]), np.abs(C1[k])) < 1e-8, (dn, k)
        for k in T1: assert rel(np.abs(oe.T[k]), np.abs(T1[k])) < 1e-8, (dn, k)
    oe = O.init_env_ctmrg(ost, chi)
    for _ in range(nsweeps): O.ctm_sweep(ost, oe)
    for k, s in O.corner_spectra(oe).items():
        assert np.abs(s - g[f"spec_{k[0][0]}_{k[0][1]}_{k[1][0]}_{k[1][1]}"]).max() < 1e-10, k
