"""GPU: the drop-in example scripts as a user runs them (subprocess, CLI flags of the reference), end to end through
JSON state I/O, ctmrg.run with the corner-spectrum convergence check, and the FINAL observables line."""
import os, subprocess, sys
import numpy as np
import pytest
import torch
from conftest import golden, PKG

pytestmark = pytest.mark.gpu


def _run(script, args):
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(PKG, "examples", "j1j2", script)] + args, capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    final = [l for l in r.stdout.splitlines() if l.startswith("FINAL")]
    assert len(final) == 1, r.stdout[-2000:]
    _run.stdout = r.stdout
    return [float(v) for v in final[0][6:].split(",")]


def test_ctmrg_j1j2_script_2site_golden(tmp_path):
    """examples/j1j2/ctmrg_j1j2.py:258-266 of the reference: 2SITE D=2 chi=32 j2=0.55 -> E = -0.4434603770143078 (tol 1e-6);
    the state file is written here from the committed fixture with the build's own write_ipeps."""
    from ipeps.ipeps import IPEPS, write_ipeps
    g = golden("twosite_D2_chi32")
    st = IPEPS({(0, 0): torch.from_numpy(g["site_0_0"]), (1, 0): torch.from_numpy(g["site_1_0"])}, lX=2, lY=1)
    f = str(tmp_path / "twosite.json")
    write_ipeps(st, f)
    vals = _run("ctmrg_j1j2.py", ["--instate", f, "--tiling", "2SITE", "--chi", "32", "--bond_dim", "2", "--j2", "0.55",
                                  "--CTMARGS_ctm_max_iter", "50", "--GLOBALARGS_device", "cuda:0", "--out_prefix", str(tmp_path / "o")])
    assert abs(vals[0] - (-0.4434603770143078)) < 1e-6
    assert abs(vals[0] - float(g["energy"])) < 1e-9


INPUTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test-input")


def test_ctmrg_j1j2_c4v_script_config0_literal(tmp_path):
    """BASELINE configs[0], literally: `ctmrg_j1j2_c4v.py --bond_dim 2 --chi 16 --instate test-input/RVB_1x1.in` (the reference's own
    data file; j2 = 0, 50 sweeps) prints `FINAL -0.5901859430133278, ..., -0.29471077912392146` (reference
    examples/j1j2/ctmrg_j1j2_c4v.py:38-193)."""
    vals = _run("ctmrg_j1j2_c4v.py", ["--bond_dim", "2", "--chi", "16", "--instate", os.path.join(INPUTS, "RVB_1x1.in"),
                                      "--GLOBALARGS_device", "cuda:0", "--out_prefix", str(tmp_path / "o")])
    assert len(vals) == 6
    assert abs(vals[0] - (-0.5901859430133278)) < 1e-10
    assert abs(vals[5] - (-0.29471077912392146)) < 1e-10
    assert all(abs(v) < 1e-12 for v in vals[1:5])                      # m, sz, sp, sm of the RVB state


def test_ctmrg_j1j2_script_reads_the_reference_state_files(tmp_path):
    """The committed copies of the reference's own JSON state files fed to ctmrg_j1j2.py directly (no re-writing through this build's
    write_ipeps): the 2SITE and BIPARTITE golden energies of examples/j1j2/ctmrg_j1j2.py:248-266."""
    vals = _run("ctmrg_j1j2.py", ["--instate", os.path.join(INPUTS, "gesdd-D2-chi50-j20.55-run0-iRND2x1_state.json"), "--tiling", "2SITE",
                                  "--chi", "32", "--bond_dim", "2", "--j2", "0.55", "--CTMARGS_ctm_max_iter", "50",
                                  "--GLOBALARGS_device", "cuda:0", "--out_prefix", str(tmp_path / "o2")])
    assert abs(vals[0] - (-0.4434603770143078)) < 1e-6
    assert abs(vals[0] - float(golden("twosite_D2_chi32")["energy"])) < 1e-9
    vals = _run("ctmrg_j1j2.py", ["--instate", os.path.join(INPUTS, "BIPARTITE_j2_0_j3_1250_h_39000_D_3_chi_32_seed_100_state.json"),
                                  "--tiling", "BIPARTITE", "--chi", "32", "--bond_dim", "3", "--j3", "0.125", "--h_uni", "3.9", "0", "0",
                                  "--CTMARGS_ctm_max_iter", "100", "--GLOBALARGS_device", "cuda:0", "--out_prefix", str(tmp_path / "o3")])
    assert abs(vals[0] - (-1.3896897615463615)) < 1e-6


def test_ctmrg_j1j2_c4v_script_rvb(tmp_path):
    """examples/j1j2/ctmrg_j1j2_c4v.py:218-260 of the reference (TestRVB): E = -0.47684229 +- 1e-8."""
    from ipeps.ipeps_c4v import IPEPS_C4V
    g = golden("rvb_c4v")
    f = str(tmp_path / "rvb.json")
    IPEPS_C4V(torch.from_numpy(g["site"])).write_to_file(f, symmetrize=False)
    vals = _run("ctmrg_j1j2_c4v.py", ["--instate", f, "--chi", "16", "--bond_dim", "3", "--j2", "0.5", "--CTMARGS_ctm_max_iter", "200",
                                      "--GLOBALARGS_device", "cuda:0", "--out_prefix", str(tmp_path / "o"),
                                      "--corrf_r", "2", "--corrf_dd_v", "--top2"])
    assert abs(vals[0] - (-0.47684229)) < 1e-8
    out = _run.stdout                                           # the optional width-2 sections of the reference script (:166-183)
    assert "DD_v r dd" in out and "spectrum(T2)" in out
    t2 = out[out.index("spectrum(T2)"):].splitlines()[1].split()
    assert t2[0] == "0" and abs(float(t2[1]) - 1.0) < 1e-12


def test_ctmrg_j1j2_script_bipartite_golden(tmp_path):
    """examples/j1j2/ctmrg_j1j2.py:248-257 of the reference: BIPARTITE D=3 chi=32, j3=0.125, h_uni=[3.9,0,0] ->
    E = -1.3896897615463615 (tol 1e-6): field terms in the plaquette operator + the j3 term from the native distance-2
    transfer-matrix correlators."""
    from ipeps.ipeps import IPEPS, write_ipeps
    g = golden("bipartite_D3_chi32")
    st = IPEPS({(0, 0): torch.from_numpy(g["site_0_0"]), (1, 0): torch.from_numpy(g["site_1_0"])}, lX=2, lY=1)
    f = str(tmp_path / "bipartite.json")
    write_ipeps(st, f)
    vals = _run("ctmrg_j1j2.py", ["--instate", f, "--tiling", "BIPARTITE", "--chi", "32", "--bond_dim", "3", "--j3", "0.125",
                                  "--h_uni", "3.9", "0", "0", "--CTMARGS_ctm_max_iter", "100", "--GLOBALARGS_device", "cuda:0",
                                  "--out_prefix", str(tmp_path / "o")])
    assert abs(vals[0] - (-1.3896897615463615)) < 1e-6
    # FINAL observables of the published line (m, m_A, m_B, ...): first four values
    for v, ref in zip(vals[1:4], (0.4884474386344192, 0.48844697363007333, 0.4884479036387651)):
        assert abs(v - ref) < 1e-6


def test_optim_j1j2_c4v_script(tmp_path):
    """examples/j1j2/optim_j1j2_c4v.py:179-200 of the reference (TestOpt: D = 2, chi = 16, three epochs from a random tensor): the
    script runs the optimisation through the differentiable native path, prints one line per epoch, writes the best state and a
    checkpoint, and ends with the observables of the best state; the energy goes down."""
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    pre = str(tmp_path / "opt")
    r = subprocess.run([sys.executable, os.path.join(PKG, "examples", "j1j2", "optim_j1j2_c4v.py"), "--bond_dim", "2", "--chi", "16",
                        "--opt_max_iter", "3", "--seed", "123", "--CTMARGS_ctm_max_iter", "20", "--GLOBALARGS_device", "cuda:0",
                        "--out_prefix", pre], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [l.split(", ") for l in r.stdout.splitlines() if l[:1].isdigit() or l.startswith("-1, ")]
    e = {int(x[0]): float(x[1]) for x in rows}
    assert set(e) >= {-1, 1, 2, 3}, r.stdout[-2000:]
    assert e[3] <= e[1] + 1e-12 and e[3] < -0.4
    assert os.path.exists(pre + "_state.json") and os.path.exists(pre + "_checkpoint.p")


def test_optim_j1j2_script(tmp_path):
    """examples/j1j2/optim_j1j2.py of the reference (TestOptim_*: a few epochs from a random state on a small cell): BIPARTITE
    tiling, D = 2, chi = 8, three epochs; one line per epoch, best state and checkpoint on disk, the energy goes down; then the
    checkpoint resumes for one more epoch."""
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    pre = str(tmp_path / "opt")
    base = [sys.executable, os.path.join(PKG, "examples", "j1j2", "optim_j1j2.py"), "--tiling", "BIPARTITE", "--bond_dim", "2", "--chi", "8",
            "--seed", "123", "--CTMARGS_ctm_max_iter", "10", "--GLOBALARGS_device", "cuda:0", "--j2", "0.2"]
    r = subprocess.run(base + ["--opt_max_iter", "3", "--out_prefix", pre], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [l.split(", ") for l in r.stdout.splitlines() if l[:1].isdigit() or l.startswith("-1, ")]
    e = {int(x[0]): float(x[1]) for x in rows}
    assert set(e) >= {-1, 1, 2, 3}, r.stdout[-2000:]
    assert e[3] < e[1]
    assert os.path.exists(pre + "_state.json") and os.path.exists(pre + "_checkpoint.p")
    r2 = subprocess.run(base + ["--opt_max_iter", "1", "--opt_resume", pre + "_checkpoint.p", "--out_prefix", pre + "b"], capture_output=True,
                        text=True, env=env, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert "resuming from check point" in r2.stdout


@pytest.mark.parametrize("extra", [["--GLOBALARGS_dtype", "complex128"], ["--CTMARGS_projector_svd_method", "SYMEIG", "--OPTARGS_line_search", "strong_wolfe"],
                                   ["--CTMARGS_projector_svd_method", "SYMEIG", "--OPTARGS_line_search", "backtracking"],
                                   ["--CTMARGS_projector_svd_method", "SYMEIG", "--OPTARGS_line_search", "backtracking",
                                    "--OPTARGS_line_search_svd_method", "SYMARP"]],
                         ids=["COMPLEX", "SYMEIG_LS_strong_wolfe", "SYMEIG_LS_backtracking", "SYMEIG_LS_backtracking_SYMARP"])
@pytest.mark.soak          # optim/ is outside SURVEY section 8's scope; the two plain optimiser scripts above stay in the selected set
def test_optim_j1j2_c4v_script_variants(tmp_path, extra):
    """The other cases of the reference's TestOpt (examples/j1j2/optim_j1j2_c4v.py:193-217): complex128 tensors, the two line
    searches, and the forward-only SYMARP method inside the line search."""
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    pre = str(tmp_path / "opt")
    r = subprocess.run([sys.executable, os.path.join(PKG, "examples", "j1j2", "optim_j1j2_c4v.py"), "--bond_dim", "2", "--chi", "16",
                        "--opt_max_iter", "3", "--seed", "123", "--CTMARGS_ctm_max_iter", "20", "--GLOBALARGS_device", "cuda:0",
                        "--out_prefix", pre] + extra, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [l.split(", ") for l in r.stdout.splitlines() if l[:1].isdigit() or l.startswith("-1, ")]
    e = {int(x[0]): float(x[1]) for x in rows}
    assert 1 in e and max(e) == 3, r.stdout[-2000:]
    assert e[3] <= e[1] + 1e-10
    assert os.path.exists(pre + "_state.json")


@pytest.mark.parametrize("extra", [["--tiling", "BIPARTITE"], ["--tiling", "BIPARTITE", "--OPTARGS_line_search", "strong_wolfe"],
                                   ["--tiling", "BIPARTITE", "--OPTARGS_line_search", "backtracking", "--OPTARGS_line_search_svd_method", "ARP"],
                                   ["--tiling", "4SITE"]],
                         ids=["GESDD_BIPARTITE", "GESDD_BIPARTITE_LS_strong_wolfe", "GESDD_BIPARTITE_LS_backtracking", "GESDD_4SITE"])
@pytest.mark.soak
def test_optim_j1j2_script_variants(tmp_path, extra):
    """The reference's TestOptBasic (examples/j1j2/optim_j1j2.py:238-300): j2 = j3 = hz_stag = 1, D = 2, chi = 8, GESDD projectors,
    three epochs -- the j3 term differentiated through the transfer-matrix correlators, the staggered field in the plaquette term."""
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    pre = str(tmp_path / "opt")
    r = subprocess.run([sys.executable, os.path.join(PKG, "examples", "j1j2", "optim_j1j2.py"), "--bond_dim", "2", "--chi", "8", "--j2", "1.",
                        "--j3", "1.", "--hz_stag", "1.", "--delta_zz", "1.", "--opt_max_iter", "3", "--seed", "123",
                        "--CTMARGS_projector_svd_method", "GESDD", "--CTMARGS_ctm_max_iter", "10", "--GLOBALARGS_device", "cuda:0",
                        "--out_prefix", pre] + extra, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [l.split(", ") for l in r.stdout.splitlines() if l[:1].isdigit() or l.startswith("-1, ")]
    e = {int(x[0]): float(x[1]) for x in rows}
    assert 1 in e and max(e) == 3, r.stdout[-2000:]
    assert e[3] <= e[1] + 1e-10
    assert os.path.exists(pre + "_state.json") and os.path.exists(pre + "_checkpoint.p")
