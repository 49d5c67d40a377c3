"""Pin the oracle against the REAL reference and write golden vectors to tests/golden/.

Runs only in the build container (needs /root/reference + torch CPU).  Usage:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden.py
Every oracle function is compared with the reference function it restates on the same
seeded inputs (assert), then inputs + reference outputs are stored as .npz fixtures.
The fixtures are data only (arrays); no reference source travels.
"""
import os, sys, json
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference')
import numpy as np
import torch
import config as cfg
from ipeps.ipeps import IPEPS, read_ipeps
from ipeps.ipeps_c4v import IPEPS_C4V, read_ipeps_c4v
from ctm.generic.env import ENV, init_env, ctmrg_conv_specC
from ctm.generic import ctmrg, rdm, ctm_components as cc, ctm_projectors as cp
from ctm.one_site_c4v.env_c4v import ENV_C4V
from ctm.one_site_c4v import env_c4v, ctmrg_c4v, rdm_c4v, ctm_components_c4v as cc4
from linalg.custom_svd import truncated_svd_gesdd
from linalg.custom_eig import truncated_eig_sym
from groups.pg import make_c4v_symm
from models import j1j2

from oracle import ctm_oracle as O, c4v_oracle as O4, j1j2_oracle as OJ

GOLD = os.path.join(REPO, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(8)
TOL = 1e-11


def close(a, b, tol=TOL, what=""):
    a = np.asarray(a); b = np.asarray(b)
    err = np.abs(a - b).max() / max(1e-300, np.abs(b).max())
    assert err < tol, f"{what}: rel err {err}"
    return err


def t2n(t):
    return t.detach().cpu().numpy()


def set_dtype(complex_):
    cfg.global_args.dtype = "complex128" if complex_ else "float64"
    cfg.global_args.torch_dtype = torch.complex128 if complex_ else torch.float64


def rand_state(D, complex_, seed, lX=2, lY=2, p=2):
    rng = np.random.default_rng(seed)
    sites = {}
    for y in range(lY):
        for x in range(lX):
            A = rng.random((p, D, D, D, D))
            if complex_:
                A = A + 1j * rng.random((p, D, D, D, D))
            sites[(x, y)] = A / np.abs(A).max()
    return sites


def ref_state(sites):
    ts = {k: torch.from_numpy(v.copy()) for k, v in sites.items()}
    lX = max(k[0] for k in sites) + 1
    lY = max(k[1] for k in sites) + 1
    return IPEPS(ts, lX=lX, lY=lY)


def env_to_np(env):
    C = {k: t2n(v) for k, v in env.C.items()}
    T = {k: t2n(v) for k, v in env.T.items()}
    return C, T


def np_env(C, T, chi):
    e = O.Env(chi)
    e.C = {k: v.copy() for k, v in C.items()}
    e.T = {k: v.copy() for k, v in T.items()}
    return e


def pack_env(prefix, C, T, out):
    for (c, v), t in C.items():
        out[f"{prefix}C_{c[0]}_{c[1]}_{v[0]}_{v[1]}"] = t
    for (c, v), t in T.items():
        out[f"{prefix}T_{c[0]}_{c[1]}_{v[0]}_{v[1]}"] = t


DIRS = {'UP': (0, -1), 'LEFT': (-1, 0), 'DOWN': (0, 1), 'RIGHT': (1, 0)}
REF_C2X2 = {O.LU: (cc.c2x2_LU_t, cc.c2x2_LU_sl_c), O.RU: (cc.c2x2_RU_t, cc.c2x2_RU_sl_c),
            O.RD: (cc.c2x2_RD_t, cc.c2x2_RD_sl_c), O.LD: (cc.c2x2_LD_t, cc.c2x2_LD_sl_c)}
REF_HALVES = {(0, -1): cc.halves_of_4x4_CTM_MOVE_UP, (-1, 0): cc.halves_of_4x4_CTM_MOVE_LEFT,
              (0, 1): cc.halves_of_4x4_CTM_MOVE_DOWN, (1, 0): cc.halves_of_4x4_CTM_MOVE_RIGHT}
REF_ABSORB = {(0, -1): ctmrg.absorb_truncate_CTM_MOVE_UP, (-1, 0): ctmrg.absorb_truncate_CTM_MOVE_LEFT,
              (0, 1): ctmrg.absorb_truncate_CTM_MOVE_DOWN, (1, 0): ctmrg.absorb_truncate_CTM_MOVE_RIGHT}


def generic_case(name, D, chi, complex_, seed, warm_sweeps=2):
    """G6: random 2x2 state; per-function vectors + one move per direction + sweeps."""
    set_dtype(complex_)
    sites = rand_state(D, complex_, seed)
    st = ref_state(sites)
    ost = O.State(sites)
    env = ENV(chi, st)
    init_env(st, env)
    C0, T0 = env_to_np(env)
    oenv = O.init_env_ctmrg(ost, chi)
    for k in C0: close(oenv.C[k], C0[k], what=f"init C{k}")
    for k in T0: close(oenv.T[k], T0[k], what=f"init T{k}")
    out = {}
    for k, v in sites.items():
        out[f"site_{k[0]}_{k[1]}"] = v
    pack_env("init_", C0, T0, out)

    # warm the env up so that all tensors are dense (reference sweeps)
    for _ in range(warm_sweeps):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _r in range(2):
                ctmrg.ctm_MOVE(d, st, env)
        O.ctm_sweep(ost, oenv)
    # spectra parity after the warm-up sweeps (gauge invariant)
    spec_ref = {k: t2n(v) for k, v in env.get_spectra().items()}
    spec_o = O.corner_spectra(oenv)
    for k in spec_ref: close(spec_o[k], spec_ref[k], 1e-9, f"spectra {k}")
    Cw, Tw = env_to_np(env)
    pack_env("warm_", Cw, Tw, out)
    wenv = np_env(Cw, Tw, chi)        # oracle env == reference env (exactly same numbers)

    # corners, closed + open, at every site
    for cid, (ft, fc) in REF_C2X2.items():
        for coord in sites:
            tens = ft(coord, st, env)
            ref = t2n(fc(*tens))
            close(O.c2x2(cid, coord, ost, wenv), ref, what=f"c2x2 {cid} {coord}")
            refo = t2n(fc(*tens, torch.ones(1, dtype=torch.bool)))
            close(O.c2x2(cid, coord, ost, wenv, open_=True), refo, what=f"c2x2 open {cid} {coord}")
            if coord == (0, 0):
                out[f"c2x2_{cid}"] = ref
                if D <= 2:
                    out[f"c2x2open_{cid}"] = refo
    # halves + projectors + absorb + move for each direction (all from the SAME warm env)
    for dn, d in DIRS.items():
        R, Rt = REF_HALVES[d]((0, 0), st, env, mode='sl')
        R, Rt = t2n(R), t2n(Rt)
        oR, oRt = O.halves(d, (0, 0), ost, wenv)
        close(oR, R, what=f"R {dn}"); close(oRt, Rt, what=f"Rt {dn}")
        P, Pt = cp.ctm_get_projectors_from_matrices(torch.from_numpy(R), torch.from_numpy(Rt), chi)
        oP, oPt, oS = O.projectors_from_matrices(R, Rt, chi, return_S=True)
        # P, Pt carry a per-column sign gauge: fix_svd_signs' argmax has exact ties for vectors that are
        # antisymmetric under ket<->bra exchange, and torch/numpy break them differently.  Compare the
        # gauge invariants |P|, |Pt| and P Pt^T.
        close(np.abs(oP), np.abs(t2n(P)), 1e-6, f"|P| {dn}"); close(np.abs(oPt), np.abs(t2n(Pt)), 1e-6, f"|Pt| {dn}")
        close(oP @ oPt.T, t2n(P) @ t2n(Pt).T, 1e-6, f"P Pt^T {dn}")
        out[f"R_{dn}"] = R; out[f"Rt_{dn}"] = Rt
        out[f"P_{dn}"] = t2n(P); out[f"Pt_{dn}"] = t2n(Pt); out[f"S_{dn}"] = oS
        # all-site projectors -> absorb at every site
        Pd, Ptd = {}, {}
        for coord in sites:
            Pd[coord], Ptd[coord] = cp.ctm_get_projectors_4x4(d, coord, st, env)
        Pn = {k: t2n(v) for k, v in Pd.items()}; Ptn = {k: t2n(v) for k, v in Ptd.items()}
        for coord in sites:
            out[f"Pall_{dn}_{coord[0]}_{coord[1]}"] = Pn[coord]
            out[f"Ptall_{dn}_{coord[0]}_{coord[1]}"] = Ptn[coord]
            ref = [t2n(x) for x in REF_ABSORB[d](coord, st, env, Pd, Ptd)]
            mine = O.absorb_truncate(d, coord, ost, wenv, Pn, Ptn)
            for a_, b_, nm in zip(mine, ref, ("nC1", "nC2", "nT")):
                close(a_, b_, what=f"absorb {dn} {coord} {nm}")
                out[f"abs_{dn}_{coord[0]}_{coord[1]}_{nm}"] = b_
        # one full move from the warm env
        e2 = env.clone()
        ctmrg.ctm_MOVE(d, st, e2)
        C2, T2 = env_to_np(e2)
        oe2 = wenv.clone()
        O.ctm_move(d, ost, oe2)
        for k in C2: close(np.abs(oe2.C[k]), np.abs(C2[k]), 1e-7, f"move {dn} |C{k}|")
        for k in T2: close(np.abs(oe2.T[k]), np.abs(T2[k]), 1e-7, f"move {dn} |T{k}|")
        pack_env(f"move_{dn}_", C2, T2, out)

    # RDMs + energy on the warm env
    model = j1j2.J1J2(j1=1.0, j2=0.5)
    rd = []
    for coord in sites:
        r = t2n(rdm.rdm2x2_legacy(coord, st, env))
        close(O.rdm2x2(coord, ost, wenv), r, 1e-10, f"rdm2x2 {coord}")
        rd.append(r)
        out[f"rdm2x2_{coord[0]}_{coord[1]}"] = r
    e_ref = sum(torch.einsum('ijklabcd,ijklabcd', torch.from_numpy(r), model.get_hp(c)) for r, c in zip(rd, sites)) / len(rd)
    e_ref = float(e_ref.real)
    close(OJ.energy_per_site(rd, 1.0, 0.5), e_ref, 1e-12, "energy")
    out["energy_j2_0.5"] = np.array(e_ref)
    r11 = t2n(rdm.rdm1x1((0, 0), st, env, mode='dl')); close(O.rdm1x1((0, 0), ost, wenv), r11, 1e-10, "rdm1x1")
    r21 = t2n(rdm.rdm2x1((0, 0), st, env, mode='dl')); close(O.rdm2x1((0, 0), ost, wenv), r21, 1e-10, "rdm2x1")
    r12 = t2n(rdm.rdm1x2((0, 0), st, env, mode='dl')); close(O.rdm1x2((0, 0), ost, wenv), r12, 1e-10, "rdm1x2")
    out["rdm1x1"] = r11; out["rdm2x1"] = r21; out["rdm1x2"] = r12

    # converged run: spectra + energy (gauge-invariant end-to-end anchor)
    env3 = ENV(chi, st); init_env(st, env3)
    cfg.ctm_args.ctm_max_iter = 60
    env3, hist, *_ = ctmrg.run(st, env3, conv_check=ctmrg_conv_specC)
    nsw = len(hist['conv_crit'])
    C3, T3 = env_to_np(env3)
    e3 = np_env(C3, T3, chi)
    rd3 = [t2n(rdm.rdm2x2_legacy(c, st, env3)) for c in sites]
    out["conv_nsweeps"] = np.array(nsw)
    out["conv_energy"] = np.array(OJ.energy_per_site(rd3, 1.0, 0.5))
    spec3 = {k: t2n(v) for k, v in env3.get_spectra().items()}
    for (c, v), s in spec3.items():
        out[f"conv_spec_{c[0]}_{c[1]}_{v[0]}_{v[1]}"] = s
    # oracle running on its own for the same number of sweeps must land on the same spectra
    oe = O.init_env_ctmrg(ost, chi)
    for _ in range(nsw): O.ctm_sweep(ost, oe)
    so = O.corner_spectra(oe)
    worst = max(np.abs(so[k] - spec3[k]).max() for k in spec3)
    eo = OJ.energy_per_site([O.rdm2x2(c, ost, oe) for c in sites], 1.0, 0.5)
    print(f"  {name}: {nsw} sweeps, oracle-vs-ref spectra {worst:.2e}, energy {eo:.15f} vs {float(out['conv_energy']):.15f}")
    assert worst < 1e-8 and abs(eo - float(out['conv_energy'])) < 1e-10 * abs(eo) + 1e-12
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


def signed_state(D, complex_, seed):
    """A ~ U(-1/2, 1/2) (+ i U(-1/2, 1/2)), A /= max|A|: full-rank environments (every truncation keeps chi significant triplets)."""
    rng = np.random.default_rng(seed)
    sites = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, D, D, D, D)) - 0.5
            if complex_:
                A = A + 1j * (rng.random((2, D, D, D, D)) - 0.5)
            sites[(x, y)] = A / np.abs(A).max()
    return sites


def _run_fixed(st, env, nsweeps):
    """The reference's ctmrg.run for exactly nsweeps sweeps (a conv_check that only counts)."""
    def count(state, env, history, ctm_args=cfg.ctm_args):
        history = history or []
        history.append(len(history))
        return len(history) >= nsweeps, history
    old = cfg.ctm_args.ctm_max_iter
    cfg.ctm_args.ctm_max_iter = nsweeps
    try:
        env, hist, *_ = ctmrg.run(st, env, conv_check=count)
    finally:
        cfg.ctm_args.ctm_max_iter = old
    assert len(hist) == nsweeps
    return env


def fixed_point_case(name, D, chi, complex_, seed, nsweeps):
    """Signed random 2x2 state, the REFERENCE's ctmrg.run for a fixed number of sweeps that ends well inside the stationary regime (the
    corner spectra stop moving at the 1e-11 level after ~16 sweeps): corner spectra, rdm2x2 energy and the per-sweep movement of the spectra.
    n = chi D^2 >= 256 and chi >= 47, so the engine's truncations are block Krylov solves and -- with ctm_args.projector_warm_tol -- its
    stationary fast path; what tests/test_gpu_stationary.py and tests/test_gpu_generic.py pin those routes against."""
    set_dtype(complex_)
    sites = signed_state(D, complex_, seed)
    st = ref_state(sites); ost = O.State(sites)
    env = ENV(chi, st); init_env(st, env)
    env = _run_fixed(st, env, nsweeps)
    spec = {k: t2n(v) for k, v in env.get_spectra().items()}
    rd = [t2n(rdm.rdm2x2_legacy(c, st, env)) for c in sites]
    e = OJ.energy_per_site(rd, 1.0, 0.5)
    oe = O.init_env_ctmrg(ost, chi)
    moved = []
    prev = None
    for _ in range(nsweeps):
        O.ctm_sweep(ost, oe)
        so = O.corner_spectra(oe)
        if prev is not None:
            moved.append(max(np.abs(so[k] - prev[k]).max() for k in so))
        prev = so
    worst = max(np.abs(so[k] - spec[k]).max() for k in spec)
    eo = OJ.energy_per_site([O.rdm2x2(c, ost, oe) for c in sites], 1.0, 0.5)
    print(f"  {name}: {nsweeps} sweeps, oracle-vs-ref spectra {worst:.2e}, energy {eo:.15f} vs {e:.15f}; movement of the spectra in the last sweeps {moved[-3:]}")
    assert worst < 1e-10 and abs(eo - e) < 1e-10 * abs(e) + 1e-13
    out = {f"site_{k[0]}_{k[1]}": v for k, v in sites.items()}
    out.update(chi=np.array(chi), nsweeps=np.array(nsweeps), energy=np.array(e), spectra_movement=np.array(moved))
    for (c, v), s_ in spec.items():
        out[f"spec_{c[0]}_{c[1]}_{v[0]}_{v[1]}"] = s_
    out["rdm2x2_0_0"] = rd[list(sites).index((0, 0))]
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


def chi_ramp_case(name, D, chi0, chi1, complex_, seed, n0, n1):
    """ENV.extend (env.py:164-202) inside a run, as the reference's scripts ramp the environment dimension: n0 sweeps at chi0 from the
    CTMRG init, extend to chi1 (zero padding), n1 more sweeps.  Stored: the extended environment right after extend() (exact), corner
    spectra, |C|, |T| and the rdm2x2 energy at the end; the oracle's env_extend + sweeps are checked against the same run."""
    set_dtype(complex_)
    sites = signed_state(D, complex_, seed)
    st = ref_state(sites); ost = O.State(sites)
    env = ENV(chi0, st); init_env(st, env)
    env = _run_fixed(st, env, n0)
    env = env.extend(chi1)
    Cx, Tx = env_to_np(env)
    env = _run_fixed(st, env, n1)
    spec = {k: t2n(v) for k, v in env.get_spectra().items()}
    rd = [t2n(rdm.rdm2x2_legacy(c, st, env)) for c in sites]
    e = OJ.energy_per_site(rd, 1.0, 0.5)
    oe = O.init_env_ctmrg(ost, chi0)
    for _ in range(n0): O.ctm_sweep(ost, oe)
    oe = O.env_extend(oe, chi1)
    for k in Cx: assert oe.C[k].shape == Cx[k].shape and np.abs(np.abs(oe.C[k]) - np.abs(Cx[k])).max() < 1e-7 * np.abs(Cx[k]).max(), k
    for k in Tx: assert oe.T[k].shape == Tx[k].shape and np.abs(np.abs(oe.T[k]) - np.abs(Tx[k])).max() < 1e-7 * np.abs(Tx[k]).max(), k
    for _ in range(n1): O.ctm_sweep(ost, oe)
    so = O.corner_spectra(oe)
    worst = max(np.abs(so[k] - spec[k]).max() for k in spec)
    eo = OJ.energy_per_site([O.rdm2x2(c, ost, oe) for c in sites], 1.0, 0.5)
    print(f"  {name}: chi {chi0} x {n0} sweeps -> extend({chi1}) -> {n1} sweeps, oracle-vs-ref spectra {worst:.2e}, energy {eo:.15f} vs {e:.15f}")
    assert worst < 1e-10 and abs(eo - e) < 1e-10 * abs(e) + 1e-13
    out = {f"site_{k[0]}_{k[1]}": v for k, v in sites.items()}
    out.update(chi0=np.array(chi0), chi1=np.array(chi1), n0=np.array(n0), n1=np.array(n1), energy=np.array(e))
    for (c, v), s_ in spec.items():
        out[f"spec_{c[0]}_{c[1]}_{v[0]}_{v[1]}"] = s_
    C1, T1 = env_to_np(env)
    pack_env("end_", C1, T1, out)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


def rect_cut_case(name, chi, nsweeps, complex_=False, seed=3):
    """Bond dimensions that differ along one cut: horizontal bonds of dimension 2 in the upper row and 3 in the lower row of a 2x2 cell make
    the halves of an UP / DOWN move rectangular (a = chi 2^2, b = chi 3^2; the reference only asserts R.shape == Rt.shape,
    ctm_projectors.py:209).  The REFERENCE's moves: one per direction from the CTMRG init, then `nsweeps` sweeps -- corner spectra,
    |C|, |T|; the oracle is checked against the same."""
    set_dtype(complex_)
    rng = np.random.default_rng(seed)
    sites = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, 2, 2 + y, 2, 2 + y)) - 0.3                     # a[p,u,l,d,r]: l = r = 2 (row 0), 3 (row 1)
            if complex_:
                A = A + 1j * (rng.random((2, 2, 2 + y, 2, 2 + y)) - 0.3)
            sites[(x, y)] = A / np.abs(A).max()
    st = ref_state(sites); ost = O.State(sites)
    out = {f"site_{k[0]}_{k[1]}": v for k, v in sites.items()}
    out["chi"] = np.array(chi); out["nsweeps"] = np.array(nsweeps)
    for dn, d in DIRS.items():
        env = ENV(chi, st); init_env(st, env)
        ctmrg.ctm_MOVE(d, st, env)
        C1, T1 = env_to_np(env)
        oe = O.init_env_ctmrg(ost, chi)
        O.ctm_move(d, ost, oe)
        for k in C1: close(np.abs(oe.C[k]), np.abs(C1[k]), 1e-8, f"rect move {dn} C{k}")
        for k in T1: close(np.abs(oe.T[k]), np.abs(T1[k]), 1e-8, f"rect move {dn} T{k}")
        pack_env(f"move_{dn}_", C1, T1, out)
    env = ENV(chi, st); init_env(st, env)
    env = _run_fixed(st, env, nsweeps)
    spec = {k: t2n(v) for k, v in env.get_spectra().items()}
    oe = O.init_env_ctmrg(ost, chi)
    for _ in range(nsweeps): O.ctm_sweep(ost, oe)
    so = O.corner_spectra(oe)
    worst = max(np.abs(so[k] - spec[k]).max() for k in spec)
    print(f"  {name}: {nsweeps} sweeps, oracle-vs-ref spectra {worst:.2e}")
    assert worst < 1e-10
    for (c, v), s_ in spec.items():
        out[f"spec_{c[0]}_{c[1]}_{v[0]}_{v[1]}"] = s_
    C1, T1 = env_to_np(env)
    pack_env("end_", C1, T1, out)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


def svd_cases():
    """truncated_svd_gesdd incl. multiplet back-off + fix_svd_signs; truncated_eig_sym."""
    set_dtype(False)
    rng = np.random.default_rng(7)
    out = {}
    n, chi = 24, 8
    # (a) generic decaying spectrum
    Q1, _ = np.linalg.qr(rng.standard_normal((n, n))); Q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.exp(-0.7 * np.arange(n))
    Ma = (Q1 * s) @ Q2.T
    # (b) a degenerate multiplet straddling chi: s[6..9] equal -> back off to index 5
    s2 = s.copy(); s2[6:10] = s2[6]
    Mb = (Q1 * s2) @ Q2.T
    # (c) rank deficient: exact zeros beyond rank 5
    s3 = s.copy(); s3[5:] = 0.
    Mc = (Q1 * s3) @ Q2.T
    for nm, M in (("a", Ma), ("b", Mb), ("c", Mc)):
        U, S, V = truncated_svd_gesdd(torch.from_numpy(M), chi, keep_multiplets=True, eps_multiplet=1e-8, abs_tol=1e-14)
        oU, oS, oV = O.truncated_svd_gesdd(M, chi, keep_multiplets=True, eps_multiplet=1e-8, abs_tol=1e-14)
        close(oS, t2n(S), 1e-12, f"svd {nm} S")
        assert (t2n(S) == 0).sum() == (oS == 0).sum()
        out[f"svd_{nm}_M"] = M; out[f"svd_{nm}_U"] = t2n(U); out[f"svd_{nm}_S"] = t2n(S); out[f"svd_{nm}_V"] = t2n(V)
    close(O.truncated_svd_gesdd(Ma, chi, keep_multiplets=True, eps_multiplet=1e-8)[0], out["svd_a_U"], 1e-9, "svd a U")
    # symmetric eig, mixed signs, + a multiplet case
    lam = np.array([3., -2.5, 2., 1.5, -1.0, 0.7, 0.7, 0.7, 0.3, -0.2] + list(0.1 * np.exp(-np.arange(n - 10))))
    H = (Q1 * lam) @ Q1.T
    H = 0.5 * (H + H.T)
    for nm, ch in (("a", 4), ("b", 6)):
        Dv, U = truncated_eig_sym(torch.from_numpy(H), ch, keep_multiplets=True)
        oD, oU = O.truncated_eig_sym(H, ch, keep_multiplets=True)
        close(oD, t2n(Dv), 1e-12, f"eig {nm}")
        out[f"eig_{nm}_D"] = t2n(Dv); out[f"eig_{nm}_U"] = t2n(U)
    out["eig_H"] = H
    np.savez_compressed(os.path.join(GOLD, "decomp.npz"), **out)
    print("  decomp ok")


def c4v_case(name, D, chi, seed, complex_=False):
    """G7: random C4v-symmetrised site."""
    set_dtype(complex_)
    rng = np.random.default_rng(seed)
    A = rng.random((2, D, D, D, D))
    A = t2n(make_c4v_symm(torch.from_numpy(A)))
    if complex_:      # the complex C4v ansatz of ipeps_c4v.py:60-66: A1-symmetric real part + i * A2-symmetric imaginary part
        B = rng.random((2, D, D, D, D)) - 0.5
        A = A + 1j * t2n(make_c4v_symm(torch.from_numpy(B), irreps=["A2"]))
    A = A / np.abs(A).max()
    st = IPEPS_C4V(torch.from_numpy(A.copy()))
    env = ENV_C4V(chi, st)
    env_c4v.init_env(st, env)
    C0, T0 = t2n(env.get_C()), t2n(env.get_T())
    oC, oT = O4.init_env_ctmrg(A, chi)
    close(np.abs(np.diag(oC)), np.abs(np.diag(C0)), 1e-12, "c4v init C")
    out = dict(site=A, init_C=C0, init_T=T0)
    cfg.ctm_args.ctm_max_iter = 6
    ctmrg_c4v.run(st, env)
    C1, T1 = t2n(env.get_C()), t2n(env.get_T())
    out["warm_C"] = C1; out["warm_T"] = T1
    c22 = t2n(cc4.c2x2_sl(st.site(), env.get_C(), env.get_T()))
    close(O4.c2x2_sl(A, C1, T1), c22, what="c4v c2x2")
    out["c2x2"] = c22
    # one move from the warm env
    def teig(M, ch):
        return truncated_eig_sym(M, ch, keep_multiplets=True)
    e2 = env.clone()
    ctmrg_c4v.ctm_MOVE_sl(st.site(), e2, teig)
    C2, T2 = t2n(e2.get_C()), t2n(e2.get_T())
    oC2, oT2 = O4.ctm_move_sl(A, C1, T1)
    close(np.diag(oC2), np.diag(C2), 1e-10, "c4v move C")
    # T is gauge dependent (signs of eigenvectors): compare |T| and the invariant contraction
    close(np.abs(oT2), np.abs(T2), 1e-8, "c4v move |T|")
    out["move_C"] = C2; out["move_T"] = T2
    # the same move with ctm_absorb_normalization = '2' (_move_normalize_c, ctmrg_c4v.py:182-197)
    e3 = env.clone()
    cfg.ctm_args.ctm_absorb_normalization = '2'
    try:
        ctmrg_c4v.ctm_MOVE_sl(st.site(), e3, teig)
    finally:
        cfg.ctm_args.ctm_absorb_normalization = 'inf'
    C3, T3 = t2n(e3.get_C()), t2n(e3.get_T())
    oC3, oT3 = O4.ctm_move_sl(A, C1, T1, norm_type='2')
    close(np.diag(oC3), np.diag(C3), 1e-10, "c4v move C (2-norm)")
    close(np.abs(oT3), np.abs(T3), 1e-8, "c4v move |T| (2-norm)")
    assert abs(np.linalg.norm(T3.ravel()) - 1.0) < 1e-13
    out["move2_C"] = C3; out["move2_T"] = T3
    for nm, fr, fo in (("rdm2x1", rdm_c4v.rdm2x1_sl, O4.rdm2x1_sl), ("rdmNN", rdm_c4v.rdm2x2_NN_lowmem_sl, O4.rdm2x2_NN_lowmem_sl),
                       ("rdmNNN", rdm_c4v.rdm2x2_NNN_lowmem_sl, O4.rdm2x2_NNN_lowmem_sl), ("rdm2x2", rdm_c4v.rdm2x2, O4.rdm2x2)):
        r = t2n(fr(st, env, sym_pos_def=True))
        close(fo(A, C1, T1, sym_pos_def=True), r, 1e-10, f"c4v {nm}")
        out[nm] = r
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.5)
    e_low = float(torch.real(model.energy_1x1_lowmem(st, env))); e_22 = float(torch.real(model.energy_1x1(st, env)))
    close(OJ.energy_1x1_lowmem(out["rdmNN"], out["rdmNNN"], 1.0, 0.5), e_low, 1e-12, "c4v e lowmem")
    close(OJ.energy_1x1(out["rdm2x2"], 1.0, 0.5), e_22, 1e-12, "c4v e 2x2")
    out["e_lowmem"] = np.array(e_low); out["e_2x2"] = np.array(e_22)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"  {name} ok  E_lowmem={e_low:.12f}")


def c4v_ad_case(name, base, nmoves=2, j2=0.5, complex_=False):
    """Gradient of the energy after `nmoves` C4v moves with respect to the on-site tensor, by the reference's autograd
    (ctmrg_c4v.ctm_MOVE_sl + truncated_eig_sym/SYMEIG.backward + rdm_c4v + J1J2_C4V_BIPARTITE.energy_1x1_lowmem); the
    environment the moves start from (golden `base`: warm_C, warm_T) is a constant."""
    set_dtype(complex_)
    g = np.load(os.path.join(GOLD, base + ".npz"))
    A = torch.from_numpy(g["site"].copy()).requires_grad_(True)
    st = IPEPS_C4V(A)
    chi = g["warm_C"].shape[0]
    env = ENV_C4V(chi, st)
    env.C[env.keyC] = torch.from_numpy(g["warm_C"].copy()); env.T[env.keyT] = torch.from_numpy(g["warm_T"].copy())
    def teig(M, ch):
        return truncated_eig_sym(M, ch, keep_multiplets=True)
    for _ in range(nmoves):
        ctmrg_c4v.ctm_MOVE_sl(st.site(), env, teig)
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=j2)
    e = model.energy_1x1_lowmem(st, env)
    e.backward()
    out = dict(site=g["site"], C0=g["warm_C"], T0=g["warm_T"], energy=np.array(float(torch.real(e))), grad=t2n(A.grad),
               C_after=t2n(env.get_C().detach()), nmoves=np.array(nmoves), j2=np.array(j2))
    # second loss: sum of the squared new corner spectrum after one move (exercises the eigenvalue branch of SYMEIG.backward alone)
    A2 = torch.from_numpy(g["site"].copy()).requires_grad_(True)
    st2 = IPEPS_C4V(A2)
    env2 = ENV_C4V(chi, st2)
    env2.C[env2.keyC] = torch.from_numpy(g["warm_C"].copy()); env2.T[env2.keyT] = torch.from_numpy(g["warm_T"].copy())
    ctmrg_c4v.ctm_MOVE_sl(st2.site(), env2, teig)
    l2 = (torch.diagonal(env2.get_C()).abs() ** 2).sum() + (env2.get_T().abs() ** 2).sum()
    l2.backward()
    out["loss_spec"] = np.array(float(l2)); out["grad_spec"] = t2n(A2.grad)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"  {name} ok  E={float(torch.real(e)):.12f} |grad|={np.linalg.norm(out['grad']):.6e} |grad_spec|={np.linalg.norm(out['grad_spec']):.6e}")


def generic_ad_case(name, base, complex_=False, moves=((0, -1), (-1, 0), (0, 1), (1, 0)), j2=0.5, projector_method='4X4', j3=0.0):
    """Gradient of energy_2x2_4site after one move per direction with respect to the four site tensors, by the reference's
    autograd through ctm_MOVE (halves, truncated_svd_gesdd / SVDGESDD.backward, projectors, absorb) and rdm2x2; the environment the
    moves start from (golden `base`: warm_*) is a constant."""
    from helpers_cpu import sites_from, env_from
    set_dtype(complex_)
    g = np.load(os.path.join(GOLD, base + ".npz"))
    sites = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sites_from(g).items()}
    st = IPEPS(sites, lX=2, lY=2)
    C, T = env_from(g, "warm_")
    chi = next(iter(C.values())).shape[0]
    env = ENV(chi, st)
    env.C = {k: torch.from_numpy(v.copy()) for k, v in C.items()}
    env.T = {k: torch.from_numpy(v.copy()) for k, v in T.items()}
    old = cfg.ctm_args.projector_method
    cfg.ctm_args.projector_method = projector_method
    try:
        for d in moves:
            ctmrg.ctm_MOVE(d, st, env)
    finally:
        cfg.ctm_args.projector_method = old
    model = j1j2.J1J2(j1=1.0, j2=j2, j3=j3)
    # energy_per_site (models/j1j2.py:223-247) with rdm2x2_legacy standing in for rdm2x2 (opt_einsum is not installed here)
    e = 0.0
    for c in st.sites:
        e = e + torch.einsum('ijklabcd,ijklabcd', rdm.rdm2x2_legacy(c, st, env), model.get_hp(c))
        if abs(j3) > 0:
            e = e + j3 * j1j2.eval_nnnn_per_site((0, 0), st, env, model.obs_ops)          # :243-244
    e = torch.real(e) / len(st.sites)
    e.backward()
    out = dict(energy=np.array(float(torch.real(e))), j2=np.array(j2), j3=np.array(j3), base=np.array(base), moves=np.array(moves),
               projector_method=np.array(projector_method))
    for k, v in sites.items():
        out[f"grad_{k[0]}_{k[1]}"] = t2n(v.grad)
    spec = {k: t2n(v) for k, v in env.get_spectra().items()}
    for (c, v), sv in spec.items():
        out[f"spec_{c[0]}_{c[1]}_{v[0]}_{v[1]}"] = sv
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    gn = np.sqrt(sum(np.linalg.norm(out[k]) ** 2 for k in out if k.startswith("grad_")))
    print(f"  {name} ok  E={float(torch.real(e)):.12f} |grad|={gn:.6e}")


def c4v_optim_case(name, base, complex_=False, epochs=4, chi=16, ctm_iter=8, j2=0.2, line_search="default"):
    """Trajectory of the reference's optimiser (examples/j1j2/optim_j1j2_c4v.py: loss_fn = symmetrise + normalise -> init_env ->
    `ctm_iter` CTM moves -> energy_1x1_lowmem; optim/ad_optim_lbfgs_mod.optimize_state, L-BFGS with the default fixed step) from
    the on-site tensor of golden `base`: loss of every epoch and the parameters it ends with."""
    import tempfile, copy
    from ipeps.ipeps_c4v import to_ipeps_c4v
    from optim.ad_optim_lbfgs_mod import optimize_state
    set_dtype(complex_)
    g = np.load(os.path.join(GOLD, base + ".npz"))
    A0 = torch.from_numpy(g["site"].copy())
    A0 = A0 / A0.norm()
    st = IPEPS_C4V(A0.clone())
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=j2)
    ctm_args = copy.deepcopy(cfg.ctm_args); ctm_args.ctm_max_iter = ctm_iter; ctm_args.ctm_conv_tol = -1.0
    opt_args = copy.deepcopy(cfg.opt_args); opt_args.line_search = line_search; opt_args.opt_logging = False
    main_args = copy.deepcopy(cfg.main_args); main_args.opt_max_iter = epochs
    tmp = tempfile.mkdtemp(); main_args.out_prefix = os.path.join(tmp, "o"); main_args.opt_resume = None

    @torch.no_grad()
    def conv_f(state, env, history, ctm_args=ctm_args):
        if not history: history = dict({"log": []})
        r = rdm_c4v.rdm2x1_sl(state, env)
        dist = float('inf')
        if len(history["log"]) > 0: dist = torch.dist(r, history["rdm"], p=2).item()
        history["rdm"] = r; history["log"].append(dist)
        return (dist < ctm_args.ctm_conv_tol or len(history["log"]) >= ctm_args.ctm_max_iter), history

    def loss_fn(state, env, ctx):
        ss = to_ipeps_c4v(state, normalize=True)
        if ctx["opt_args"].opt_ctm_reinit: env_c4v.init_env(ss, env)
        env, *log_ = ctmrg_c4v.run(ss, env, conv_check=conv_f, ctm_args=ctx["ctm_args"])
        return (model.energy_1x1_lowmem(ss, env), env, *log_)
    sites_hist = []
    def obs_fn(state, env, ctx):
        if not ctx["line_search"]: sites_hist.append(t2n(state.site().detach().clone()))
    env = ENV_C4V(chi, to_ipeps_c4v(st))
    env_c4v.init_env(to_ipeps_c4v(st), env)
    hist = {}
    def post(state, env, ctx): hist["loss"] = list(ctx["loss_history"]["loss"])
    optimize_state(st, env, loss_fn, obs_fn=obs_fn, post_proc=post, main_args=main_args, opt_args=opt_args, ctm_args=ctm_args)
    best = read_ipeps_c4v(main_args.out_prefix + "_state.json")
    out = dict(site0=t2n(A0), losses=np.array(hist["loss"]), site_final=t2n(st.site().detach()), sites=np.stack(sites_hist),
               best=t2n(best.site()), chi=np.array(chi), ctm_iter=np.array(ctm_iter), j2=np.array(j2), epochs=np.array(epochs))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"  {name} ok  losses={hist['loss']}")


def generic_optim_case(name, base, complex_=False, epochs=3, chi=8, ctm_iter=3, j2=0.3):
    """Trajectory of the reference's optimiser on a generic 2x2 cell (examples/j1j2/optim_j1j2.py: loss_fn = init_env ->
    `ctm_iter` CTM iterations of 8 directional moves -> energy_2x2_4site; L-BFGS, default fixed step) from the sites of golden
    `base`: loss of every epoch and the parameters it ends with."""
    import tempfile, copy
    import optim.ad_optim_lbfgs_mod as _opt
    from optim.ad_optim_lbfgs_mod import optimize_state
    if not isinstance(_opt.NoFixedPointError, type):          # without yastn the module binds an exception INSTANCE: `except` needs a class
        class _NoFixedPoint(Exception): pass
        _opt.NoFixedPointError = _NoFixedPoint
    set_dtype(complex_)
    g = np.load(os.path.join(GOLD, base + ".npz"))
    sites = {tuple(int(v) for v in k.split('_')[1:]): g[k].copy() for k in g.files if k.startswith('site_')}
    st = ref_state(sites)
    model = j1j2.J1J2(j1=1.0, j2=j2)
    ctm_args = copy.deepcopy(cfg.ctm_args); ctm_args.ctm_max_iter = ctm_iter
    opt_args = copy.deepcopy(cfg.opt_args); opt_args.opt_logging = False
    main_args = copy.deepcopy(cfg.main_args); main_args.opt_max_iter = epochs
    tmp = tempfile.mkdtemp(); main_args.out_prefix = os.path.join(tmp, "o"); main_args.opt_resume = None

    @torch.no_grad()
    def conv_f(state, env, history, ctm_args=ctm_args):
        history = (history or []) + [0.]
        return len(history) >= ctm_args.ctm_max_iter, history

    def loss_fn(state, env, ctx):
        if ctx["opt_args"].opt_ctm_reinit: init_env(state, env)
        env_out, *log_ = ctmrg.run(state, env, conv_check=conv_f, ctm_args=ctx["ctm_args"])
        # energy_per_site (models/j1j2.py:223-247) with rdm2x2_legacy standing in for rdm2x2 (opt_einsum is not installed here)
        e = 0.
        for c in state.sites.keys():
            e = e + torch.einsum('ijklabcd,ijklabcd', rdm.rdm2x2_legacy(c, state, env), model.get_hp(c))
        e = e / len(state.sites)
        return (torch.real(e) if e.is_complex() else e, env, *log_)
    hist = {}
    def post(state, env, ctx): hist["loss"] = list(ctx["loss_history"]["loss"])
    env = ENV(chi, st)
    init_env(st, env)
    optimize_state(st, env, loss_fn, post_proc=post, main_args=main_args, opt_args=opt_args, ctm_args=ctm_args)
    out = dict(losses=np.array(hist["loss"]), chi=np.array(chi), ctm_iter=np.array(ctm_iter), j2=np.array(j2), epochs=np.array(epochs))
    for c, t in sites.items():
        out[f"site0_{c[0]}_{c[1]}"] = t
    for c, t in st.sites.items():
        out[f"final_{c[0]}_{c[1]}"] = t2n(t.detach())
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"  {name} ok  losses={hist['loss']}")


def c4v_j3_case():
    """rdm3x1_sl and the j3 term of energy_1x1_lowmem (rdm_c4v.py:829-994, models/j1j2.py:672-676) on the warm C4v goldens."""
    out = {}
    for base, cplx in (("c4v_D2_chi8", False), ("c4v_D3_chi18", False), ("c4v_D2_chi8_c128", True)):
        set_dtype(cplx)
        g = np.load(os.path.join(GOLD, base + ".npz"))
        A = g["site"]
        st = IPEPS_C4V(torch.from_numpy(A.copy()))
        env = ENV_C4V(g["warm_C"].shape[0], st)
        env.C[env.keyC] = torch.from_numpy(g["warm_C"].copy()); env.T[env.keyT] = torch.from_numpy(g["warm_T"].copy())
        r = t2n(rdm_c4v.rdm3x1_sl(st, env, sym_pos_def=True))
        close(O4.rdm3x1_sl(A, g["warm_C"], g["warm_T"], sym_pos_def=True), r, 1e-10, f"rdm3x1_sl {base}")
        close(t2n(rdm_c4v.rdm3x1(st, env, sym_pos_def=True)), r, 1e-10, f"rdm3x1 (dl) {base}")
        model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.3, j3=0.2)
        e = float(torch.real(model.energy_1x1_lowmem(st, env)))
        out[f"{base}_rdm3x1"] = r; out[f"{base}_e_j3"] = np.array(e)
        out[f"{base}_e1x1_j3"] = np.array(float(torch.real(model.energy_1x1(st, env))))
        vals, labels = model.eval_obs(st, env)
        out[f"{base}_obs"] = np.array([complex(v) for v in vals]); out[f"{base}_obs_labels"] = np.array(",".join(labels))
    # what the reference script prints after FINAL: rho_1x1, spin-spin correlators (plain and canonical), transfer-operator spectrum
    from ctm.one_site_c4v import corrf_c4v as ref_corrf4, transferops_c4v as ref_top4
    for base, cplx in (("c4v_D2_chi8", False), ("c4v_D3_chi18", False), ("c4v_D2_chi8_c128", True)):
        set_dtype(cplx)
        g = np.load(os.path.join(GOLD, base + ".npz"))
        A = g["site"]
        st = IPEPS_C4V(torch.from_numpy(A.copy()))
        env = ENV_C4V(g["warm_C"].shape[0], st)
        env.C[env.keyC] = torch.from_numpy(g["warm_C"].copy()); env.T[env.keyT] = torch.from_numpy(g["warm_T"].copy())
        r11 = t2n(rdm_c4v.rdm1x1(st, env))
        close(O4.rdm1x1(A, g["warm_C"], g["warm_T"]), r11, 1e-12, f"rdm1x1 {base}")
        out[f"{base}_rdm1x1"] = r11
        model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.3)
        for canon in (False, True):
            c = model.eval_corrf_SS(st, env, 4, canonical=canon)
            for k, v in c.items():
                out[f"{base}_corr{'_canon' if canon else ''}_{k}"] = t2n(v)
        sz = t2n(model.obs_ops["sz"]).astype(A.dtype)
        oc = O4.corrf_1sO1sO(A, g["warm_C"], g["warm_T"], sz, lambda r: sz, 3)
        rc = t2n(ref_corrf4.corrf_1sO1sO(st, env, model.obs_ops["sz"].to(st.site().dtype), lambda r: model.obs_ops["sz"].to(st.site().dtype), 3))
        close(oc, rc, 1e-11, f"corrf_1sO1sO {base}")
        out[f"{base}_corr_szsz_plain"] = rc
        out[f"{base}_top"] = t2n(ref_top4.get_Top_spec_c4v(3, st, env))
        out[f"{base}_corr_dd"] = t2n(model.eval_corrf_DD_H(st, env, 3)["dd"])
        out[f"{base}_corr_dd_v"] = t2n(model.eval_corrf_DD_V(st, env, 2)["dd"])
        out[f"{base}_eh3"] = t2n(ref_top4.get_EH_spec_Ttensor(2, 3, st, env))
        if base != "c4v_D3_chi18":
            out[f"{base}_top2"] = t2n(ref_top4.get_Top2_spec_c4v(2, st, env))
    np.savez_compressed(os.path.join(GOLD, "c4v_j3.npz"), **out)
    print("  c4v_j3 ok")


def generic_corr_case():
    """J1J2.eval_corrf_SS / eval_corrf_SpSm (models/j1j2.py:477-527) on the warm generic goldens, both directions."""
    from helpers_cpu import sites_from, env_from
    out = {}
    for base, cplx in (("generic_D2_chi8_f64", False), ("generic_D2_chi8_c128", True)):
        set_dtype(cplx)
        g = np.load(os.path.join(GOLD, base + ".npz"))
        st = ref_state(sites_from(g))
        C, T = env_from(g, "warm_")
        env = ENV(next(iter(C.values())).shape[0], st)
        env.C = {k: torch.from_numpy(v.copy()) for k, v in C.items()}
        env.T = {k: torch.from_numpy(v.copy()) for k, v in T.items()}
        model = j1j2.J1J2(j1=1.0, j2=0.5)
        for d in ((1, 0), (0, 1)):
            for conj_ in (False, True):
                c = model.eval_corrf_SS((0, 0), d, st, env, 2, conjugate=conj_)
                for k, v in c.items():
                    out[f"{base}_ss_{d[0]}{d[1]}_{int(conj_)}_{k}"] = t2n(v)
            c = model.eval_corrf_SpSm((1, 0), d, st, env, 2)
            for k, v in c.items():
                out[f"{base}_spsm_{d[0]}{d[1]}_{k}"] = t2n(v)
    np.savez_compressed(os.path.join(GOLD, "generic_corr.npz"), **out)
    print("  generic_corr ok")


def envinit_case():
    """init_prod / init_from_ipeps_obc (ctm/generic/env.py:274-365, 538-716) on random 2x2 states, chi below and above D^2."""
    out = {}
    for tag, D, chi, cplx, seed in (("f64_D2_chi3", 2, 3, False, 41), ("f64_D2_chi6", 2, 6, False, 42), ("c128_D3_chi7", 3, 7, True, 43)):
        set_dtype(cplx)
        sites = rand_state(D, cplx, seed)
        st = ref_state(sites)
        ost = O.State(sites)
        for k, v in sites.items():
            out[f"{tag}_site_{k[0]}_{k[1]}"] = v
        for kind, fo in (("PROD", O.init_env_prod), ("CTMRG_OBC", O.init_env_obc)):
            env = ENV(chi, st)
            old = cfg.ctm_args.ctm_env_init_type
            cfg.ctm_args.ctm_env_init_type = kind
            try:
                init_env(st, env)
            finally:
                cfg.ctm_args.ctm_env_init_type = old
            C, T = env_to_np(env)
            oe = fo(ost, chi)
            for k in C: close(oe.C[k], C[k], 1e-13, f"{kind} C{k}")
            for k in T: close(oe.T[k], T[k], 1e-13, f"{kind} T{k}")
            pack_env(f"{tag}_{kind}_", C, T, out)
    # one-site C4v variants (ctm/one_site_c4v/env_c4v.py:215-246, 315-355)
    for tag, D, chi, cplx, seed in (("c4v_f64_D2_chi3", 2, 3, False, 51), ("c4v_f64_D3_chi12", 3, 12, False, 52), ("c4v_c128_D2_chi6", 2, 6, True, 53)):
        set_dtype(cplx)
        rng = np.random.default_rng(seed)
        A = t2n(make_c4v_symm(torch.from_numpy(rng.random((2, D, D, D, D)))))
        if cplx:
            A = A + 1j * t2n(make_c4v_symm(torch.from_numpy(rng.random((2, D, D, D, D)) - 0.5), irreps=["A2"]))
        A = A / np.abs(A).max()
        out[f"{tag}_site"] = A
        st4 = IPEPS_C4V(torch.from_numpy(A.copy()))
        for kind, fo in (("PROD", O4.init_env_prod), ("CTMRG_OBC", O4.init_env_obc)):
            e4 = ENV_C4V(chi, st4)
            old = cfg.ctm_args.ctm_env_init_type
            cfg.ctm_args.ctm_env_init_type = kind
            try:
                env_c4v.init_env(st4, e4)
            finally:
                cfg.ctm_args.ctm_env_init_type = old
            C4, T4 = t2n(e4.get_C()), t2n(e4.get_T())
            oC, oT = fo(A, chi)
            close(oC, C4, 1e-13, f"c4v {kind} C")
            if kind == "PROD":       # the leading eigenvector is defined up to a phase
                ph = np.vdot(oT[0, 0, :], T4[0, 0, :]); ph = ph / abs(ph)
                close(oT * ph, T4, 1e-12, f"c4v {kind} T")
            else:
                close(oT, T4, 1e-13, f"c4v {kind} T")
            out[f"{tag}_{kind}_C"] = C4; out[f"{tag}_{kind}_T"] = T4
    np.savez_compressed(os.path.join(GOLD, "envinit.npz"), **out)
    print("  envinit ok")


def rvb_case():
    """G1/G2: the reference's own known-answer test (examples/j1j2/ctmrg_j1j2_c4v.py:218-260):
    RVB_1x1 D=3 chi=16 j2=0.5 -> E = -0.47684229 +- 1e-8."""
    set_dtype(False)
    st = read_ipeps_c4v('/root/reference/test-input/RVB_1x1.in')
    A = t2n(st.site())
    chi = 16
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.5)
    env = ENV_C4V(chi, st); env_c4v.init_env(st, env)
    e_prev = [0.]
    def conv(state, env, history, ctm_args=cfg.ctm_args):
        if not history: history = []
        e = float(model.energy_1x1_lowmem(state, env)); history.append(e)
        done = len(history) > 1 and abs(history[-1] - history[-2]) < 1e-12   # script's ctmrg_conv_energy, ctm_conv_tol=1e-12? -> keep explicit
        return done or len(history) >= 200, history
    cfg.ctm_args.ctm_max_iter = 200
    env, hist, *_ = ctmrg_c4v.run(st, env, conv_check=conv)
    E = hist[-1]
    assert abs(E - (-0.47684229)) < 1e-8, E
    # oracle end-to-end on the same state
    C, T = O4.init_env_ctmrg(A, chi)
    for _ in range(len(hist)):
        C, T = O4.ctm_move_sl(A, C, T)
    Eo = OJ.energy_1x1_lowmem(O4.rdm2x2_NN_lowmem_sl(A, C, T, True), O4.rdm2x2_NNN_lowmem_sl(A, C, T, True), 1.0, 0.5)
    assert abs(Eo - E) < 1e-10, (Eo, E)
    spec = np.abs(np.diag(t2n(env.get_C())))
    np.savez_compressed(os.path.join(GOLD, "rvb_c4v.npz"), site=A, energy=np.array(E), nsweeps=np.array(len(hist)),
                        spec=spec, energy_published=np.array(-0.47684229))
    print(f"  rvb ok: E={E:.14f} after {len(hist)} sweeps; zeros in spec: {(spec == 0).sum()}")


def file_state_case(name, fname, tiling, chi, j1, j2, E_pub, tol_pub, j3=0.0, h_uni=(0., 0., 0.)):
    """G3/G4: the golden-value states of examples/j1j2/ctmrg_j1j2.py:244-266."""
    set_dtype(False)
    if tiling == "BIPARTITE":
        def lattice_to_site(coord):
            vx = (coord[0] + abs(coord[0]) * 2) % 2; vy = abs(coord[1])
            return ((vx + vy) % 2, 0)
    elif tiling == "2SITE":
        def lattice_to_site(coord):
            vx = (coord[0] + abs(coord[0]) * 2) % 2
            return (vx, 0)
    st = read_ipeps(os.path.join('/root/reference/test-input', fname), vertexToSite=lattice_to_site)
    sites = {k: t2n(v) for k, v in st.sites.items()}
    env = ENV(chi, st); init_env(st, env)
    cfg.ctm_args.ctm_max_iter = 40
    env, hist, *_ = ctmrg.run(st, env, conv_check=ctmrg_conv_specC)
    nsw = len(hist['conv_crit'])
    # reference energy: models/j1j2.py:223-247 with rdm2x2_legacy standing in for rdm2x2 (opt_einsum is not installed here)
    from ctm.generic import corrf as ref_corrf
    rmodel = j1j2.J1J2(j1=j1, j2=j2, j3=j3, h_uni=list(h_uni))
    E = 0.0
    for c in st.sites:
        E = E + torch.einsum('ijklabcd,ijklabcd', rdm.rdm2x2_legacy(c, st, env), rmodel.get_hp(c))
        if abs(j3) > 0:
            E = E + j3 * j1j2.eval_nnnn_per_site((0, 0), st, env, rmodel.obs_ops)
    E = float(torch.real(E)) / len(st.sites)
    ost = O.State(sites, lX=st.lX, lY=st.lY, vertexToSite=lattice_to_site)
    oe = O.init_env_ctmrg(ost, chi)
    for _ in range(nsw): O.ctm_sweep(ost, oe)
    Eo = 0.0
    ocorr = lambda cc, d, o1, g2, dist: O.corrf_1sO1sO(cc, d, ost, oe, o1, g2, dist)
    for c in ost.sites:
        Eo += np.einsum('ijklabcd,ijklabcd', O.rdm2x2(c, ost, oe), OJ.get_hp(j1, j2, h_uni=h_uni, coord=c))
        if abs(j3) > 0:
            Eo += j3 * OJ.eval_nnnn_per_site(ocorr, (0, 0))
    Eo = float(np.real(Eo)) / len(ost.sites)
    print(f"  {name}: {nsw} sweeps E_ref={E:.14f} E_oracle={Eo:.14f} published={E_pub}")
    assert abs(E - Eo) < 1e-10
    if E_pub is not None:
        assert abs(E - E_pub) < tol_pub, (E, E_pub)
    out = {f"site_{k[0]}_{k[1]}": v for k, v in sites.items()}
    out.update(energy=np.array(E), nsweeps=np.array(nsw), lX=np.array(st.lX), lY=np.array(st.lY),
               tiling=np.array(tiling), chi=np.array(chi), j1=np.array(j1), j2=np.array(j2), j3=np.array(j3), h_uni=np.array(h_uni))
    spec = {k: t2n(v) for k, v in env.get_spectra().items()}
    for (c, v), s in spec.items():
        out[f"spec_{c[0]}_{c[1]}_{v[0]}_{v[1]}"] = s
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


def aklt_case():
    """G5: AKLT S=2 states of examples/akltS2/ctmrg_akltS2.py:166-221,224-279 (E < 1e-12, magnetisations < 1e-12): parsed
    site tensors + the reference's converged energy (RDMs through mode='dl': opt_einsum is not installed here)."""
    from models import akltS2
    set_dtype(False)
    bip = lambda c: ((((c[0] + abs(c[0]) * 2) % 2) + abs(c[1])) % 2, 0)
    for name, fname, v2s in (("aklt_S2_2x1", "AKLT-S2_2x1_biLat.in", bip), ("aklt_S2_2x2", "AKLT-S2_2x2_ABCD.in", None)):
        st = read_ipeps(os.path.join('/root/reference/test-input', fname), vertexToSite=v2s)
        sites = {k: t2n(v) for k, v in st.sites.items()}
        model = akltS2.AKLTS2()
        env = ENV(32, st); init_env(st, env)
        cfg.ctm_args.ctm_max_iter = 30
        env, hist, *_ = ctmrg.run(st, env, conv_check=ctmrg_conv_specC)
        nsw = len(hist['conv_crit'])
        E = 0.0
        for c in st.sites:
            E = E + torch.einsum('ijab,ijab', rdm.rdm2x1(c, st, env, mode='dl'), model.h) + torch.einsum('ijab,ijab', rdm.rdm1x2(c, st, env, mode='dl'), model.h)
        E = float(torch.real(E)) / len(st.sites)
        ost = O.State(sites, lX=st.lX, lY=st.lY, vertexToSite=v2s)
        oe = O.init_env_ctmrg(ost, 32)
        for _ in range(nsw): O.ctm_sweep(ost, oe)
        h = t2n(model.h)
        Eo = sum(np.einsum('ijab,ijab', O.rdm2x1(c, ost, oe), h) + np.einsum('ijab,ijab', O.rdm1x2(c, ost, oe), h) for c in ost.sites) / len(ost.sites)
        print(f"  {name}: {nsw} sweeps E_ref={E:.3e} E_oracle={float(np.real(Eo)):.3e} (reference test: E < 1e-12)")
        assert abs(E) < 1e-12 and abs(Eo) < 1e-12
        out = {f"site_{k[0]}_{k[1]}": v for k, v in sites.items()}
        out.update(energy=np.array(E), nsweeps=np.array(nsw), lX=np.array(st.lX), lY=np.array(st.lY), h=h)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


def variants_check():
    """Pins the oracle's non-default variants against the real reference (no vectors stored: assertions only):
    projector_method='4X2' (ctm_projectors.py:66-136) and ctm_absorb_normalization='2' (ctmrg.py:210-230), two sweeps
    from the CTMRG init on the committed D=2 chi=8 states, float64 and complex128."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    for name, cx in (("generic_D2_chi8_f64", False), ("generic_D2_chi8_c128", True)):
        g = np.load(os.path.join(GOLD, name + ".npz"))
        sites = {tuple(int(v) for v in k.split('_')[1:]): g[k] for k in g.files if k.startswith('site_')}
        set_dtype(cx)
        for method, norm in (("4X2", "inf"), ("4X4", "2")):
            st = IPEPS({k: torch.from_numpy(v.copy()) for k, v in sites.items()})
            cfg.ctm_args.projector_method, cfg.ctm_args.ctm_absorb_normalization = method, norm
            env = ENV(8, st); init_env(st, env)
            ost = O.State(sites); oe = O.init_env_ctmrg(ost, 8)
            for _ in range(2):
                for d in [(0, -1), (-1, 0), (0, 1), (1, 0)]:
                    for _r in range(2):
                        ctmrg.ctm_MOVE(d, st, env)
                        O.ctm_move(d, ost, oe, norm_type=norm, projector_method=method)
            for k in oe.C: close(np.abs(t2n(env.C[k])), np.abs(oe.C[k]), 1e-8, f"{name} {method} norm={norm} C{k}")
            for k in oe.T: close(np.abs(t2n(env.T[k])), np.abs(oe.T[k]), 1e-8, f"{name} {method} norm={norm} T{k}")
            print("variant ok:", name, method, norm)
    cfg.ctm_args.projector_method, cfg.ctm_args.ctm_absorb_normalization = "4X4", "inf"
    # transfer-matrix correlators (ctm/generic/corrf.py): edges, one transfer step with an operator, <Sz Sz>(r) -- on the
    # reference's environment after two sweeps, all four directions
    from ctm.generic import corrf as ref_corrf
    I2, sz, sp, sm = OJ.su2_ops(2)
    for name, cx in (("generic_D2_chi8_f64", False), ("generic_D2_chi8_c128", True)):
        g = np.load(os.path.join(GOLD, name + ".npz"))
        sites = {tuple(int(v) for v in k.split('_')[1:]): g[k] for k in g.files if k.startswith('site_')}
        set_dtype(cx)
        st = IPEPS({k: torch.from_numpy(v.copy()) for k, v in sites.items()})
        env = ENV(8, st); init_env(st, env)
        for _ in range(2):
            for d in [(0, -1), (-1, 0), (0, 1), (1, 0)]:
                for _r in range(2): ctmrg.ctm_MOVE(d, st, env)
        ost = O.State(sites); oe = O.Env(8)
        oe.C = {k: t2n(v) for k, v in env.C.items()}; oe.T = {k: t2n(v) for k, v in env.T.items()}
        tt = lambda a: torch.from_numpy(a).to(st.dtype)
        for d in [(0, -1), (-1, 0), (0, 1), (1, 0)]:
            for c in [(0, 0), (1, 1)]:
                Eo = O.get_edge(c, d, ost, oe)
                close(t2n(ref_corrf.get_edge(c, d, st, env)), Eo, 1e-13, f"{name} edge {d}")
                close(t2n(ref_corrf.apply_TM_1sO(c, d, st, env, tt(Eo), op=tt(sp))), O.apply_TM_1sO(c, d, ost, oe, Eo, op=sp.astype(Eo.dtype)), 1e-13, f"{name} TM {d}")
                close(t2n(ref_corrf.corrf_1sO1sO(c, d, st, env, tt(sz), lambda r: tt(sz), 3)),
                      O.corrf_1sO1sO(c, d, ost, oe, sz.astype(Eo.dtype), lambda r: sz.astype(Eo.dtype), 3), 1e-12, f"{name} corrf {d}")
        # leading eigenvalues of the width-0 transfer operator (ctm/generic/transferops.py:119-207)
        from ctm.generic import transferops as ref_top
        for d in [(1, 0), (0, 1)]:
            L = t2n(ref_top.get_Top_spec(4, (0, 0), d, st, env))
            close(np.abs(L[:, 0] + 1j * L[:, 1]), np.abs(O.get_Top_spec(4, (0, 0), d, ost, oe)), 1e-10, f"{name} Top spec {d}")
        print("corrf + transfer-operator spectrum ok:", name)


def backward_case():
    """f4 (first part): the reference's regularised SVD / eig backward (svd_gesdd.py:209-328, eig_sym.py:57-75) on seeded factors
    and gradients, real and complex, square and thin -- pins oracle.svd_backward / eigh_backward; fixtures for the native kernels."""
    from linalg.svd_gesdd import SVDGESDD
    from linalg.eig_sym import SYMEIG
    out = {}
    rng = np.random.default_rng(77)

    class Ctx:
        diagnostics = None
    for tag, cx, m, n, k in (("sq_f64", False, 24, 24, 24), ("thin_f64", False, 30, 26, 9), ("sq_c128", True, 20, 20, 20), ("thin_c128", True, 28, 22, 7)):
        A = rng.standard_normal((m, n)) + (1j * rng.standard_normal((m, n)) if cx else 0.0)
        U, S, Vh = np.linalg.svd(A, full_matrices=False)
        U, S, V = U[:, :k], S[:k], Vh.conj().T[:, :k]
        g = lambda r, c: rng.standard_normal((r, c)) + (1j * rng.standard_normal((r, c)) if cx else 0.0)
        gU, gS, gV = g(m, k), rng.standard_normal(k), g(n, k)
        eps = 1e-12
        ctx = Ctx(); ctx.saved_tensors = (torch.from_numpy(U), torch.from_numpy(S), torch.from_numpy(V), torch.tensor(eps, dtype=torch.float64))
        dA = t2n(SVDGESDD.backward(ctx, torch.from_numpy(gU), torch.from_numpy(gS), torch.from_numpy(gV))[0])
        close(O.svd_backward(U, S, V, gU, gS, gV, eps), dA, 1e-12, f"svd backward {tag}")
        for nm, v in (("U", U), ("S", S), ("V", V), ("gU", gU), ("gS", gS), ("gV", gV), ("dA", dA)):
            out[f"svd_{tag}_{nm}"] = v
    for tag, cx, n in (("f64", False, 18), ("c128", True, 16)):
        H = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cx else 0.0)
        H = 0.5 * (H + H.conj().T)
        D, U = np.linalg.eigh(H)
        p = np.argsort(-np.abs(D), kind='stable'); D, U = D[p], U[:, p]
        gD = rng.standard_normal(n)
        gU = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cx else 0.0)
        ctx = Ctx(); ctx.saved_tensors = (torch.from_numpy(D), torch.from_numpy(U), torch.tensor(1e-12, dtype=torch.float64))
        dA = t2n(SYMEIG.backward(ctx, torch.from_numpy(gD).to(torch.from_numpy(U).dtype) if False else torch.from_numpy(gD), torch.from_numpy(gU))[0])
        close(O.eigh_backward(D, U, gD, gU, 1e-12), dA, 1e-12, f"eigh backward {tag}")
        for nm, v in (("D", D), ("U", U), ("gD", gD), ("gU", gU), ("dA", dA)):
            out[f"eig_{tag}_{nm}"] = v
    np.savez_compressed(os.path.join(GOLD, "backward.npz"), **out)
    print("  backward ok: svd (square, thin; f64, c128) and eigh (f64, c128) adjoints pinned against the reference")


def input_files_case():
    """a18: the reference's own test-input DATA files (legacy "entries" format with aux_seq, "1D" format, multi-site cells with a
    pattern) copied as fixtures, next to the arrays the REFERENCE's read_ipeps / read_ipeps_c4v parse from them."""
    import shutil
    set_dtype(False)
    src = '/root/reference/test-input'
    dst = os.path.join(GOLD, 'test-input')
    os.makedirs(dst, exist_ok=True)
    files = ["RVB_1x1.in", "RVB_2x1_AB.in", "RVB_2x2_ABCD.in", "VBS_1x2_AB_D2.in", "AKLT-S2_2x1_biLat.in", "AKLT-S2_2x2_ABCD.in",
             "gesdd-D2-chi50-j20.55-run0-iRND2x1_state.json", "BIPARTITE_j2_0_j3_1250_h_39000_D_3_chi_32_seed_100_state.json"]
    out = {}
    for f in files:
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)
        st = read_ipeps(os.path.join(src, f))
        tag = f.replace('.', '_').replace('-', '_')
        out[f"{tag}__lXlY"] = np.array([st.lX, st.lY])
        for c, t in st.sites.items():
            out[f"{tag}__site_{c[0]}_{c[1]}"] = t2n(t)
        # the tiling the file's own pattern (or the default PBC) defines, probed on a window around the origin
        win = [(x, y) for y in range(-3, 4) for x in range(-3, 4)]
        out[f"{tag}__v2s"] = np.array([st.vertexToSite(v) for v in win])
        for asq in ([0, 1, 2, 3], [3, 0, 1, 2]):
            st2 = read_ipeps(os.path.join(src, f), aux_seq=asq)
            out[f"{tag}__aux{''.join(map(str, asq))}"] = t2n(next(iter(st2.sites.values())))
    s4 = read_ipeps_c4v(os.path.join(src, "RVB_1x1.in"))
    out["RVB_1x1_in__c4v_site"] = t2n(s4.site())
    np.savez_compressed(os.path.join(GOLD, "test_input_parsed.npz"), **out)
    print("  input files: copied", len(files), "reference data files and the arrays the reference parses from them")


def chiral_case():
    """energy_per_site with the chiral plaquette term (reference models/j1j2.py:236-247: the UN-rotated chiral_term added per plaquette)
    on the stored complex128 state: the reference's own model tensors contracted with the reference's rdm2x2 of every site."""
    set_dtype(True)
    g = np.load(os.path.join(GOLD, "generic_D2_chi8_c128.npz"))
    coords = [(0, 0), (1, 0), (0, 1), (1, 1)]
    model = j1j2.J1J2(j1=1.0, j2=0.5, lmbd=0.3)
    e = 0.
    for c in coords:
        r = torch.from_numpy(g[f"rdm2x2_{c[0]}_{c[1]}"])
        e = e + torch.einsum('ijklabcd,ijklabcd', r, model.get_hp(c)) + model.lmbd * torch.einsum('ijklabcd,ijklabcd', r, model.chiral_term)
    e = (e / len(coords))
    assert abs(e.imag) < 1e-12
    np.savez_compressed(os.path.join(GOLD, "chiral.npz"), energy_j2_05_lmbd_03=np.array(float(e.real)), lmbd=np.array(0.3),
                        chiral_term=t2n(model.chiral_term))
    print(f"  chiral: energy_per_site(j2=0.5, lmbd=0.3) on generic_D2_chi8_c128 = {float(e.real):.15f}")
    set_dtype(False)


if __name__ == "__main__":
    which = sys.argv[1:] or ["decomp", "generic", "c4v", "c4v_ad", "generic_ad", "c4v_j3", "generic_corr", "envinit", "rvb", "files", "variants", "aklt", "inputs", "backward", "fixed_point", "chi_ramp", "rect_cut"]
    if "chiral" in which:
        chiral_case()
    if "backward" in which:
        backward_case()
    if "inputs" in which:
        input_files_case()
    if "aklt" in which:
        aklt_case()
    if "variants" in which:
        variants_check()
    if "decomp" in which:
        svd_cases()
    if "rect_cut" in which:
        rect_cut_case("rect_cut_chi5_f64", 5, 3)
        rect_cut_case("rect_cut_chi5_c128", 5, 2, complex_=True, seed=4)
    if "fixed_point" in which:
        fixed_point_case("fixed_point_D3_chi48_f64", 3, 48, False, 11, 26)
        fixed_point_case("fixed_point_D3_chi48_c128", 3, 48, True, 12, 26)
    if "chi_ramp" in which:
        chi_ramp_case("chi_ramp_D3_chi16_36_f64", 3, 16, 36, False, 14, 3, 4)
        chi_ramp_case("chi_ramp_D2_chi6_12_c128", 2, 6, 12, True, 15, 3, 4)
    if "generic" in which:
        generic_case("generic_D2_chi8_f64", 2, 8, False, 11)
        generic_case("generic_D3_chi18_f64", 3, 18, False, 12)
        generic_case("generic_D2_chi8_c128", 2, 8, True, 13)
    if "c4v" in which:
        c4v_case("c4v_D2_chi8", 2, 8, 21)
        c4v_case("c4v_D3_chi18", 3, 18, 22)
        c4v_case("c4v_D2_chi8_c128", 2, 8, 23, complex_=True)
        c4v_case("c4v_D3_chi18_c128", 3, 18, 24, complex_=True)
    if "c4v_ad" in which:
        c4v_ad_case("c4v_ad_D2_chi8", "c4v_D2_chi8")
        c4v_ad_case("c4v_ad_D3_chi18", "c4v_D3_chi18")
        c4v_ad_case("c4v_ad_D2_chi8_c128", "c4v_D2_chi8_c128", complex_=True)
    if "generic_ad" in which:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        generic_ad_case("generic_ad_D2_chi8_f64", "generic_D2_chi8_f64")
        generic_ad_case("generic_ad_D2_chi8_c128", "generic_D2_chi8_c128", complex_=True)
        generic_ad_case("generic_ad_D2_chi8_f64_4x2", "generic_D2_chi8_f64", moves=((0, -1), (1, 0)), projector_method='4X2')
        generic_ad_case("generic_ad_D2_chi8_f64_j3", "generic_D2_chi8_f64", moves=((0, -1), (-1, 0)), j3=0.3)
        generic_ad_case("generic_ad_D2_chi8_c128_j3", "generic_D2_chi8_c128", complex_=True, moves=((0, 1),), j3=0.3)
    if "c4v_optim" in which:
        c4v_optim_case("c4v_optim_D2_chi16", "c4v_D2_chi8")
        c4v_optim_case("c4v_optim_D2_chi16_c128", "c4v_D2_chi8_c128", complex_=True)
    if "generic_optim" in which:
        generic_optim_case("generic_optim_D2_chi8_f64", "generic_D2_chi8_f64")
        generic_optim_case("generic_optim_D2_chi8_c128", "generic_D2_chi8_c128", complex_=True)
    if "c4v_j3" in which:
        c4v_j3_case()
    if "generic_corr" in which:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        generic_corr_case()
    if "envinit" in which:
        envinit_case()
    if "rvb" in which:
        rvb_case()
    if "files" in which:
        file_state_case("twosite_D2_chi32", "gesdd-D2-chi50-j20.55-run0-iRND2x1_state.json", "2SITE", 32, 1.0, 0.55,
                        -0.4434603770143078, 1e-6)
        file_state_case("bipartite_D3_chi32", "BIPARTITE_j2_0_j3_1250_h_39000_D_3_chi_32_seed_100_state.json", "BIPARTITE", 32, 1.0, 0.0,
                        -1.3896897615463615, 1e-6, j3=0.125, h_uni=(3.9, 0., 0.))
    print("golden vectors written to", GOLD)
