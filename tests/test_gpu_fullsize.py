"""GPU: the BASELINE configurations at their FULL sizes (C4v D4 chi64; generic 2x2 D6 chi128, D8 chi256; one unit of
D8 chi384 complex128), checked through size-independent properties -- the oracle cannot run at these sizes in seconds:

 * singular triplets of the truncation: residuals |M v - s u|, |M^T u - s v| <= 1e-12 s0, orthonormal U and V, and
   completeness of the leading chi (norm of the deflated operator M - U S V^T, estimated by power iteration, does not
   exceed the smallest kept value);
 * the fused, implicit-operator projector path against the explicit n x n matrices: same singular values (and the
   same as host LAPACK's svdvals where that finishes in seconds), biorthogonality P^T Pt = 1 on the kept subspace;
 * invariances of a whole sweep: scale invariance under a -> 2a (exact in binary floating point up to the max-abs
   normalisation), every new tensor has max-abs 1, and a 1x1 state tiled on the 2x2 cell gives the same environment
   on all four sites (checks the concurrent scheduling of the site units);
 * C4v: the eigenvalues of the move equal torch.linalg.eigh of the enlarged corner, ordered by magnitude, signs kept;
   the new T is symmetric in its two environment legs.
Tolerances are relative to the largest singular/eigen value; the north-star bound is 1e-10."""
import numpy as np
import pytest
import torch
from helpers import dev

pytestmark = pytest.mark.gpu
UP = (0, -1)


def _state(D, seed, cplx=False, signed=False, tiled=False):
    from ipeps.ipeps import IPEPS
    rng = np.random.default_rng(seed)
    sites = {}
    for y in range(2):
        for x in range(2):
            if tiled and sites:
                sites[(x, y)] = sites[(0, 0)].copy(); continue
            A = rng.random((2, D, D, D, D)) - (0.5 if signed else 0.0)
            if cplx:
                A = A + 1j * (rng.random((2, D, D, D, D)) - (0.5 if signed else 0.0))
            sites[(x, y)] = A / np.abs(A).max()
    return IPEPS({k: dev(v) for k, v in sites.items()})


def _sweep(st, env, n=1):
    import config as cfg
    from ctm.generic import ctmrg
    for _ in range(n):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _r in range(2):
                ctmrg.ctm_MOVE(d, st, env)


def _deflated_norm(M, U, S, V, iters=12, seed=0):
    """Power-iteration estimate of |M - U diag(S) V^H|_2 (a lower bound that converges from below)."""
    g = torch.Generator(device=M.device); g.manual_seed(seed)
    x = torch.randn(M.shape[1], 4, dtype=torch.float64, device=M.device, generator=g).to(M.dtype)
    est = 0.0
    for _ in range(iters):
        x = x / torch.linalg.vector_norm(x, dim=0, keepdim=True)
        y = M @ x - U @ (S.to(M.dtype)[:, None] * (V.conj().T @ x))
        est = float(torch.linalg.vector_norm(y, dim=0).max())
        x = M.conj().T @ y - V @ (S.to(M.dtype)[:, None] * (U.conj().T @ y))
    return est


def _check_unit(eng, st, env, chi, coord=(0, 0), direction=UP, torch_svdvals=False, host_arpack=False):
    from ctm.generic.ctm_components import _halves_t
    t16 = _halves_t(direction, coord, st, env)
    R, Rt = eng.halves(direction, t16)
    M = eng.gemm(R, Rt, transA=True)                                   # M = R^T Rt, plain transpose (ctm_projectors.py:263)
    n = M.shape[0]
    P, Pt, S = eng.projectors_4x4(direction, t16, chi, return_S=True)   # fused path: the n x n matrices are never formed
    U, S2, V = eng.truncated_svd(M, chi)                                # explicit matrix
    s0 = float(S[0])
    assert float((S - S2).abs().max()) < 1e-11 * s0
    if torch_svdvals:
        ref = torch.linalg.svdvals(M.cpu())[:chi].to(S.device)       # LAPACK on the host, the reference's own route (20-50 s at n = 4608)
        kept = (S > 0)
        assert float((S - ref)[kept].abs().max()) < 1e-11 * s0
    if host_arpack:
        # ARPACK on the host (scipy's svds: an independent implementation, the route of the reference's own partial solver
        # linalg/svd_arnoldi.py) with the matrix-vector products on the device through torch -- the explicit M never travels (9.7 GB at
        # n = 24576 complex128, where ~400 host products used to be a minute of the test)
        from scipy.sparse.linalg import svds, LinearOperator
        Mh = M.conj().T.contiguous() if M.is_complex() else M.T.contiguous()
        todev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(M.device).to(M.dtype)
        op = LinearOperator(M.shape, matvec=lambda x: (M @ todev(x)).cpu().numpy(), rmatvec=lambda x: (Mh @ todev(x)).cpu().numpy(),
                            dtype=np.complex128 if M.is_complex() else np.float64)
        ka = int(host_arpack) if host_arpack is not True else 20
        ka = max(2, min(ka, int((S > 1e-9 * s0).sum())))               # (values at the rounding level of M are not a case for a Krylov method)
        ref = np.sort(svds(op, k=ka, which='LM', tol=1e-14, return_singular_vectors=False))[::-1]
        del Mh
        assert np.abs(S[:ka].cpu().numpy() - ref).max() < 1e-11 * s0
    k = int((S2 > 0).sum())
    U, V, Sk = U[:, :k], V[:, :k], S2[:k]
    I = torch.eye(k, dtype=M.dtype, device=M.device)
    assert float((U.conj().T @ U - I).abs().max()) < 1e-12
    assert float((V.conj().T @ V - I).abs().max()) < 1e-12
    assert float((M @ V - U * Sk.to(M.dtype)).abs().max()) < 1e-12 * s0
    assert float((M.conj().T @ U - V * Sk.to(M.dtype)).abs().max()) < 1e-12 * s0
    # completeness: nothing larger than the smallest kept value is left in the deflated operator (up to the multiplet
    # back-off, which may drop values within 1e-8 relative gap of the cut; and to the numerical rank of M)
    rest = _deflated_norm(M, U, Sk, V)
    assert rest <= max(float(Sk[-1]) * (1 + 1e-6), 1e-12 * s0), (rest, float(Sk[-1]))
    # biorthogonality of the fused projectors on the kept subspace: P^T Pt = S^-1/2 U^H M V S^-1/2 = 1, entries weighted
    # by sqrt(s_i s_j)/s0 (the 1/sqrt(s) of the smallest kept triplets amplifies their rounding)
    kk = int((S > 1e-8 * s0).sum())                                     # projector columns below svd_reltol are zero (ctm_projectors.py:266-270)
    G = P[:, :kk].T @ Pt[:, :kk] - torch.eye(kk, dtype=M.dtype, device=M.device)
    w = torch.sqrt(S[:kk] / s0)
    assert float((G.abs() * w[:, None] * w[None, :]).max()) < 1e-11
    return n


@pytest.mark.parametrize("D,chi,signed", [(6, 128, False), (6, 128, True), (8, 256, False), (8, 256, True)],
                         ids=["D6chi128", "D6chi128-signed", "D8chi256", "D8chi256-signed"])
def test_generic_unit_at_full_size(eng, D, chi, signed):
    from ctm.generic.env import ENV, init_env
    st = _state(D, 3, signed=signed)
    env = ENV(chi, st); init_env(st, env)
    lz0 = eng.stat("lz_hits")
    _sweep(st, env, 2 if (D == 6 or signed) else 1)
    if signed:
        # the full-rank regime: the block Krylov solver must have produced the truncations of the sweeps above (n = chi D^2:
        # 4608 and 16384) -- and the unit checked below goes through it again, on the implicit and on the explicit operator
        assert eng.stat("lz_hits") > lz0, "signed state did not reach the block Krylov solver"
        assert int(min((s_ > 1e-8 * s_[0]).sum() for s_ in env.get_spectra().values())) >= chi // 2    # far from the rank-<= 30 of a positive state
    lz1 = eng.stat("lz_hits")
    # host LAPACK singular values of the explicit M (the reference's own route) at n = 4608.  At the full n = 16384 of BASELINE
    # configs[3] a dense LAPACK svdvals is > 8 minutes of host time (measured: the bidiagonalisation streams the 2.1 GB matrix
    # ~16000 times), so the full-rank state is checked there against host ARPACK -- scipy's svds, an independent implementation and
    # the route of the reference's own partial solver (linalg/svd_arnoldi.py) -- on the leading singular values
    # (round 6: host LAPACK on all kept values only for the signed D = 6 state -- the one whose truncation is a block Krylov solve; the
    # positive D = 6 state against ARPACK on every value above the rounding level of M: the dense host decomposition was 20-50 s per case)
    n = _check_unit(eng, st, env, chi, torch_svdvals=(D == 6 and signed), host_arpack=(20 if (D == 8 and signed) else chi if (D == 6 and not signed) else False))
    if signed:
        assert eng.stat("lz_hits") >= lz1 + 2
    assert n == chi * D * D
    env.__dict__.pop("_corner_cache", None)
    eng.trim()


def test_unit_of_the_complex_config_at_full_size(eng):
    """One unit of generic D8 chi384 complex128 (n = 24576): fused projectors only (P^T Pt = 1, plain transpose)."""
    from ctm.generic.env import ENV, init_env
    from ctm.generic.ctm_components import _halves_t
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~100 GB of free HBM")
    chi, D = 384, 8
    st = _state(D, 4, cplx=True)
    env = ENV(chi, st); init_env(st, env)
    t16 = _halves_t(UP, (0, 0), st, env)
    P, Pt, S = eng.projectors_4x4(UP, t16, chi, return_S=True)
    assert P.shape == (chi * D * D, chi) and P.dtype == torch.complex128
    s0 = float(S[0]); kk = int((S > 1e-8 * s0).sum())
    assert kk >= 1 and bool((S[:kk - 1] >= S[1:kk]).all())
    G = P[:, :kk].T @ Pt[:, :kk] - torch.eye(kk, dtype=P.dtype, device=P.device)
    w = torch.sqrt(S[:kk] / s0)
    assert float((G.abs() * w[:, None] * w[None, :]).max()) < 1e-11
    eng.trim()


def test_complex_config_in_its_full_rank_regime(eng):
    """BASELINE configs[4] (generic 2x2, D = 8, chi = 384, complex128, n = 24576) where it is hard: signed complex tensors, sweeps from
    the CTMRG init until every corner has at least chi/2 values above 1e-8 (rank D^2, D^4, ... : a few sweeps), then one more sweep --
    whose 32 truncations must have run the COMPLEX block Krylov solver (reference semantics: ctm_projectors.py:263-283,
    linalg/custom_svd.py:66-95 on the full n x n matrix) -- then the unit property set on that environment: fused implicit operator
    against the explicit n x n matrix, residuals of both relations <= 1e-12 s0, orthonormal factors, deflated norm, biorthogonality,
    host ARPACK on the leading 20 values."""
    from ctm.generic.env import ENV, init_env
    free, _ = torch.cuda.mem_get_info()
    if free < 200e9:
        pytest.skip("needs ~200 GB of free HBM")
    chi, D = 384, 8
    st = _state(D, 4, cplx=True, signed=True)
    env = ENV(chi, st); init_env(st, env)
    rank = lambda: int(min((s_ > 1e-8 * s_[0]).sum() for s_ in env.get_spectra().values()))
    warm = 0
    while rank() < chi // 2:
        assert warm < 6, "ceil(chi / D^2) = 6 warm-up sweeps did not fill the environment"
        _sweep(st, env, 1); warm += 1
    lz0, si0 = eng.stat("lz_hits"), eng.stat("si_fallbacks")
    _sweep(st, env, 1)                                                    # (two sweeps until round 5: 35 s each)
    assert rank() >= chi // 2
    assert eng.stat("lz_hits") >= lz0 + 24, "the truncations of a full-rank complex sweep did not run the block Krylov solver"
    assert eng.stat("si_fallbacks") == si0                                # nothing fell back to the dense decomposition
    for k, t in list(env.C.items()) + list(env.T.items()):
        assert abs(float(t.abs().max()) - 1.0) < 1e-13, k
    env.__dict__.pop("_corner_cache", None)                               # room for the explicit n x n matrices of the check (9.7 GB each)
    eng.trim()
    lz1 = eng.stat("lz_hits")
    n = _check_unit(eng, st, env, chi, host_arpack=True)
    assert n == chi * D * D and eng.stat("lz_hits") >= lz1 + 2            # implicit and explicit operator both went through it
    eng.trim()


def test_sweep_invariances_at_full_size(eng):
    from ctm.generic.env import ENV, init_env
    D, chi = 6, 128
    st = _state(D, 8, tiled=True)
    env = ENV(chi, st); init_env(st, env)
    _sweep(st, env, 2)
    for k, t in list(env.C.items()) + list(env.T.items()):
        assert abs(float(t.abs().max()) - 1.0) < 1e-14, k
    # 1x1 state tiled on 2x2: identical environments on all sites
    for (c, v), t in env.C.items():
        assert float((t.abs() - env.C[((0, 0), v)].abs()).abs().max()) < 1e-9, (c, v)
    spec = env.get_spectra()
    for (c, v), s in spec.items():
        assert float((s - spec[((0, 0), v)]).abs().max()) < 1e-11, (c, v)
    # a -> 2a: the max-abs normalised environment does not change
    from ipeps.ipeps import IPEPS
    st2 = IPEPS({k: 2.0 * v for k, v in st.sites.items()})
    env2 = ENV(chi, st2); init_env(st2, env2)
    _sweep(st2, env2, 2)
    spec2 = env2.get_spectra()
    for k in spec:
        assert float((spec[k] - spec2[k]).abs().max()) < 1e-12, k


def test_c4v_move_at_full_size(eng):
    from groups.pg import make_c4v_symm
    D, chi = 4, 64
    rng = np.random.default_rng(2)
    A = torch.from_numpy(rng.random((2, D, D, D, D))).cuda()
    A = make_c4v_symm(A); A = A / A.abs().max()
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    from ctm.one_site_c4v import ctmrg_c4v
    st = IPEPS_C4V(A)
    env = ENV_C4V(chi, st); init_env(st, env)
    for _ in range(6):
        ctmrg_c4v.ctm_MOVE_sl(A, env)
    C_, T = env.get_C(), env.get_T()
    X = eng.c2x2_c4v(A, C_, T)
    assert float((X - X.T).abs().max()) < 1e-12 * float(X.abs().max())
    nC, nT, Dv = eng.move_c4v(A, C_, T)
    w = torch.linalg.eigvalsh(0.5 * (X + X.T))
    w = w[torch.argsort(w.abs(), descending=True)][:chi]
    kept = Dv != 0
    assert float((Dv - w)[kept].abs().max()) < 1e-12 * float(w.abs().max())
    assert float((nT - nT.permute(1, 0, 2)).abs().max()) < 1e-13
    assert abs(float(nT.abs().max()) - 1.0) < 1e-14


def _energy_and_rdms(st, env, sites=None):
    from ctm.generic import rdm
    from models import j1j2
    model = j1j2.J1J2(j1=1.0, j2=0.5)
    rdms = {c: rdm.rdm2x2(c, st, env) for c in (sites or st.sites)}
    return float(model.energy_per_site(st, env)), rdms


def test_rdm2x2_and_energy_at_D6_chi128(eng):
    """rdm2x2 + J1-J2 energy at BASELINE configs[2] size (n = 4608), through invariants: Hermitian, unit trace, positive up to
    rounding, identical on the four sites of a tiled 1x1 state, unchanged under a -> 2a."""
    from ctm.generic.env import ENV, init_env
    from ipeps.ipeps import IPEPS
    D, chi = 6, 128
    st = _state(D, 21, tiled=True)
    env = ENV(chi, st); init_env(st, env)
    _sweep(st, env, 2)
    e, rdms = _energy_and_rdms(st, env)
    r0 = rdms[(0, 0)].reshape(16, 16)
    for c, r in rdms.items():
        r = r.reshape(16, 16)
        assert abs(float(torch.trace(r)) - 1.0) < 1e-13
        assert float((r - r.T).abs().max()) < 1e-13
        assert float(torch.linalg.eigvalsh(r.cpu()).min()) > -1e-12
        assert float((r - r0).abs().max()) < 1e-10, c                 # tiled state: every plaquette is the same
    st2 = IPEPS({k: 2.0 * v for k, v in st.sites.items()})
    env2 = ENV(chi, st2); init_env(st2, env2)
    _sweep(st2, env2, 2)
    e2, rdms2 = _energy_and_rdms(st2, env2)
    assert abs(e - e2) < 1e-11 * abs(e)
    assert float((rdms2[(0, 0)] - rdms[(0, 0)]).abs().max()) < 1e-11
    assert -2.0 < e < 2.0
    env.__dict__.pop("_corner_cache", None); env2.__dict__.pop("_corner_cache", None)
    eng.trim()


def test_rdm2x2_and_energy_at_D8_chi256(eng):
    """The energy half of the metric at BASELINE configs[3] size (generic 2x2, D = 8, chi = 256, n = 16384; reference models/j1j2.py:236-240,
    ctm/generic/rdm.py:1362-1592): rdm2x2 of all four sites + energy_per_site after two sweeps of the signed (full-rank) state.
    Invariants: Hermitian, unit trace, positive up to rounding, E/site unchanged under a -> 2a (new environment, new sweeps)."""
    from ctm.generic.env import ENV, init_env
    from ipeps.ipeps import IPEPS
    D, chi = 8, 256
    st = _state(D, 23, signed=True)
    env = ENV(chi, st); init_env(st, env)
    _sweep(st, env, 2)
    e, rdms = _energy_and_rdms(st, env, sites=[(0, 0), (1, 1)])          # (the energy evaluates all four plaquettes itself)
    for c, r in rdms.items():
        r = r.reshape(16, 16)
        assert abs(float(torch.trace(r)) - 1.0) < 1e-13, c
        assert float((r - r.T).abs().max()) < 1e-13, c
        assert float(torch.linalg.eigvalsh(r.cpu()).min()) > -1e-11, c
    assert -2.0 < e < 2.0
    env.__dict__.pop("_corner_cache", None); env.__dict__.pop("_warm", None); eng.trim(); torch.cuda.empty_cache()
    st2 = IPEPS({k: 2.0 * v for k, v in st.sites.items()})
    env2 = ENV(chi, st2); init_env(st2, env2)
    _sweep(st2, env2, 2)
    # (ONE plaquette of the rescaled state instead of four + energy: 104 -> ~75 s; the D = 6 twin above compares all of them)
    from ctm.generic import rdm
    r2 = rdm.rdm2x2((0, 0), st2, env2)
    assert float((r2 - rdms[(0, 0)]).abs().max()) < 1e-10
    env2.__dict__.pop("_corner_cache", None); env2.__dict__.pop("_warm", None)
    eng.trim()


def test_rdm2x2_and_energy_at_D4_chi64_against_the_oracle(eng):
    """n = 1024: the native rdm2x2 / energy of every site against the numpy oracle evaluated on the SAME (downloaded) environment
    after two native sweeps of a signed 4-site state."""
    from ctm.generic.env import ENV, init_env
    from oracle import ctm_oracle as O, j1j2_oracle as OJ
    D, chi = 4, 64
    st = _state(D, 22, signed=True)
    env = ENV(chi, st); init_env(st, env)
    _sweep(st, env, 2)
    e, rdms = _energy_and_rdms(st, env)
    ost = O.State({k: v.cpu().numpy() for k, v in st.sites.items()})
    oe = O.Env(chi)
    oe.C = {k: v.cpu().numpy() for k, v in env.C.items()}
    oe.T = {k: v.cpu().numpy() for k, v in env.T.items()}
    ordm = {c: O.rdm2x2(c, ost, oe) for c in ost.sites}
    for c in ordm:
        assert np.abs(rdms[c].cpu().numpy() - ordm[c]).max() < 1e-11, c
    eo = OJ.energy_per_site([ordm[c] for c in ost.sites], 1.0, 0.5)
    assert abs(e - eo) < 1e-10 * abs(eo)
    eng.trim()


def test_whole_move_of_the_complex_config_at_full_size(eng):
    """BASELINE configs[4] (generic D = 8, chi = 384, complex128, n = 24576): ONE whole directional move -- fused projectors of the
    four sites AND the four complex absorbs -- from the CTMRG init.  Checks: every new tensor has max-abs 1, the move is invariant
    under a -> 2a, and the absorb on the non-zero projector prefix equals the absorb with all chi columns (to rounding)."""
    import config as cfg
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from ipeps.ipeps import IPEPS
    free, _ = torch.cuda.mem_get_info()
    if free < 200e9:
        pytest.skip("needs ~190 GB of free HBM")
    chi, D = 384, 8
    st = _state(D, 4, cplx=True)

    def one_move(state, **kw):
        env = ENV(chi, state); init_env(state, env)
        old = {k: getattr(cfg.ctm_args, k, None) for k in kw}
        for k, v in kw.items(): setattr(cfg.ctm_args, k, v)
        try:
            ctmrg.ctm_MOVE(UP, state, env)
        finally:
            for k, v in old.items(): setattr(cfg.ctm_args, k, v)
        out = {("C", k): v for k, v in env.C.items() if k[1] in ((1, -1), (-1, -1))}
        out.update({("T", k): v for k, v in env.T.items() if k[1] == UP})
        env.__dict__.pop("_corner_cache", None); env.__dict__.pop("_warm", None)
        return out
    a = one_move(st)
    assert len(a) == 12
    for k, t in a.items():
        assert t.dtype == torch.complex128 and abs(float(t.abs().max()) - 1.0) < 1e-13, k
        assert t.shape == ((chi, chi) if k[0] == "C" else (chi, D * D, chi)), k
    eng.trim(); torch.cuda.empty_cache()
    b = one_move(st, absorb_skip_zero_columns=False)
    for k in a:
        # same numbers up to the summation order of the differently shaped products (split-K slices follow the shape)
        assert float((a[k] - b[k]).abs().max()) < 1e-13, k
    del b; eng.trim(); torch.cuda.empty_cache()
    c = one_move(IPEPS({k: 2.0 * v for k, v in st.sites.items()}))
    for k in a:
        assert float((a[k] - c[k]).abs().max()) < 1e-12, k
    eng.trim()


def test_full_chunked_plaquette_of_the_complex_config_at_full_size(eng):
    """BASELINE configs[4] (n = 24576, complex128): the WHOLE plaquette RDM of one site through rdm.rdm2x2 -- all 16 lower-half slices,
    looped over in chunks on one GPU because the open halves (241 GB) do not fit at once (rdm._rdm2x2_raw; reference
    ctm/generic/rdm.py:1362-1592).  Hermitian, unit trace, positive up to rounding; and its (cl = 0) block reproduces the single part
    the next test evaluates directly."""
    from ctm.generic.env import ENV, init_env
    from ctm.generic import rdm
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 200e9:
        pytest.skip("needs ~130 GB of free HBM")
    chi, D = 384, 8
    st = _state(D, 4, cplx=True)
    env = ENV(chi, st); init_env(st, env)
    r = rdm.rdm2x2((0, 0), st, env)
    assert r.shape == (2,) * 8 and r.dtype == torch.complex128
    m = r.reshape(16, 16)
    tr = torch.trace(m)
    assert abs(float(tr.real) - 1.0) < 1e-12 and abs(float(tr.imag)) < 1e-12
    assert float((m - m.conj().T).abs().max()) < 1e-12
    assert float(torch.linalg.eigvalsh(m.cpu()).min()) > -1e-11
    eng.trim(); torch.cuda.empty_cache()


@pytest.mark.soak          # 91 s; the chunked whole-plaquette test above runs every part at this size
def test_one_part_of_rdm2x2_of_the_complex_config_at_full_size(eng):
    """BASELINE configs[4] (n = 24576, complex128): the plaquette RDM does not fit one GPU at once (241 GB of open halves); it is
    evaluated in parts (ctm_rdm2x2_part: ranges of lower-half slices, shared by a rank group or looped over on one GPU).  One part
    at full size: the block R[(s0 t0 s1 t1), cl = 0] (lower sites projected on |0><0|) is a Hermitian, positive matrix in
    (s0 s1),(t0 t1), and it equals the corresponding block computed with a two-slice range."""
    from ctm.generic.env import ENV, init_env
    from ctm.generic.ctm_components import _corner_t, LU, RU, RD, LD
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 200e9:
        pytest.skip("needs ~130 GB of free HBM")
    chi, D = 384, 8
    st = _state(D, 4, cplx=True)
    env = ENV(chi, st); init_env(st, env)
    t = _corner_t(LU, (0, 0), st, env) + _corner_t(RU, (1, 0), st, env) + _corner_t(RD, (1, 1), st, env) + _corner_t(LD, (0, 1), st, env)
    r0 = eng.rdm2x2_part(t, 0, 1)                       # (16, 1): [(s0 t0 s1 t1), cl = 0]
    m = r0.reshape(2, 2, 2, 2).permute(0, 2, 1, 3).reshape(4, 4)            # [(s0 s1), (t0 t1)]
    tr = torch.trace(m)
    assert abs(float(tr.imag)) < 1e-12 * abs(float(tr.real)) and float(tr.real) > 0
    m = m / tr
    assert float((m - m.conj().T).abs().max()) < 1e-12
    assert float(torch.linalg.eigvalsh(0.5 * (m + m.conj().T).cpu()).min()) > -1e-12
    eng.trim(); torch.cuda.empty_cache()
    r01 = eng.rdm2x2_part(t, 0, 2)
    assert float((r01[:, :1] - r0).abs().max()) < 1e-12 * float(r0.abs().max())
    eng.trim()
