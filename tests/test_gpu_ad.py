"""GPU: the differentiable C4v path on the native kernels -- every node of the graph runs `ctm_einsum`,
`ctm_truncated_eigh` or `ctm_eigh_backward`; gradients against the reference's autograd (tests/golden/c4v_ad_*.npz)."""
import numpy as np
import pytest
import torch
from conftest import golden
from helpers import dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cplx", [False, True])
def test_contraction_node_gradients_on_the_engine(eng, cplx):
    from linalg.native_einsum import einsum
    g = torch.Generator().manual_seed(9)
    dt = torch.complex128 if cplx else torch.float64
    shapes = [(5, 5), (5, 5, 3, 3), (5, 5, 3, 3), (2, 3, 3, 3, 3), (2, 3, 3, 3, 3)]
    expr, conj = "xy,cyuU,xelL,suldr,sULDR->edDcrR", (4,)
    ops = [torch.randn(*s, generator=g, dtype=dt).cuda().requires_grad_(True) for s in shapes]
    ref = [o.detach().clone().requires_grad_(True) for o in ops]
    mine = einsum(expr, *ops, conj=conj)
    tor = torch.einsum(expr, *[(o.conj() if i in conj else o) for i, o in enumerate(ref)])
    assert float((mine - tor).abs().max()) < 1e-12 * float(tor.abs().max())
    w = torch.randn(*mine.shape, generator=g, dtype=dt).cuda()
    (mine * w).sum().abs().backward(); (tor * w).sum().abs().backward()
    for a, b in zip(ops, ref):
        assert float((a.grad - b.grad).abs().max()) < 1e-11 * float(b.grad.abs().max())


def _run(g, checkpoint=False):
    import config as cfg
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V
    from ctm.one_site_c4v import ctmrg_c4v
    from models import j1j2
    A = dev(g["site"]).requires_grad_(True)
    st = IPEPS_C4V(A)
    env = ENV_C4V(g["C0"].shape[0], st)
    env.C[env.keyC] = dev(g["C0"]); env.T[env.keyT] = dev(g["T0"])
    cfg.ctm_args.fwd_checkpoint_move = checkpoint
    try:
        for _ in range(int(g["nmoves"])):
            ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
    finally:
        cfg.ctm_args.fwd_checkpoint_move = False
    e = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=float(g["j2"])).energy_1x1_lowmem(st, env)
    e.backward()
    return float(e.detach()), A.grad.cpu().numpy(), env


@pytest.mark.parametrize("name", ["c4v_ad_D2_chi8", "c4v_ad_D3_chi18", "c4v_ad_D2_chi8_c128"])
def test_c4v_energy_gradient_equals_the_reference_autograd(eng, name):
    g = golden(name)
    e, grad, env = _run(g)
    assert abs(e - float(g["energy"])) < 1e-11
    ref = g["grad"]
    assert float(np.abs(grad - ref).max()) < 1e-9 * max(1.0, float(np.abs(ref).max()))
    _, grad2, _ = _run(g, checkpoint=True)
    assert float(np.abs(grad - grad2).max()) < 1e-12


def test_forward_values_of_the_differentiable_move_equal_the_fused_move(eng):
    """The graph path and the one-call native move are the same mathematics: C' equal, T' equal up to the eigenvector signs."""
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V
    from ctm.one_site_c4v import ctmrg_c4v
    g = golden("c4v_ad_D3_chi18")
    outs = []
    for rg in (True, False):
        A = dev(g["site"]).requires_grad_(rg)
        st = IPEPS_C4V(A)
        env = ENV_C4V(g["C0"].shape[0], st)
        env.C[env.keyC] = dev(g["C0"]); env.T[env.keyT] = dev(g["T0"])
        ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
        outs.append((torch.diagonal(env.get_C()).detach(), env.get_T().detach().abs()))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 1e-11
    assert float((outs[0][1] - outs[1][1]).abs().max()) < 1e-9


def test_gradient_at_a_larger_size_against_finite_differences(eng):
    """D = 3, chi = 30 (n = 270, the iterative forward solvers are not involved: SYMEIG is the full decomposition): directional
    derivative of the energy after two moves along a random C4v-symmetric direction, central differences."""
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env
    from ctm.one_site_c4v import ctmrg_c4v
    from groups.pg import make_c4v_symm
    from models import j1j2
    rng = np.random.default_rng(3)
    a0 = make_c4v_symm(torch.from_numpy(rng.random((2, 3, 3, 3, 3)))).cuda()
    da = make_c4v_symm(torch.from_numpy(rng.random((2, 3, 3, 3, 3)) - 0.5)).cuda()
    st0 = IPEPS_C4V(a0.clone()); env0 = ENV_C4V(30, st0); init_env(st0, env0)
    for _ in range(6):
        ctmrg_c4v.ctm_MOVE_sl(st0.site(), env0)
    C0, T0 = env0.get_C().clone(), env0.get_T().clone()
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.3)

    def energy(a):
        st = IPEPS_C4V(a); env = ENV_C4V(30, st)
        env.C[env.keyC] = C0.clone(); env.T[env.keyT] = T0.clone()
        for _ in range(2):
            ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
        return model.energy_1x1_lowmem(st, env)

    a = a0.clone().requires_grad_(True)
    e = energy(a); e.backward()
    lin = float((a.grad * da).sum())
    h = 1e-5
    with torch.no_grad():
        fd = (float(energy(a0 + h * da)) - float(energy(a0 - h * da))) / (2 * h)
    assert abs(lin - fd) < 1e-6 * max(1.0, abs(fd)), (lin, fd)


# ---- generic 2x2 cell ---------------------------------------------------------------------------------------------------------
def _generic(name, checkpoint=False):
    import config as cfg
    from helpers_cpu import sites_from, env_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    from ctm.generic import ctmrg
    from models import j1j2
    g = golden(name)
    b = golden(str(g["base"]))
    sites = {k: dev(v).requires_grad_(True) for k, v in sites_from(b).items()}
    st = IPEPS(sites, lX=2, lY=2)
    C, T = env_from(b, "warm_")
    env = ENV(next(iter(C.values())).shape[0], st)
    env.C = {k: dev(v) for k, v in C.items()}
    env.T = {k: dev(v) for k, v in T.items()}
    old = cfg.ctm_args.projector_method
    cfg.ctm_args.projector_method = str(g["projector_method"])
    cfg.ctm_args.fwd_checkpoint_move = checkpoint
    try:
        for d in g["moves"]:
            ctmrg.ctm_MOVE(tuple(int(x) for x in d), st, env)
    finally:
        cfg.ctm_args.projector_method = old
        cfg.ctm_args.fwd_checkpoint_move = False
    e = j1j2.J1J2(j1=1.0, j2=float(g["j2"]), j3=float(g["j3"]) if "j3" in g else 0.0).energy_2x2_4site(st, env)
    e.backward()
    return g, float(e.detach()), {k: v.grad.cpu().numpy() for k, v in sites.items()}, env


@pytest.mark.parametrize("name", ["generic_ad_D2_chi8_f64", "generic_ad_D2_chi8_c128", "generic_ad_D2_chi8_f64_4x2",
                                  "generic_ad_D2_chi8_f64_j3", "generic_ad_D2_chi8_c128_j3"])
def test_generic_energy_gradient_equals_the_reference_autograd(eng, name):
    g, e, grads, env = _generic(name)
    assert abs(e - float(g["energy"])) < 1e-11
    for k, gr in grads.items():
        ref = g[f"grad_{k[0]}_{k[1]}"]
        assert float(np.abs(gr - ref).max()) < 1e-9 * max(1.0, float(np.abs(ref).max())), k
    for (c, v), s in env.get_spectra().items():
        assert float(np.abs(s.detach().cpu().numpy() - g[f"spec_{c[0]}_{c[1]}_{v[0]}_{v[1]}"]).max()) < 1e-10


def test_generic_differentiable_move_equals_the_fused_move(eng):
    """Same state and environment with and without requires_grad: corner spectra after one sweep agree (the fused path uses the
    implicit operator and the leading-chi solver, the differentiable one the explicit halves and the full SVD)."""
    import config as cfg
    from helpers_cpu import sites_from, env_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    from ctm.generic import ctmrg
    b = golden("generic_D3_chi18_f64")
    spectra = []
    for rg in (True, False):
        sites = {k: dev(v).requires_grad_(rg) for k, v in sites_from(b).items()}
        st = IPEPS(sites, lX=2, lY=2)
        C, T = env_from(b, "warm_")
        env = ENV(next(iter(C.values())).shape[0], st)
        env.C = {k: dev(v) for k, v in C.items()}; env.T = {k: dev(v) for k, v in T.items()}
        for d in cfg.ctm_args.ctm_move_sequence:
            ctmrg.ctm_MOVE(d, st, env)
        spectra.append({k: v.detach() for k, v in env.get_spectra().items()})
    for k in spectra[0]:
        assert float((spectra[0][k] - spectra[1][k]).abs().max()) < 1e-10


def test_c4v_rdm_graphs_equal_the_fused_rdms(eng):
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V
    from ctm.one_site_c4v import rdm_c4v
    g = golden("c4v_ad_D3_chi18")
    for fn in (rdm_c4v.rdm2x1_sl, rdm_c4v.rdm2x2_NN_lowmem_sl, rdm_c4v.rdm2x2_NNN_lowmem_sl, rdm_c4v.rdm2x2):
        vals = []
        for rg in (True, False):
            st = IPEPS_C4V(dev(g["site"]).requires_grad_(rg))
            env = ENV_C4V(g["C0"].shape[0], st)
            env.C[env.keyC] = dev(g["C0"]); env.T[env.keyT] = dev(g["T0"])
            vals.append(fn(st, env, sym_pos_def=True).detach())
        assert float((vals[0] - vals[1]).abs().max()) < 1e-12, fn.__name__


def test_generic_rdm2x2_graph_equals_the_fused_rdm(eng):
    from helpers_cpu import sites_from, env_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    from ctm.generic import rdm
    b = golden("generic_D2_chi8_c128")
    vals = []
    for rg in (True, False):
        sites = {k: dev(v).requires_grad_(rg) for k, v in sites_from(b).items()}
        st = IPEPS(sites, lX=2, lY=2)
        C, T = env_from(b, "warm_")
        env = ENV(next(iter(C.values())).shape[0], st)
        env.C = {k: dev(v) for k, v in C.items()}; env.T = {k: dev(v) for k, v in T.items()}
        vals.append(rdm.rdm2x2((1, 0), st, env).detach())
    assert float((vals[0] - vals[1]).abs().max()) < 1e-12


def test_generic_small_rdm_graphs_equal_the_fused_rdms(eng):
    from helpers_cpu import sites_from, env_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    from ctm.generic import rdm
    b = golden("generic_D3_chi18_f64")
    for fn in (rdm.rdm1x1, rdm.rdm2x1, rdm.rdm1x2):
        vals = []
        for rg in (True, False):
            sites = {k: dev(v).requires_grad_(rg) for k, v in sites_from(b).items()}
            st = IPEPS(sites, lX=2, lY=2)
            C, T = env_from(b, "warm_")
            env = ENV(next(iter(C.values())).shape[0], st)
            env.C = {k: dev(v) for k, v in C.items()}; env.T = {k: dev(v) for k, v in T.items()}
            vals.append(fn((1, 1), st, env).detach())
        assert float((vals[0] - vals[1]).abs().max()) < 1e-12, fn.__name__


def test_generic_small_rdm_gradients_on_the_engine(eng):
    """Backward of the rdm1x1 / rdm2x1 / rdm1x2 graphs on the kernels (every adjoint network must be contractible left to right
    without an outer product): directional derivative vs central differences."""
    from helpers_cpu import sites_from, env_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    from ctm.generic import rdm
    b = golden("generic_D2_chi8_c128")
    C, T = env_from(b, "warm_")
    chi = next(iter(C.values())).shape[0]
    g = torch.Generator().manual_seed(6)

    def build(sites):
        st = IPEPS(sites, lX=2, lY=2)
        env = ENV(chi, st)
        env.C = {k: dev(v) for k, v in C.items()}; env.T = {k: dev(v) for k, v in T.items()}
        return st, env

    s0 = {k: dev(v) for k, v in sites_from(b).items()}
    ds = {k: (torch.randn(v.shape, generator=g, dtype=torch.float64) * 1e-1).to(v.dtype).cuda() for k, v in s0.items()}
    for fn in (rdm.rdm1x1, rdm.rdm2x1, rdm.rdm1x2):
        sites = {k: v.clone().requires_grad_(True) for k, v in s0.items()}
        st, env = build(sites)
        r = fn((1, 0), st, env)
        O = torch.randn(r.shape, generator=g, dtype=torch.float64).to(r.dtype).to(r.device)

        def val(sites_):
            st_, env_ = build(sites_)
            return torch.real((fn((1, 0), st_, env_) * O).sum())

        val(sites).backward()
        lin = sum(float(torch.real((sites[k].grad.conj() * ds[k]).sum())) for k in sites if sites[k].grad is not None)
        h = 1e-6
        with torch.no_grad():
            fd = (float(val({k: s0[k] + h * ds[k] for k in s0})) - float(val({k: s0[k] - h * ds[k] for k in s0}))) / (2 * h)
        assert abs(lin - fd) < 1e-6 * max(1.0, abs(fd)), (fn.__name__, lin, fd)


@pytest.mark.parametrize("name", ["c4v_optim_D2_chi16", "c4v_optim_D2_chi16_c128"])
def test_c4v_optimizer_follows_the_reference_trajectory_on_the_engine(eng, name, tmp_path):
    """The caller of the differentiable path: optim.ad_optim_lbfgs_mod.optimize_state (L-BFGS, fixed step) over init_env -> 8 CTM
    moves -> energy_1x1_lowmem, all forward and adjoint contractions / eigendecompositions on the native kernels.  Loss of every
    epoch and the final parameters against the reference's own optimisation run (oracle/gen_golden.py c4v_optim_case)."""
    import config as cfg
    from helpers_cpu import run_c4v_optimizer
    g = golden(name)
    old = cfg.global_args.torch_dtype
    cfg.global_args.torch_dtype = torch.complex128 if name.endswith("c128") else torch.float64
    try:
        losses, site, best = run_c4v_optimizer(g, tmp_path, device="cuda")
    finally:
        cfg.global_args.torch_dtype = old
    assert len(losses) == len(g["losses"])
    assert float(np.abs(np.array(losses) - g["losses"]).max()) < 1e-8, (losses, g["losses"])
    assert float(np.abs(site - g["site_final"]).max()) < 1e-6
    assert float(np.abs(best - g["best"]).max()) < 1e-6


@pytest.mark.parametrize("name", ["generic_optim_D2_chi8_f64", "generic_optim_D2_chi8_c128"])
def test_generic_optimizer_follows_the_reference_trajectory_on_the_engine(eng, name, tmp_path):
    """examples/j1j2/optim_j1j2.py on a 2x2 cell: init_env -> 3 CTM iterations (8 directional moves each, 96 full SVDs with the
    regularised backward in one graph) -> plaquette energy, under optim.ad_optim_lbfgs_mod.optimize_state; losses and final
    tensors against the reference's own run (oracle/gen_golden.py generic_optim_case)."""
    import config as cfg
    from helpers_cpu import run_generic_optimizer
    g = golden(name)
    old = cfg.global_args.torch_dtype
    cfg.global_args.torch_dtype = torch.complex128 if name.endswith("c128") else torch.float64
    try:
        losses, sites = run_generic_optimizer(g, tmp_path, device="cuda")
    finally:
        cfg.global_args.torch_dtype = old
    assert len(losses) == len(g["losses"])
    assert float(np.abs(np.array(losses) - g["losses"]).max()) < 1e-8, (losses, g["losses"])
    for c, t in sites.items():
        assert float(np.abs(t - g[f"final_{c[0]}_{c[1]}"]).max()) < 1e-6, c


def test_concurrently_built_graphs_are_reproducible(eng, tmp_path):
    """The site units of a differentiable move are built from worker threads on their own streams and replayed by autograd on those
    streams: three identical optimisation runs in one process must give the same losses to rounding (a cross-stream ordering or
    allocator hazard shows up as run-to-run scatter far above that)."""
    from helpers_cpu import run_generic_optimizer
    g = golden("generic_optim_D2_chi8_f64")
    runs = [run_generic_optimizer(g, tmp_path / f"r{i}", device="cuda")[0] for i in range(3)]
    for r in runs[1:]:
        assert float(np.abs(np.array(r) - np.array(runs[0])).max()) < 1e-12, runs

