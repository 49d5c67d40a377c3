"""CPU: the host side of the differentiable path -- the adjoint networks that linalg/native_einsum.py builds for a
contraction node, and the C4v move / RDM / energy graph assembled from such nodes and the SYMEIG node, checked against gradients
the REFERENCE's autograd produced (tests/golden/c4v_ad_*.npz, written by oracle/gen_golden.py c4v_ad).  The engine is the
oracle-backed double, so only the graph construction is under test here; tests/test_gpu_ad.py runs the same on the native kernels."""
import numpy as np
import pytest
import torch
import backend
from fake_engine import FakeEngine
from conftest import golden


@pytest.fixture()
def fake():
    import config as cfg
    old = cfg.global_args.device
    cfg.global_args.device = 'cpu'
    backend.set_engine(FakeEngine())
    yield cfg
    backend.set_engine(None)
    cfg.global_args.device = old


NETS = [("ab,bc->ac", [(3, 4), (4, 5)], ()),
        ("xy,cyuU,xelL,suldr,sULDR->edDcrR", [(3, 3), (3, 3, 2, 2), (3, 3, 2, 2), (2, 2, 2, 2, 2), (2, 2, 2, 2, 2)], (4,)),
        ("xuUi,xelL,suldr,sULDR,edDj->ijrR", [(3, 2, 2, 3), (3, 3, 2, 2), (2, 2, 2, 2, 2), (2, 2, 2, 2, 2), (3, 2, 2, 3)], (3, 4)),
        ("abst,bcuv->acstuv", [(4, 4, 2, 2), (4, 4, 2, 2)], ())]


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("net", NETS, ids=[n[0] for n in NETS])
def test_adjoint_networks_of_a_contraction_node(fake, net, cplx):
    from linalg.native_einsum import einsum
    expr, shapes, conj = net
    g = torch.Generator().manual_seed(5)
    dt = torch.complex128 if cplx else torch.float64
    ops = [torch.randn(*s, generator=g, dtype=dt).requires_grad_(True) for s in shapes]
    assert torch.autograd.gradcheck(lambda *o: einsum(expr, *o, conj=conj), ops, eps=1e-6, atol=1e-7, rtol=1e-6)
    # the same tensor in two slots (a and conj(a)): autograd adds the slots' gradients
    if conj:
        ref_ops = [o.detach().clone().requires_grad_(True) for o in ops]
        lhs, out = expr.split("->")
        tor = torch.einsum(expr, *[(o.conj() if i in conj else o) for i, o in enumerate(ref_ops)])
        mine = einsum(expr, *ops, conj=conj)
        w = torch.randn(*mine.shape, generator=g, dtype=dt)
        (mine * w).sum().abs().backward(); (tor * w).sum().abs().backward()
        for a, b in zip(ops, ref_ops):
            assert float((a.grad - b.grad).abs().max()) < 1e-11


def test_index_traced_out_alone_is_refused(fake):
    from linalg.native_einsum import einsum
    x = torch.randn(3, 3, 2, 2, dtype=torch.float64, requires_grad=True)
    with pytest.raises((NotImplementedError, Exception)):
        einsum("abii->ab", x).sum().backward()


def _c4v_energy(g, fake_cfg, checkpoint=False):
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V
    from ctm.one_site_c4v import ctmrg_c4v
    from models import j1j2
    A = torch.from_numpy(g["site"].copy()).requires_grad_(True)
    st = IPEPS_C4V(A)
    env = ENV_C4V(g["C0"].shape[0], st)
    env.C[env.keyC] = torch.from_numpy(g["C0"].copy()); env.T[env.keyT] = torch.from_numpy(g["T0"].copy())
    fake_cfg.ctm_args.fwd_checkpoint_move = checkpoint
    try:
        for _ in range(int(g["nmoves"])):
            ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
    finally:
        fake_cfg.ctm_args.fwd_checkpoint_move = False
    e = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=float(g["j2"])).energy_1x1_lowmem(st, env)
    e.backward()
    return float(e), A.grad, env


@pytest.mark.parametrize("name", ["c4v_ad_D2_chi8", "c4v_ad_D3_chi18", "c4v_ad_D2_chi8_c128"])
def test_c4v_energy_gradient_equals_the_reference_autograd(fake, name):
    g = golden(name)
    e, grad, env = _c4v_energy(g, fake)
    assert abs(e - float(g["energy"])) < 1e-11
    assert float(np.abs(np.diag(env.get_C().detach().numpy()) - np.diag(g["C_after"])).max()) < 1e-10
    ref = g["grad"]
    assert float(np.abs(grad.numpy() - ref).max()) < 1e-9 * max(1.0, float(np.abs(ref).max()))


def test_c4v_gradient_with_checkpointed_moves_is_the_same(fake):
    g = golden("c4v_ad_D2_chi8")
    _, g0, _ = _c4v_energy(g, fake)
    _, g1, _ = _c4v_energy(g, fake, checkpoint=True)
    assert float((g0 - g1).abs().max()) < 1e-13


@pytest.mark.parametrize("name", ["c4v_ad_D2_chi8", "c4v_ad_D2_chi8_c128"])
def test_c4v_spectrum_loss_gradient(fake, name):
    """One move, loss = |C'|^2 + |T'|^2: the eigenvalue AND eigenvector branches of SYMEIG.backward with the detached scales."""
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V
    from ctm.one_site_c4v import ctmrg_c4v
    g = golden(name)
    A = torch.from_numpy(g["site"].copy()).requires_grad_(True)
    st = IPEPS_C4V(A)
    env = ENV_C4V(g["C0"].shape[0], st)
    env.C[env.keyC] = torch.from_numpy(g["C0"].copy()); env.T[env.keyT] = torch.from_numpy(g["T0"].copy())
    ctmrg_c4v.ctm_MOVE_sl(st.site(), env)
    l = (torch.diagonal(env.get_C()).abs() ** 2).sum() + (env.get_T().abs() ** 2).sum()
    l.backward()
    assert abs(float(l) - float(g["loss_spec"])) < 1e-10 * float(g["loss_spec"])
    assert float(np.abs(A.grad.numpy() - g["grad_spec"]).max()) < 1e-8 * float(np.abs(g["grad_spec"]).max())


# ---- generic 2x2 cell: one move per direction + energy_2x2_4site, gradients with respect to the four site tensors ---------------
def _generic_energy(name, fake_cfg, checkpoint=False):
    from helpers_cpu import sites_from, env_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    from ctm.generic import ctmrg
    from models import j1j2
    g = golden(name)
    b = golden(str(g["base"]))
    sites = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sites_from(b).items()}
    st = IPEPS(sites, lX=2, lY=2)
    C, T = env_from(b, "warm_")
    env = ENV(next(iter(C.values())).shape[0], st)
    env.C = {k: torch.from_numpy(v.copy()) for k, v in C.items()}
    env.T = {k: torch.from_numpy(v.copy()) for k, v in T.items()}
    old = fake_cfg.ctm_args.projector_method
    fake_cfg.ctm_args.projector_method = str(g["projector_method"])
    fake_cfg.ctm_args.fwd_checkpoint_move = checkpoint
    try:
        for d in g["moves"]:
            ctmrg.ctm_MOVE(tuple(int(x) for x in d), st, env)
    finally:
        fake_cfg.ctm_args.projector_method = old
        fake_cfg.ctm_args.fwd_checkpoint_move = False
    e = j1j2.J1J2(j1=1.0, j2=float(g["j2"]), j3=float(g["j3"]) if "j3" in g else 0.0).energy_2x2_4site(st, env)
    e.backward()
    return g, float(e.detach()), {k: v.grad for k, v in sites.items()}


@pytest.mark.parametrize("name", ["generic_ad_D2_chi8_f64", "generic_ad_D2_chi8_c128", "generic_ad_D2_chi8_f64_4x2",
                                  "generic_ad_D2_chi8_f64_j3", "generic_ad_D2_chi8_c128_j3"])
def test_generic_energy_gradient_equals_the_reference_autograd(fake, name):
    g, e, grads = _generic_energy(name, fake)
    assert abs(e - float(g["energy"])) < 1e-11
    for k, gr in grads.items():
        ref = g[f"grad_{k[0]}_{k[1]}"]
        assert float(np.abs(gr.numpy() - ref).max()) < 1e-9 * max(1.0, float(np.abs(ref).max())), k


def test_generic_gradient_with_checkpointed_moves_is_the_same(fake):
    _, _, g0 = _generic_energy("generic_ad_D2_chi8_f64", fake)
    _, _, g1 = _generic_energy("generic_ad_D2_chi8_f64", fake, checkpoint=True)
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) < 1e-13


@pytest.mark.parametrize("which", ["energy_1x1", "SS2x1", "lowmem_j3"])
def test_c4v_other_rdm_graphs_against_finite_differences(fake, which):
    """rdm2x2 (energy_1x1) and rdm2x1 (nearest-neighbour S.S of eval_obs) of the C4v network as differentiable graphs: directional
    derivative along a random C4v-symmetric direction vs central differences (environment fixed)."""
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V
    from ctm.one_site_c4v import rdm_c4v
    from groups.pg import make_c4v_symm
    from models import j1j2
    g = golden("c4v_ad_D2_chi8")
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.4)
    C0, T0 = torch.from_numpy(g["C0"].copy()), torch.from_numpy(g["T0"].copy())

    def f(a):
        st = IPEPS_C4V(a)
        env = ENV_C4V(C0.shape[0], st)
        env.C[env.keyC] = C0; env.T[env.keyT] = T0
        if which == "energy_1x1":
            return model.energy_1x1(st, env)
        if which == "lowmem_j3":                                    # NN + NNN + the 3x1 pair of the j3 term
            return j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.4, j3=0.25).energy_1x1_lowmem(st, env)
        r2 = rdm_c4v.rdm2x1_sl(st, env, sym_pos_def=True)
        return torch.einsum('ijab,ijab', r2, model.SS_rot.to(r2.dtype))

    a0 = torch.from_numpy(g["site"].copy())
    rng = np.random.default_rng(8)
    da = make_c4v_symm(torch.from_numpy(rng.random(a0.shape) - 0.5))
    a = a0.clone().requires_grad_(True)
    f(a).backward()
    lin = float((a.grad * da).sum())
    h = 1e-6      # the engine double has no fused C4v RDM: the shifted values also go through the graph route
    fd = (float(f((a0 + h * da).requires_grad_(True)).detach()) - float(f((a0 - h * da).requires_grad_(True)).detach())) / (2 * h)
    assert abs(lin - fd) < 1e-7 * max(1.0, abs(fd)), (lin, fd)


@pytest.mark.parametrize("base", ["generic_D2_chi8_f64", "generic_D2_chi8_c128"])
def test_generic_small_rdm_graphs_values_and_finite_differences(fake, base):
    """rdm1x1 / rdm2x1 / rdm1x2 as differentiable graphs: values equal the reference's (golden), directional derivative of
    tr(rho O) with respect to the site tensors vs central differences."""
    from helpers_cpu import sites_from, env_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    from ctm.generic import rdm
    b = golden(base)
    C, T = env_from(b, "warm_")
    chi = next(iter(C.values())).shape[0]
    g = torch.Generator().manual_seed(4)

    def build(sites):
        st = IPEPS(sites, lX=2, lY=2)
        env = ENV(chi, st)
        env.C = {k: torch.from_numpy(v.copy()) for k, v in C.items()}
        env.T = {k: torch.from_numpy(v.copy()) for k, v in T.items()}
        return st, env

    s0 = {k: torch.from_numpy(v.copy()) for k, v in sites_from(b).items()}
    ds = {k: torch.randn(v.shape, generator=g, dtype=torch.float64).to(v.dtype) * 1e-1 for k, v in s0.items()}
    for fn, key in ((rdm.rdm1x1, "rdm1x1"), (rdm.rdm2x1, "rdm2x1"), (rdm.rdm1x2, "rdm1x2")):
        sites = {k: v.clone().requires_grad_(True) for k, v in s0.items()}
        st, env = build(sites)
        r = fn((0, 0), st, env)
        assert float(np.abs(r.detach().numpy() - b[key]).max()) < 1e-10, key
        O = torch.randn(r.shape, generator=g, dtype=torch.float64).to(r.dtype)

        def val(sites_):
            st_, env_ = build(sites_)
            return torch.real((fn((0, 0), st_, env_) * O).sum())

        val(sites).backward()
        lin = sum(float(torch.real((sites[k].grad.conj() * ds[k]).sum())) for k in sites if sites[k].grad is not None)
        h = 1e-6
        fd = (float(val({k: (s0[k] + h * ds[k]).requires_grad_(True) for k in s0}).detach())
              - float(val({k: (s0[k] - h * ds[k]).requires_grad_(True) for k in s0}).detach())) / (2 * h)
        assert abs(lin - fd) < 1e-6 * max(1.0, abs(fd)), (key, lin, fd)


@pytest.mark.parametrize("base", ["c4v_D2_chi8", "c4v_D2_chi8_c128"])
def test_c4v_rdm3x1_host_layer(fake, base):
    """rdm3x1_sl (the j3 pair of the C4v model) is three native contractions in the host layer: values against the reference."""
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V
    from ctm.one_site_c4v import rdm_c4v
    g, j = golden(base), golden("c4v_j3")
    st = IPEPS_C4V(torch.from_numpy(g["site"].copy()))
    env = ENV_C4V(g["warm_C"].shape[0], st)
    env.C[env.keyC] = torch.from_numpy(g["warm_C"].copy()); env.T[env.keyT] = torch.from_numpy(g["warm_T"].copy())
    r = rdm_c4v.rdm3x1_sl(st, env, sym_pos_def=True)
    assert float(np.abs(r.numpy() - j[f"{base}_rdm3x1"]).max()) < 1e-10


@pytest.mark.parametrize("base", ["c4v_D2_chi8", "c4v_D2_chi8_c128"])
def test_c4v_correlators_host_layer(fake, base):
    """rdm1x1 and eval_corrf_SS (plain and canonical) of the C4v model: host graph of native contractions vs the reference's numbers."""
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.one_site_c4v.env_c4v import ENV_C4V
    from ctm.one_site_c4v import rdm_c4v
    from models import j1j2
    g, j = golden(base), golden("c4v_j3")
    st = IPEPS_C4V(torch.from_numpy(g["site"].copy()))
    env = ENV_C4V(g["warm_C"].shape[0], st)
    env.C[env.keyC] = torch.from_numpy(g["warm_C"].copy()); env.T[env.keyT] = torch.from_numpy(g["warm_T"].copy())
    assert float(np.abs(rdm_c4v.rdm1x1(st, env).numpy() - j[f"{base}_rdm1x1"]).max()) < 1e-12
    model = j1j2.J1J2_C4V_BIPARTITE(j1=1.0, j2=0.3)
    for canon in (False, True):
        c = model.eval_corrf_SS(st, env, 4, canonical=canon)
        for k, v in c.items():
            assert float(np.abs(v.numpy() - j[f"{base}_corr{'_canon' if canon else ''}_{k}"]).max()) < 1e-10, (canon, k)
    dd = model.eval_corrf_DD_H(st, env, 3)["dd"]
    assert float(np.abs(dd.numpy() - j[f"{base}_corr_dd"]).max()) < 1e-10
    ddv = model.eval_corrf_DD_V(st, env, 2)["dd"]                           # width-2 channel
    assert float(np.abs(ddv.numpy() - j[f"{base}_corr_dd_v"]).max()) < 1e-10
    from ctm.one_site_c4v import transferops_c4v
    eh, ref = transferops_c4v.get_EH_spec_Ttensor(2, 3, st, env).numpy(), j[f"{base}_eh3"]
    assert float(np.abs(np.hypot(eh[:, 0], eh[:, 1]) - np.hypot(ref[:, 0], ref[:, 1])).max()) < 1e-8


@pytest.mark.parametrize("name", ["c4v_optim_D2_chi16", "c4v_optim_D2_chi16_c128"])
def test_c4v_optimizer_follows_the_reference_trajectory(fake, name, tmp_path):
    """optim.ad_optim_lbfgs_mod.optimize_state driving the differentiable C4v path (init_env, moves, energy: all graphs of native
    nodes): the loss of every L-BFGS epoch and the parameters after the last one equal the reference's own optimisation run
    (oracle/gen_golden.py c4v_optim_case) from the same start tensor."""
    from helpers_cpu import run_c4v_optimizer
    g = golden(name)
    losses, site, best = run_c4v_optimizer(g, tmp_path)
    assert len(losses) == len(g["losses"])
    assert float(np.abs(np.array(losses) - g["losses"]).max()) < 1e-8, (losses, g["losses"])
    assert float(np.abs(site - g["site_final"]).max()) < 1e-6
    assert float(np.abs(best - g["best"]).max()) < 1e-6


@pytest.mark.parametrize("ls", ["strong_wolfe", "backtracking"])
def test_c4v_optimizer_line_searches_lower_the_energy(fake, ls, tmp_path):
    """OPTARGS.line_search = strong_wolfe / backtracking (reference lbfgs_modified.py:312-365): the loss never rises from one epoch
    to the next, the best state on file is the lowest point visited, and a checkpoint resumes."""
    import copy, os
    from helpers_cpu import run_c4v_optimizer
    g = golden("c4v_optim_D2_chi16")
    losses, site, best = run_c4v_optimizer(g, tmp_path, line_search=ls, epochs=3)
    assert len(losses) >= 2 and all(b <= a + 1e-12 for a, b in zip(losses, losses[1:])), losses
    assert losses[-1] < losses[0] - 1e-3
    assert os.path.exists(os.path.join(str(tmp_path), "o_checkpoint.p"))
    from ipeps.ipeps_c4v import IPEPS_C4V
    st = IPEPS_C4V()
    st.load_checkpoint(os.path.join(str(tmp_path), "o_checkpoint.p"))
    assert float((st.site().detach().cpu() - torch.from_numpy(site)).abs().max()) == 0.0 and not st.site().requires_grad


@pytest.mark.parametrize("name", ["generic_optim_D2_chi8_f64", "generic_optim_D2_chi8_c128"])
def test_generic_optimizer_follows_the_reference_trajectory(fake, name, tmp_path):
    """The same on a 2x2 cell (examples/j1j2/optim_j1j2.py): init_env, 3 CTM iterations of 8 directional moves (32 full SVDs each
    with the regularised backward) and the plaquette energy as one graph; L-BFGS losses and final tensors against the reference's
    run (oracle/gen_golden.py generic_optim_case)."""
    from helpers_cpu import run_generic_optimizer
    g = golden(name)
    losses, sites = run_generic_optimizer(g, tmp_path)
    assert len(losses) == len(g["losses"])
    assert float(np.abs(np.array(losses) - g["losses"]).max()) < 1e-8, (losses, g["losses"])
    for c, t in sites.items():
        assert float(np.abs(t - g[f"final_{c[0]}_{c[1]}"]).max()) < 1e-6, c
