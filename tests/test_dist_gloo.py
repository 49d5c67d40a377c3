"""CPU, world_size 2, gloo: the site-sharded directional move (parallel.py) gives every rank the same
environment as the single-process move.  Compute runs on the oracle-backed engine double."""
import os, sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, out_dir, name):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "peps-torch_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import config as cfg
    cfg.global_args.device = 'cpu'
    import backend
    from fake_engine import FakeEngine
    attached = []
    if os.environ.get("CTM_TEST_SHARED_UNITS"):
        # the orchestration of a unit shared by a rank pair (twice as many ranks as sites; DESIGN.md section 6 "Round 6") with an engine double
        # whose set_group() only proves that both members attach the same process group at the same point (the split passes themselves
        # are native: tests/test_gpu_dist.py)
        class SharingEngine(FakeEngine):
            def set_group(self, members, capacity_doubles=0):
                if members:
                    import parallel as par
                    got = [torch.zeros(1, dtype=torch.int64) for _ in members]
                    dist.all_gather(got, torch.tensor([rank], dtype=torch.int64), group=par._process_group(list(members)))
                    attached.append(sorted(int(x) for x in got))
        backend.set_engine(SharingEngine())
    else:
        backend.set_engine(FakeEngine())
    from conftest import golden
    from helpers_cpu import sites_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from models import j1j2
    import parallel
    g = golden(name)
    st = IPEPS({k: torch.from_numpy(v.copy()) for k, v in sites_from(g).items()})
    env = ENV(8, st)
    init_env(st, env)
    assert parallel.is_distributed() and len(parallel.my_units(list(st.sites))) == len([i for i in range(4) if i % world == rank])
    for _ in range(2):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _r in range(2):
                ctmrg.ctm_MOVE(d, st, env)
    if os.environ.get("CTM_TEST_SHARED_UNITS"):
        shared_ok = world == 8 and not next(iter(st.sites.values())).is_complex()
        assert len(attached) == (16 if shared_ok else 0), (rank, len(attached))          # every move of a float64 run with 2 x Nsites ranks
        assert all(a == [rank % 4, rank % 4 + 4] for a in attached), attached
    e = float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))
    key = lambda k: f"{k[0][0]}_{k[0][1]}_{k[1][0]}_{k[1][1]}"
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), energy=e, **{"C" + key(k): t.numpy() for k, t in env.C.items()},
             **{"T" + key(k): t.numpy() for k, t in env.T.items()})
    dist.barrier()
    dist.destroy_process_group()


# world 2: two sites per rank (one stacked all-gather per phase); world 3: uneven ownership (per-site broadcasts);
# world 5: more ranks than sites (rank 4 owns nothing -- the 8-GPU case of a 4-site cell)
# world 8: two ranks per site -- in the moves ranks 4-7 own nothing, the plaquette RDM of each site is split over its rank pair
# (lower-half slices, one all-reduce inside the pair; world 5: only site 0 has a pair)
@pytest.mark.parametrize("name,world", [("generic_D2_chi8_f64", 2), ("generic_D2_chi8_c128", 2), ("generic_D2_chi8_f64", 3),
                                        ("generic_D2_chi8_c128", 5), ("generic_D2_chi8_f64", 8)])
def test_sharded_move_equals_single_process(tmp_path, name, world):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), name), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    for r in range(1, world):
        r1 = np.load(tmp_path / f"rank{r}.npz")
        for k in r0.files:
            assert np.array_equal(r0[k], r1[k]), (r, k)          # replicated env identical on all ranks
    # single-process oracle reference
    from conftest import golden
    from helpers_cpu import sites_from
    from oracle import ctm_oracle as O, j1j2_oracle as OJ
    g = golden(name)
    ost = O.State(sites_from(g))
    oe = O.init_env_ctmrg(ost, 8)
    for _ in range(2):
        O.ctm_sweep(ost, oe)
    key = lambda k: f"{k[0][0]}_{k[0][1]}_{k[1][0]}_{k[1][1]}"
    for k, t in oe.C.items():
        assert np.abs(r0["C" + key(k)] - t).max() < 1e-12
    for k, t in oe.T.items():
        assert np.abs(r0["T" + key(k)] - t).max() < 1e-12
    e = OJ.energy_per_site([O.rdm2x2(c, ost, oe) for c in ost.sites], 1.0, 0.5)
    assert abs(float(r0["energy"]) - e) < 1e-12


def _ad_worker(rank, world, port, out_dir, name):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "peps-torch_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import config as cfg
    cfg.global_args.device = 'cpu'
    import backend
    from fake_engine import FakeEngine
    backend.set_engine(FakeEngine())
    from conftest import golden
    from helpers_cpu import sites_from, env_from
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV
    from ctm.generic import ctmrg
    from models import j1j2
    import parallel
    g = golden(name)
    b = golden(str(g["base"]))
    if b["site_0_0"].dtype.kind == "c":
        cfg.global_args.torch_dtype = torch.complex128
    sites = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sites_from(b).items()}
    st = IPEPS(sites, lX=2, lY=2)
    C, T = env_from(b, "warm_")
    env = ENV(next(iter(C.values())).shape[0], st)
    env.C = {k: torch.from_numpy(v.copy()) for k, v in C.items()}
    env.T = {k: torch.from_numpy(v.copy()) for k, v in T.items()}
    for d in g["moves"]:
        ctmrg.ctm_MOVE(tuple(int(x) for x in d), st, env)
    e = j1j2.J1J2(j1=1.0, j2=float(g["j2"]), j3=float(g["j3"])).energy_2x2_4site(st, env)
    e.backward()
    parallel.average_grads(list(sites.values()))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), energy=float(e.detach()), **{f"grad_{k[0]}_{k[1]}": v.grad.numpy() for k, v in sites.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("generic_ad_D2_chi8_f64", 2), ("generic_ad_D2_chi8_c128", 2), ("generic_ad_D2_chi8_f64_j3", 4),
                                        ("generic_ad_D2_chi8_f64", 3)])
def test_sharded_differentiable_move_gives_the_reference_gradient(tmp_path, name, world):
    """The differentiable route under torch.distributed: sites sharded over the ranks, exchanges as autograd nodes (cotangents summed
    over ranks and returned to the owner), local gradients averaged -- energy and gradient on every rank equal the REFERENCE's
    single-process autograd (tests/golden/generic_ad_*.npz).  World 3: uneven ownership (per-key broadcasts in the forward)."""
    import socket
    from conftest import golden
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ad_worker, args=(world, port, str(tmp_path), name), nprocs=world, join=True)
    g = golden(name)
    for r in range(world):
        out = np.load(tmp_path / f"rank{r}.npz")
        assert abs(float(out["energy"]) - float(g["energy"])) < 1e-11, r
        for k in out.files:
            if k.startswith("grad_"):
                assert float(np.abs(out[k] - g[k]).max()) < 1e-9 * max(1.0, float(np.abs(g[k]).max())), (r, k)


@pytest.mark.parametrize("name,world", [("generic_D2_chi8_f64", 8), ("generic_D2_chi8_c128", 8), ("generic_D2_chi8_f64", 5)])
def test_units_shared_by_rank_pairs_equal_single_process(tmp_path, name, world):
    """Twice as many ranks as sites: the ranks {i, i + 4} both work on unit i (ctmrg._ctm_MOVE_units: `shared`; owner of the exchanges = the
    lower rank), attach their pair to the engine for the projector phase of every move -- float64 only; complex128 and rank counts other
    than 2 x Nsites keep the one-owner sharding -- and every rank ends with the single-process environment."""
    os.environ["CTM_TEST_SHARED_UNITS"] = "1"
    try:
        test_sharded_move_equals_single_process(tmp_path, name, world)
    finally:
        os.environ.pop("CTM_TEST_SHARED_UNITS", None)
