"""RCCL on the one GPU this build loop has: a process group of ONE rank on backend "nccl" (= RCCL on ROCm), with
`parallel.single_rank_is_distributed` set so that every collective of peps-torch_amd/parallel.py really executes -- communicator
init, device buffers, the uniform in-place all-gather and the per-key broadcast branch of exchange(), float64 and complex128 (the
(re,im) view), sub-groups (new_group), the integer reductions, the autograd exchange + gradient averaging -- and then two whole
sharded sweeps of a 2x2 cell against the same sweeps without a process group (bit-identical: one rank owns every unit).
Also attaches the communicator-shaped handle to the native contexts (ctm_set_comm, one-rank group).  Prints one JSON line."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "peps-torch_amd")); sys.path.insert(0, REPO)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch
import torch.distributed as dist
import config as cfg
cfg.global_args.device = "cuda:0"
torch.cuda.set_device(0)
import _native, parallel
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
from models import j1j2


def sweeps(dtype):
    rng = np.random.default_rng(5)
    D, chi = 3, 18
    sites = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, D, D, D, D)) - 0.5 + (1j * (rng.random((2, D, D, D, D)) - 0.5) if dtype == torch.complex128 else 0.0)
            sites[(x, y)] = torch.from_numpy(A / np.abs(A).max()).to(dtype).cuda()
    st = IPEPS(sites)
    env = ENV(chi, st); init_env(st, env)
    for _ in range(2):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _r in range(2):
                ctmrg.ctm_MOVE(d, st, env)
    e = j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env)
    return env, complex(e)


ref = {dt: sweeps(dt) for dt in (torch.float64, torch.complex128)}

port = int(sys.argv[1]) if len(sys.argv) > 1 else 29533
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
parallel.single_rank_is_distributed = True
assert parallel.is_distributed() and parallel.world() == (0, 1)
out = {"backend": dist.get_backend(), "checks": []}
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(1)
for dtype in (torch.float64, torch.complex128):
    like = torch.empty(0, dtype=dtype, device=dev)
    mk = lambda *s: torch.randn(*s, dtype=dtype, device=dev, generator=g)
    # uniform branch, one tensor per rank (the tensor itself is the send buffer) and several (in-place gather)
    for per in (1, 3):
        keys = [f"k{i}" for i in range(per)]
        loc = {k: mk(7, 5, 4) for k in keys}
        got = parallel.exchange(loc, keys, {k: (7, 5, 4) for k in keys}, like)
        assert all(torch.equal(got[k], loc[k]) for k in keys)
    # per-key broadcast branch (different shapes)
    keys = ["a", "b"]
    loc = {"a": mk(6, 6), "b": mk(3, 2, 2)}
    got = parallel.exchange(loc, keys, {"a": (6, 6), "b": (3, 2, 2)}, like)
    assert all(torch.equal(got[k], loc[k]) for k in keys)
    # sub-group all-reduce (rank group of a site's plaquette evaluation) and the integer reductions
    t = mk(16, 16)
    parallel.prepare_groups([[0]])
    assert torch.equal(parallel.allreduce_sum_group(t.clone(), [0]), t)
    assert parallel.allreduce_min_int_group(5, [0], dev) == 5 and parallel.allreduce_max_int(7, dev) == 7
    assert parallel.allreduce_sum_scalar(1.25, dev) == 1.25
    # autograd exchange + gradient averaging
    x = mk(4, 4).requires_grad_(True)
    y = parallel.exchange_ad({"x": x * 2.0}, ["x"], {"x": (4, 4)}, like)["x"]
    (y.abs() ** 2).sum().backward()
    parallel.average_grads([x])
    assert torch.allclose(x.grad, 8.0 * x.detach())
    out["checks"].append(str(dtype))
    # two sharded sweeps + energy (comm timed), against the run without a process group
    parallel.comm_timing = True; parallel.comm_time_s(reset=True)
    env, e = sweeps(dtype)
    out[f"comm_s_{dtype}"] = parallel.comm_time_s(reset=True); parallel.comm_timing = False
    renv, re_ = ref[dtype]
    assert e == re_, (e, re_)
    for k in renv.C:
        assert torch.equal(env.C[k], renv.C[k]), k
    for k in renv.T:
        assert torch.equal(env.T[k], renv.T[k]), k
    out[f"energy_{dtype}"] = [e.real, e.imag]
# the communicator handle at the C-ABI: a one-rank group attaches, a larger one without a communicator is refused loudly
eng = _native.engine()
eng.set_comm(None, 0, 1)
try:
    eng.set_comm(None, 0, 2)
    raise SystemExit("ctm_set_comm accepted a two-rank group without a communicator")
except _native.NativeError as ex:
    out["set_comm_two_ranks"] = str(ex)
# ... and the RCCL path of the shared corner pass (include/ctm_hip.h: ctm_set_comm; rows_times_shared in csrc/svd_leading.hip) with a raw
# ONE-rank ncclComm_t made through librccl's own API (_native.Engine.attach_rccl_group -- what CTM_GROUP_TRANSPORT=rccl uses for a pair): every corner pass of a unit's truncation then computes its one "column block" (the
# whole product) into the staging buffer, ncclAllGather on the context's stream gathers it onto itself and the unpack kernel writes it back
# -- the exact code a pair executes, with one part.  Same bits as the unit without a communicator.
from ctm.generic.ctm_components import _halves_t
rng = np.random.default_rng(23)
D, chi = 4, 48
sites = {(x, y): torch.from_numpy((lambda a: a / np.abs(a).max())(rng.random((2, D, D, D, D)) - 0.5)).cuda() for y in range(2) for x in range(2)}
st = IPEPS(sites); env = ENV(chi, st); init_env(st, env)
parallel.single_rank_is_distributed = False
for _ in range(2):
    for d in cfg.ctm_args.ctm_move_sequence:
        for _r in range(2):
            ctmrg.ctm_MOVE(d, st, env)
t16 = _halves_t((0, -1), (0, 0), st, env)
P0, Pt0, S0 = eng.projectors_4x4((0, -1), t16, chi, return_S=True)
c0 = eng.stat("comm_calls")
eng.attach_rccl_group([0])           # ncclGetUniqueId, its trip through torch.distributed, ncclCommInitRank, ctm_set_comm
P1, Pt1, S1 = eng.projectors_4x4((0, -1), t16, chi, return_S=True)
eng.sync()
eng.detach_group()
out["rccl_shared_passes_one_rank"] = int(eng.stat("comm_calls") - c0)
assert out["rccl_shared_passes_one_rank"] > 0
assert torch.equal(S0, S1) and torch.equal(P0, P1) and torch.equal(Pt0, Pt1), "one-rank RCCL passes changed the bits"
_native.Engine._librccl().ncclCommDestroy(eng._rccl_comms.pop((0,)))
dist.barrier()
dist.destroy_process_group()
import ctypes
ctypes.CDLL(None).fflush(None)          # RCCL's own "Librccl path" line sits in C stdio until flushed: keep the JSON line last
print(json.dumps(out), flush=True)
