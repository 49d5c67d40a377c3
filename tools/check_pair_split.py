"""A unit shared by a pair of ranks (twice as many ranks as sites; include/ctm_hip.h: ctm_set_comm_ops): run under
torch.distributed.run with 2 ranks (gloo; both ranks may share one device with CTM_BENCH_ONE_DEVICE=1) on a ONE-site cell -- the two ranks
then form the group of the only unit: every corner pass of its truncation is split by output columns inside the native solver and
all-gathered in the pair -- or alone (1 rank: the unsplit solve).  Every rank writes the corner spectra, a checksum of its environment after
`nsweeps` sweeps and the number of shared passes.  usage: check_pair_split.py out_prefix [D chi nsweeps]"""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "peps-torch_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import torch.distributed as dist
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = 0 if os.environ.get("CTM_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group(os.environ.get("CTM_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
import config as cfg
cfg.global_args.device = f"cuda:{local}"
import _native
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
D, chi, nsweeps = (int(a) for a in (sys.argv[2:5] + ["4", "48", "3"][len(sys.argv) - 2:]))
rng = np.random.default_rng(17)
A = rng.random((2, D, D, D, D)) - 0.5
st = IPEPS({(0, 0): torch.from_numpy(A / np.abs(A).max()).cuda()}, lX=1, lY=1)
env = ENV(chi, st); init_env(st, env)
eng = _native.engine()
for _ in range(nsweeps):
    for d in cfg.ctm_args.ctm_move_sequence:
        ctmrg.ctm_MOVE(d, st, env)
spec = env.get_spectra()
out = {f"{k}": v.cpu().tolist() for k, v in spec.items()}
out["checksum"] = float(sum(float(t.abs().sum()) for t in list(env.C.values()) + list(env.T.values())))
out["shared_passes"] = int(eng.stat("comm_calls"))
out["krylov_solves"] = int(eng.stat("lz_hits"))
out["power_iteration_solves"] = int(eng.stat("si_hits"))
json.dump(out, open(sys.argv[1] + f".rank{rank}.json", "w"))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
