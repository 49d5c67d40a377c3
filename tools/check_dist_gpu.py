"""Sharded move on GPU ranks vs one process: run under torch.distributed.run with N ranks (gloo or nccl; ranks may share one
device with CTM_BENCH_ONE_DEVICE=1), rank 0 writes the corner spectra and a checksum of the environment after two sweeps of
a low-rank state at n >= 8192 (so that the masked-column absorb and the corner cache are active).  Compare the files of two runs."""
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
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
D, chi = 8, 128
rng = np.random.default_rng(3)
signed = len(sys.argv) > 2 and sys.argv[2] == "signed"
sites = {}
for y in range(2):
    for x in range(2):
        A = rng.random((2, D, D, D, D)) - (0.5 if signed else 0.0)
        sites[(x, y)] = torch.from_numpy(A / np.abs(A).max()).cuda()
st = IPEPS(sites)
env = ENV(chi, st); init_env(st, env)
for _ in range(2):
    for d in cfg.ctm_args.ctm_move_sequence:
        for _r in range(2):
            ctmrg.ctm_MOVE(d, st, env)
spec = env.get_spectra()
if rank == 0:
    out = {f"{k}": v.cpu().tolist() for k, v in spec.items()}
    out["checksum"] = float(sum(float(t.abs().sum()) for t in list(env.C.values()) + list(env.T.values())))
    out["ncol"] = {str(k): v for k, v in (env.__dict__.get("_ncol") or {}).items()}
    json.dump(out, open(sys.argv[1], "w"))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
