"""Differentiable sharded moves on GPU ranks (torch.distributed.run, gloo or nccl; ranks may share one device with
CTM_BENCH_ONE_DEVICE=1): energy after the moves of a gradient golden and its gradient with respect to the four site tensors; every
rank writes <out>.rank<r>.npz.  usage: check_dist_gpu_ad.py OUT GOLDEN_NAME"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "peps-torch_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import torch.distributed as dist
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = 0 if os.environ.get("CTM_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group(os.environ.get("CTM_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
import config as cfg
cfg.global_args.device = f"cuda:{local}"
from helpers_cpu import sites_from, env_from
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV
from ctm.generic import ctmrg
from models import j1j2
import parallel
g = np.load(os.path.join(REPO, "tests", "golden", sys.argv[2] + ".npz"))
b = np.load(os.path.join(REPO, "tests", "golden", str(g["base"]) + ".npz"))
if b["site_0_0"].dtype.kind == "c":
    cfg.global_args.torch_dtype = torch.complex128
sites = {k: torch.from_numpy(v.copy()).cuda().requires_grad_(True) for k, v in sites_from(b).items()}
st = IPEPS(sites, lX=2, lY=2)
C, T = env_from(b, "warm_")
env = ENV(next(iter(C.values())).shape[0], st)
env.C = {k: torch.from_numpy(v.copy()).cuda() for k, v in C.items()}
env.T = {k: torch.from_numpy(v.copy()).cuda() for k, v in T.items()}
for d in g["moves"]:
    ctmrg.ctm_MOVE(tuple(int(x) for x in d), st, env)
e = j1j2.J1J2(j1=1.0, j2=float(g["j2"]), j3=float(g["j3"])).energy_2x2_4site(st, env)
e.backward()
parallel.average_grads(list(sites.values()))
np.savez(sys.argv[1] + f".rank{rank}.npz", energy=float(e.detach()), **{f"grad_{k[0]}_{k[1]}": v.grad.cpu().numpy() for k, v in sites.items()})
if world > 1:
    dist.barrier(); dist.destroy_process_group()
