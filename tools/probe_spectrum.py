import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "peps-torch_amd")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, config as cfg
cfg.global_args.device = "cuda:0"
import _native
from bench import synth_sites
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
from ctm.generic.ctm_components import _halves
D, chi = int(sys.argv[1]), int(sys.argv[2])
eng = _native.engine()
st = IPEPS({k: torch.from_numpy(v).cuda() for k, v in synth_sites("generic", D).items()})
env = ENV(chi, st); init_env(st, env)
for sw in range(int(sys.argv[3])):
    for d in cfg.ctm_args.ctm_move_sequence:
        for _ in range(2): ctmrg.ctm_MOVE(d, st, env)
    R, Rt = _halves((0, -1), (0, 0), st, env)
    P, Pt, S = eng.projectors(R, Rt, chi, return_S=True)
    s = (S / S[0]).cpu().numpy()
    print(f"sweep {sw+1}: S/S0 at idx 1,2,4,8,16,32,64,{chi-1}:", " ".join(f"{s[i]:.1e}" for i in (1, 2, 4, 8, 16, 32, 64, chi - 1) if i < chi), " #>1e-8:", int((s > 1e-8).sum()), "#>1e-14:", int((s > 1e-14).sum()))
