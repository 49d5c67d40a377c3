"""Full singular spectrum of the explicit M = R^T Rt of one unit (host LAPACK) on the signed benchmark state after `nsweeps` sweeps:
input of the CPU emulations of the block Krylov solver / Ritz extraction.  usage: dump_unit_spectrum.py D chi nsweeps out.npy"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "peps-torch_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, config as cfg
cfg.global_args.device = "cuda:0"
import _native
from bench import synth_sites
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
from ctm.generic.ctm_components import _halves_t
D, chi, ns, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
eng = _native.engine()
sites = synth_sites("generic", D, signed=True)
st = IPEPS({k: torch.from_numpy(v).cuda() for k, v in sites.items()})
env = ENV(chi, st); init_env(st, env)
for sw in range(ns):
    for d in cfg.ctm_args.ctm_move_sequence:
        for _ in range(2): ctmrg.ctm_MOVE(d, st, env)
t16 = _halves_t((0, -1), (0, 0), st, env)
R, Rt = eng.halves((0, -1), t16)
M = eng.gemm(R, Rt, transA=True)
s = torch.linalg.svdvals(M.cpu()).numpy()
np.save(out, s)
print("n", M.shape[0], "s[:5]/s0", s[:5] / s[0], "s[chi]/s0", s[chi] / s[0], "s[2chi]", s[2 * chi] / s[0], "s[4chi]", s[4 * chi] / s[0], "s[-1]", s[-1] / s[0])
