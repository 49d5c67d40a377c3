import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "peps-torch_amd")); sys.path.insert(0, REPO)
import torch, numpy as np
import bench, config as cfg, _native
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
from ctm.generic.ctm_components import _halves_t
D, chi = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0); cfg.global_args.device = "cuda:0"
sites = bench.synth_sites("generic", D)
sites = {k: 2.0 * v - 1.0 for k, v in sites.items()}
sites = {k: v / np.abs(v).max() for k, v in sites.items()}
st = IPEPS({k: torch.from_numpy(v).to(dev) for k, v in sites.items()})
env = ENV(chi, st); init_env(st, env)
for _ in range(2):
    for d in cfg.ctm_args.ctm_move_sequence:
        for _r in range(2): ctmrg.ctm_MOVE(d, st, env)
eng = _native.engine()
R, Rt = eng.halves((0, -1), _halves_t((0, -1), (0, 0), st, env))
M = eng.gemm(R, Rt, True, False)
S = torch.linalg.svdvals(M.cpu()).numpy()
S = S / S[0]
k = chi + 1
print("n", M.shape[0], "k", k)
for f in (0.1, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0, 3.0, 5.0, 10.0):
    i = min(int(f * k), len(S) - 1); print(f"  S[{i}]/S0 = {S[i]:.3e}")
np.save(os.path.join(REPO, "gpurun_out", f"hard_spectrum_D{D}_chi{chi}.npy"), S)
