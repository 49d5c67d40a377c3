"""Signed (full-rank) random state: time per sweep and truncation statistics as the environment converges."""
import sys, os, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "peps-torch_amd")); sys.path.insert(0, REPO)
import torch, numpy as np
import bench, config as cfg, _native
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env, ctmrg_conv_specC
from ctm.generic import ctmrg
D, chi, nsw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0); cfg.global_args.device = "cuda:0"
sites = bench.synth_sites("generic", D)
sites = {k: 2.0 * v - 1.0 for k, v in sites.items()}
sites = {k: v / np.abs(v).max() for k, v in sites.items()}
st = IPEPS({k: torch.from_numpy(v).to(dev) for k, v in sites.items()})
env = ENV(chi, st); init_env(st, env)
eng = _native.engine()
keys = ("si_hits", "si_total_iters", "lz_hits", "lz_total_steps", "si_fallbacks", "si_warm_starts", "si_warm_skips")
def stats():
    out = {}
    for k in keys:
        out[k] = eng.stat(k) + sum(w.stat(k) for w in getattr(eng, "workers", []))
    return out
hist = None
prev = stats()
for i in range(nsw):
    if i == nsw - 1 and len(sys.argv) > 4:
        eng.set_option('jacobi_verbose', 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for d in cfg.ctm_args.ctm_move_sequence:
        for _r in range(2): ctmrg.ctm_MOVE(d, st, env)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    conv, hist = ctmrg_conv_specC(st, env, hist)
    cur = stats()
    keys = keys if "si_warm_skips" in keys else keys + ("si_warm_skips",)
    print(f"sweep {i}: {dt:.3f} s  conv_crit {hist['conv_crit'][-1] if isinstance(hist, dict) else hist[-1]:.2e}  " +
          " ".join(f"{k}+{cur[k]-prev[k]:.0f}" for k in keys), flush=True)
    prev = cur
