import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "peps-torch_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, config as cfg
cfg.global_args.device = "cuda:0"
import _native
from test_gpu_fullsize import _state, _sweep
from ctm.generic.env import ENV, init_env
eng = _native.engine()
for kv in sys.argv[1:]:
    k, v = kv.split("="); eng.set_option(k, float(v))
st = _state(6, 3, signed=True)
env = ENV(128, st); init_env(st, env)
for sw in range(2):
    _sweep(st, env, 1)
    bad = [k for k, v in list(env.C.items()) + list(env.T.items()) if not torch.isfinite(v).all()]
    print("sweep", sw + 1, "non-finite tensors:", len(bad), "lz_hits", eng.stat("lz_hits"), "fallbacks", eng.stat("lz_async_fallbacks"), "third", eng.stat("lz_third_passes"), "si_fallbacks", eng.stat("si_fallbacks"), flush=True)
