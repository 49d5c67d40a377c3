"""Per-sweep record of a generic run on the signed (full-rank) benchmark state: wall time, change of the normalised corner spectra,
block-Krylov steps / residual estimates, Jacobi sweeps.  usage: probe_sweep_conv.py D chi nsweeps [c128] [opt=value ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "peps-torch_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, config as cfg
cfg.global_args.device = "cuda:0"
import _native
from bench import synth_sites
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
D, chi, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
eng = _native.engine()
cplx = "c128" in sys.argv[4:]
for kv in [a for a in sys.argv[4:] if "=" in a]:
    k, v = kv.split("="); eng.set_option(k, float(v))
sites = synth_sites("generic", D, dtype="c128" if cplx else "f64", signed=True)
st = IPEPS({k: torch.from_numpy(v).cuda() for k, v in sites.items()})
env = ENV(chi, st); init_env(st, env)
prev = None
for sw in range(ns):
    eng.timers(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for d in cfg.ctm_args.ctm_move_sequence:
        for _ in range(2): ctmrg.ctm_MOVE(d, st, env)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    spec = {k: (s / s[0]).cpu().numpy() for k, s in env.get_spectra().items()}
    ds = max(np.abs(spec[k] - prev[k]).max() for k in spec) if prev else float("nan")
    prev = spec
    lz = eng.stat("lz_hits"); steps = eng.stat("lz_total_steps") / max(lz, 1)
    rank = min(int((s_ > 1e-8).sum()) for s_ in spec.values())
    print(f"sweep {sw+1:2d}: {dt:6.3f} s  dspec {ds:9.2e}  min corner rank {rank:3d}  hbm {torch.cuda.mem_get_info()[0] / 2**30:5.1f} GiB free  krylov solves {int(lz):3d} avg steps {steps:5.2f}  extractions {int(eng.stat('lz_extractions')):3d}  "
          f"power-iter hits {int(eng.stat('si_hits')):3d}  jacobi calls {int(eng.stat('jacobi_calls')):4d} avg sweeps {eng.stat('total_sweeps')/max(eng.stat('jacobi_calls'),1):5.2f}  async fallbacks {int(eng.stat('lz_async_fallbacks'))} third passes {int(eng.stat('lz_third_passes'))}  arena high {eng.stat('arena_high') / 2**30:.1f} GiB (all contexts)", flush=True)
