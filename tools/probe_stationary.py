"""Development probe of the stationary fast path (ctm_args.projector_warm_tol): per sweep -- accepted / refused / full solves, the corner-spectra
convergence measure, and the header words (distance, side, run, skip, fails) of one unit's workspace.  python tools/probe_stationary.py D chi [tol] [sweeps]"""
import os, sys, time, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "peps-torch_amd"))
import numpy as np, torch
import config as cfg
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env, ctmrg_conv_specC
from ctm.generic import ctmrg
import _native
D, chi = int(sys.argv[1]), int(sys.argv[2])
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
nsw = int(sys.argv[4]) if len(sys.argv) > 4 else 30
eng = _native.engine()
rng = np.random.default_rng(11)
sites = {}
for y in range(2):
    for x in range(2):
        A = rng.random((2, D, D, D, D)) - 0.5
        if os.environ.get("CPLX"):
            A = A + 1j * (rng.random((2, D, D, D, D)) - 0.5)
        sites[(x, y)] = torch.from_numpy(A / np.abs(A).max()).cuda()
args = copy.deepcopy(cfg.ctm_args); args.projector_warm_tol = tol; args.ctm_conv_tol = 0.0; args.ctm_max_iter = 10 ** 6
st = IPEPS(sites); env = ENV(chi, st); init_env(st, env)
hist = None
for i in range(nsw):
    a0, r0, l0, s0 = eng.stat("warm_accepts"), eng.stat("warm_rejects"), eng.stat("lz_hits"), eng.stat("si_hits")
    if os.environ.get("VERBOSE") and i >= int(os.environ["VERBOSE"]):
        for e in [eng] + list(eng.workers): e.set_option("jacobi_verbose", 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for d in args.ctm_move_sequence:
        for _ in range(2):
            ctmrg.ctm_MOVE(d, st, env, ctm_args=args)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for e in [eng] + list(eng.workers): e.set_option("jacobi_verbose", 0)
    _, hist = ctmrg_conv_specC(st, env, hist, ctm_args=args)
    ws = env.__dict__.get("_warm", {})
    key = sorted(ws.keys())[0] if ws else None
    hdr = ws[key][-1, :10].cpu().numpy() if key is not None else None
    dists = sorted(float(b[-1, 5]) for b in ws.values())
    print(f"sweep {i:2d} {1e3 * dt:8.1f} ms  accepted {int(eng.stat('warm_accepts') - a0):2d} refused {int(eng.stat('warm_rejects') - r0):2d} krylov {int(eng.stat('lz_hits') - l0):2d} "
          f"power {int(eng.stat('si_hits') - s0):2d}  conv {hist['conv_crit'][-1]:.2e}  dist min/med/max {dists[0]:.1e} {dists[len(dists) // 2]:.1e} {dists[-1]:.1e}  hdr[{key}] {np.array2string(hdr, precision=2)}", flush=True)
