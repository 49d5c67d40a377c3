"""Development probe: the stationary path on the SAME operator (one projector unit called repeatedly with the same tensors and workspace)."""
import os, sys, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "peps-torch_amd"))
import numpy as np, torch
import config as cfg
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
from ctm.generic.ctm_components import _halves_t
import _native
D, chi = int(sys.argv[1]), int(sys.argv[2])
eng = _native.engine()
rng = np.random.default_rng(11)
sites = {(x, y): None for y in range(2) for x in range(2)}
for k in sites:
    A = rng.random((2, D, D, D, D)) - 0.5
    sites[k] = torch.from_numpy(A / np.abs(A).max()).cuda()
st = IPEPS(sites); env = ENV(chi, st); init_env(st, env)
args = copy.deepcopy(cfg.ctm_args); args.native_move = False
for _ in range(3):
    for d in args.ctm_move_sequence:
        for _r in range(2):
            ctmrg.ctm_MOVE(d, st, env, ctm_args=args)
d = (0, -1)
t16 = _halves_t(d, (0, 0), st, env)
n = chi * D * D
basis = eng.warm_basis(chi, n, torch.float64)
eng.set_option("warm_accept_tol", 1e-9); eng.set_option("jacobi_verbose", 2)
ref = None
for call in range(5):
    a0, r0, l0 = eng.stat("warm_accepts"), eng.stat("warm_rejects"), eng.stat("lz_hits")
    P, Pt, S = eng.projectors_4x4(d, t16, chi, return_S=True, basis=basis)
    print(f"call {call}: accepted {int(eng.stat('warm_accepts') - a0)} refused {int(eng.stat('warm_rejects') - r0)} krylov {int(eng.stat('lz_hits') - l0)}  hdr {basis[-1, :10].cpu().numpy()}", flush=True)
    PPt = P @ Pt.t()
    if ref is None: ref = (S.clone(), PPt.clone())
    else: print(f"   |S - S0|/s0 = {float((S - ref[0]).abs().max() / ref[0][0]):.2e}   |P Pt^T - ref| = {float((PPt - ref[1]).abs().max() / ref[1].abs().max()):.2e}   biorth |Pt^T P - 1| = {float((Pt.t() @ P - torch.eye(chi, device=P.device, dtype=P.dtype)).abs().max()):.2e}", flush=True)
R, Rt = eng.halves(d, t16)
M = R.t() @ Rt
sv = torch.linalg.svdvals(M.cpu())
print("true s[60..70]/s0:", (sv[60:70] / sv[0]).numpy())
print("true s[38..43]/s0:", (sv[38:43] / sv[0]).numpy())
print("engine S[60..64]/s0:", (S[60:64] / S[0]).cpu().numpy(), " max |S - true|/s0 over chi:", float((S.cpu() - sv[:chi]).abs().max() / sv[0]))
U_, S_, Vh_ = torch.linalg.svd(M.cpu())
W = basis[:chi + 1].cpu()
print("warm rows vs true right vectors: |W_i . v_i| for i = 38..42, 62..64:", [round(float((W[i] @ Vh_[i]).abs()), 6) for i in (38, 39, 40, 41, 42, 62, 63, 64)])
print("norm of warm rows outside the true top-(chi+1) right space:", float(torch.linalg.norm(W - (W @ Vh_[:chi + 1].T) @ Vh_[:chi + 1]) ))
