"""Development probe: SVDGESDD node on a rank-deficient M = R^T R~ of the D=2 chi=8 golden, native vs numpy/oracle formulas."""
import sys, os, numpy as np, torch
R_ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R_, "peps-torch_amd")); sys.path.insert(0, os.path.join(R_, "tests")); sys.path.insert(0, R_)
import _native, config as cfg
from helpers_cpu import sites_from, env_from
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV
from ctm.generic import ctm_ad
from linalg.svd_gesdd import SVDGESDD
from oracle import ctm_oracle as O
eng = _native.engine()
b = np.load(os.path.join(R_, "tests", "golden", "generic_D2_chi8_f64.npz"))
sites = {k: torch.from_numpy(v.copy()).cuda() for k, v in sites_from(b).items()}
st = IPEPS(sites, lX=2, lY=2)
C, T = env_from(b, "warm_")
env = ENV(8, st); env.C = {k: torch.from_numpy(v.copy()).cuda() for k, v in C.items()}; env.T = {k: torch.from_numpy(v.copy()).cuda() for k, v in T.items()}
with torch.no_grad():
    Rm, Rt = ctm_ad.halves((0, -1), (0, 0), st, env)
    M = (Rm.t() @ Rt).contiguous()
Mn = M.cpu().numpy()
print("singular values", np.linalg.svd(Mn, compute_uv=False))
g = torch.Generator().manual_seed(2)
W1 = torch.randn(32, 8, generator=g, dtype=torch.float64); W2 = torch.randn(32, 8, generator=g, dtype=torch.float64)
Mg = M.clone().requires_grad_(True)
U, S, V = SVDGESDD.apply(Mg, 1e-12)
print("native S", S.detach().cpu().numpy())
print("orth U", float((U.t() @ U - torch.eye(32, device=U.device)).abs().max()), "orth V", float((V.t() @ V - torch.eye(32, device=U.device)).abs().max()))
Ss = torch.rsqrt(S[:8])
loss = ((U[:, :8] * Ss) * W1.cuda()).sum() + ((V[:, :8] * Ss) * W2.cuda()).sum()
loss.backward()
dn = Mg.grad.cpu().numpy()
# oracle
Uo, So, Vo = O.truncated_svd_gesdd(Mn, 32)
Ut, St, Vt = (torch.from_numpy(x).requires_grad_(True) for x in (Uo, So, Vo))
Sst = torch.rsqrt(St[:8])
lo = ((Ut[:, :8] * Sst) * W1).sum() + ((Vt[:, :8] * Sst) * W2).sum()
lo.backward()
do = O.svd_backward(Uo, So, Vo, Ut.grad.numpy(), St.grad.numpy(), Vt.grad.numpy(), 1e-12)
print("loss native", float(loss), "oracle", float(lo))
print("|dM native|", np.abs(dn).max(), "|dM oracle|", np.abs(do).max(), "diff", np.abs(dn - do).max())
