"""Development probe: complex warm restart on a Hermitian matrix with +-m pairs."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "peps-torch_amd"))
import _native
eng = _native.engine()
eng.set_option("jacobi_verbose", 1)
n, chi = 768, 48
mod = 0.8 ** (torch.arange(n, dtype=torch.float64) // 2)
lam = mod * torch.where(torch.arange(n) % 2 == 0, 1.0, -1.0)
g = torch.Generator().manual_seed(21)
Z = torch.randn(n, n, generator=g, dtype=torch.float64) + 1j * torch.randn(n, n, generator=g, dtype=torch.float64)
Q, _ = torch.linalg.qr(Z)
A = ((Q * lam.to(torch.complex128)) @ Q.conj().T).cuda()
basis = eng.warm_basis_c4v(chi, n, A.dtype)
for i in range(3):
    D, U = eng.truncated_eigh(A, chi, basis=basis)
    print("call", i, "hits", eng.stat("eigh_warm_hits"), "rejects", eng.stat("eigh_warm_rejects"), "si_hits", eng.stat("si_hits"), "fallbacks", eng.stat("si_fallbacks"), flush=True)
