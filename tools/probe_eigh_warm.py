"""Development probe: warm restart of the symmetric truncation on a stationary synthetic matrix (verbose solver log)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "peps-torch_amd"))
import _native
eng = _native.engine()
variant = sys.argv[1] if len(sys.argv) > 1 else "alt"
eng.set_option("jacobi_verbose", int(sys.argv[2]) if len(sys.argv) > 2 else 0)
n, chi = 768, 48
for kv in sys.argv[3:]:
    k_, v_ = kv.split("="); eng.set_option(k_, float(v_))
sg = torch.tensor([1.0, -1.0]).repeat(30) if "alt" in variant else torch.ones(60)
tail = (0.1 * 0.5 ** torch.arange(n - 60, dtype=torch.float64)) if "fast" in variant else (0.1 * 0.9 ** torch.arange(n - 60, dtype=torch.float64))
top = (0.97 ** torch.arange(60, dtype=torch.float64)) if "geo" in variant else torch.linspace(1.0, 0.2, 60).double()
lam = torch.cat([top * sg, tail * float(top[-1]) / 0.2]).double()
g = torch.Generator().manual_seed(3)
Q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
A = ((Q * lam) @ Q.T).cuda()
basis = eng.warm_basis_c4v(chi, n)
for i in range(4):
    D, U = eng.truncated_eigh(A, chi, basis=basis)
    print(variant, "call", i, "hits", eng.stat("eigh_warm_hits"), "rejects", eng.stat("eigh_warm_rejects"), "si_hits", eng.stat("si_hits"), "fallbacks", eng.stat("si_fallbacks"),
          "norms", float(basis.norm(dim=1).min()), float(basis.norm(dim=1).max()), "si_iters", eng.stat("si_total_iters") if hasattr(eng, "stat") else None, flush=True)
