"""Which operations survive the guard allocator?  Run under LD_PRELOAD=tools/bin/libguard_malloc.so with different GUARD_ALIGN / GUARD_MODE."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "peps-torch_amd"))
import numpy as np
import torch
tag = f"align={os.environ.get('GUARD_ALIGN')} mode={os.environ.get('GUARD_MODE')} min={os.environ.get('GUARD_MIN')} preload={'guard' in os.environ.get('LD_PRELOAD', '')}"
def report(name, err):
    print(f"[{tag}] {name}: {err:.3e}", flush=True)
rng = np.random.default_rng(0)
for n in (5, 625, 1000, 4097):
    a = rng.standard_normal(n)
    t = torch.from_numpy(a).cuda()
    report(f"h2d-d2h n={n}", float(np.abs(t.cpu().numpy() - a).max()))
    report(f"dev add n={n}", float(np.abs((t + 1.0).cpu().numpy() - (a + 1.0)).max()))
    report(f"d2d clone n={n}", float(np.abs(t.clone().cpu().numpy() - a).max()))
A = rng.standard_normal((50, 37)); B = rng.standard_normal((37, 41))
tA, tB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
report("torch matmul", float(np.abs((tA @ tB).cpu().numpy() - A @ B).max()))
r = torch.rand(50, 50, dtype=torch.float64, device="cuda")
report("torch.rand mean-0.5", float(r.mean().cpu()) - 0.5)
import _native
eng = _native.engine()
report("eng.gemm", float(np.abs(eng.gemm(tA, tB).cpu().numpy() - A @ B).max()))
M = rng.standard_normal((96, 96)); tM = torch.from_numpy(M).cuda()
report("eng.svdvals", float(np.abs(eng.svdvals(tM).cpu().numpy() - np.linalg.svd(M, compute_uv=False)).max()))
U, S, V = eng.truncated_svd(tM, 96)
report("eng.truncated_svd recon", float(np.abs((U.cpu().numpy() * S.cpu().numpy()) @ V.cpu().numpy().T - M).max()))
print(f"[{tag}] done", flush=True)
