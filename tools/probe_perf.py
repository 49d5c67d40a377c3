"""GPU probe: raw GEMM rate and Jacobi SVD/eigh behaviour (sweeps, time) -- development aid."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "peps-torch_amd"))
import numpy as np, torch
import _native
eng = _native.engine()

def tgemm(M, N, K, tA=False, tB=False, reps=5):
    A = torch.rand((K, M) if tA else (M, K), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((N, K) if tB else (K, N), dtype=torch.float64, device="cuda") - 0.5
    eng.gemm(A, B, tA, tB); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): eng.gemm(A, B, tA, tB)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"gemm {M}x{N}x{K} tA={int(tA)} tB={int(tB)}: {dt*1e3:.3f} ms  {2*M*N*K/dt/1e12:.2f} TF", flush=True)

def tsvd(n, chi, kind="graded"):
    rng = np.random.default_rng(n)
    if kind == "graded":
        M = rng.standard_normal((n, n)) * np.exp(-8.0 * np.arange(n) / n)[None, :]
        M = rng.standard_normal((n, n)) @ M / n
    else:
        M = rng.random((n, n))
    Md = torch.from_numpy(M).cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    U, S, V = eng.truncated_svd(Md, chi)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sw, off = eng.stat("last_sweeps"), eng.stat("last_offnorm")
    Sr = np.linalg.svd(M, compute_uv=False)[:chi] if n <= 2048 else None
    err = float(np.abs(S.cpu().numpy() - Sr).max() / Sr[0]) if Sr is not None else float("nan")
    Un, Vn = U.cpu().numpy(), V.cpu().numpy()
    orthU = np.abs(Un.T @ Un - np.eye(chi)).max(); orthV = np.abs(Vn.T @ Vn - np.eye(chi)).max()
    res = np.abs(Un.T @ M @ Vn - np.diag(S.cpu().numpy())).max() / S[0].item()
    print(f"svd n={n} chi={chi} {kind}: {dt:.3f} s sweeps={sw:.0f} off={off:.2e} Serr={err:.2e} orthU={orthU:.1e} orthV={orthV:.1e} res={res:.1e}", flush=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "svd"]
    if "gemm" in which:
        for s in (1024, 2048, 4096, 8192):
            tgemm(s, s, s)
        tgemm(4096, 4096, 4096, True, False); tgemm(4096, 4096, 4096, False, True)
        tgemm(4608, 128, 4608); tgemm(128, 128, 4608, True, False); tgemm(64, 4608, 64)
    if "svd" in which:
        for n, chi in ((256, 32), (1024, 64), (2048, 128)):
            tsvd(n, chi)
        tsvd(1024, 64, "dense")
    if "svdbig" in which:
        tsvd(4608, 128)
