import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "peps-torch_amd"))
import numpy as np, torch, _native
eng = _native.engine()
eng.set_option("jacobi_verbose", int(os.environ.get("JV", "1")))
def run(n, chi, kind):
    rng = np.random.default_rng(n)
    dec = {"graded": 8.0, "steep": 40.0}.get(kind)
    if dec: M = rng.standard_normal((n, n)) @ (rng.standard_normal((n, n)) * np.exp(-dec * np.arange(n) / n)[None, :]) / n
    else: M = rng.random((n, n))
    Md = torch.from_numpy(M).cuda()
    for si in (1, 0):
        if si == 0 and n > 2048: continue
        eng.set_option("si_enable", si)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        U, S, V = eng.truncated_svd(Md, chi)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        Un, Vn, Sn = U.cpu().numpy(), V.cpu().numpy(), S.cpu().numpy()
        res = np.abs(Un.T @ M @ Vn - np.diag(Sn)).max() / Sn[0]
        r2 = np.linalg.norm(M @ Vn - Un * Sn[None, :], axis=0).max() / Sn[0]
        oU = np.abs(Un.T @ Un - np.eye(chi)).max(); oV = np.abs(Vn.T @ Vn - np.eye(chi)).max()
        print(f"n={n} chi={chi} {kind} si={si}: {dt:.3f}s hits={eng.stat('si_hits'):.0f} fb={eng.stat('si_fallbacks'):.0f} iters={eng.stat('si_last_iters'):.0f} res={res:.1e} r2={r2:.1e} orthU={oU:.1e} orthV={oV:.1e}", flush=True)
for a in sys.argv[1:]:
    n, chi, kind = a.split(":")
    run(int(n), int(chi), kind)
