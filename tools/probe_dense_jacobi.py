"""Time of the dense many-panel block Jacobi SVD (the Ritz extraction's kernel sequence) alone: n x n graded matrix through
ctm_truncated_svd with the leading-k iteration off.  usage: probe_dense_jacobi.py n [opt=value ...]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "peps-torch_amd"))
import numpy as np, torch, _native
n = int(sys.argv[1])
eng = _native.engine()
eng.set_option("si_enable", 0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); eng.set_option(k, float(v))
rng = np.random.default_rng(3)
U, _ = np.linalg.qr(rng.standard_normal((n, n))); V, _ = np.linalg.qr(rng.standard_normal((n, n)))
s = np.exp(-16.0 * (np.arange(n) / n) ** 0.5)
M = torch.from_numpy((U * s) @ V.T).cuda()
cfg = eng.cfg(keep_multiplets=False)
for rep in range(3):
    eng.timers(reset=True); torch.cuda.synchronize(); t0 = time.perf_counter()
    Ug, Sg, Vg = eng.truncated_svd(M, n // 3, cfg)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n} {' '.join(sys.argv[2:])}: {dt*1e3:.2f} ms, jacobi sweeps {eng.stat('total_sweeps')/max(eng.stat('jacobi_calls'),1):.1f}, err {float((Sg.cpu() - torch.from_numpy(s[:n//3])).abs().max()):.1e}", flush=True)
