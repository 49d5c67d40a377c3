import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "peps-torch_amd"))
import numpy as np, torch, _native
eng = _native.engine()
n, chi = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "graded"
inner = int(sys.argv[4]) if len(sys.argv) > 4 else 12
rng = np.random.default_rng(n)
if kind == "graded":
    M = rng.standard_normal((n, n)) @ (rng.standard_normal((n, n)) * np.exp(-8.0 * np.arange(n) / n)[None, :]) / n
elif kind == "steep":
    M = rng.standard_normal((n, n)) @ (rng.standard_normal((n, n)) * np.exp(-40.0 * np.arange(n) / n)[None, :]) / n
else:
    M = rng.random((n, n))
Md = torch.from_numpy(M).cuda()
eng.set_option("jacobi_verbose", 1); eng.set_option("jacobi_inner_sweeps", inner)
torch.cuda.synchronize(); t0 = time.perf_counter()
U, S, V = eng.truncated_svd(Md, chi)
torch.cuda.synchronize(); print(f"n={n} {kind} inner={inner}: {time.perf_counter()-t0:.3f} s sweeps={eng.stat('last_sweeps')}")
