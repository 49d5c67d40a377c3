import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "peps-torch_amd"))
import torch, _native
eng = _native.engine()
D, chi = 8, 256
g = lambda *s: torch.rand(*s, dtype=torch.float64, device="cuda")
C, T1, T2, a = g(chi, chi), g(chi, D * D, chi), g(chi, chi, D * D), g(2, D, D, D, D)
for dbg in (0, 1, 4, 5, 2, 7):
    eng.set_option("layer2_dbg", dbg)
    eng.c2x2(0, C, T1, T2, a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): eng.c2x2(0, C, T1, T2, a)
    torch.cuda.synchronize()
    print(f"D={D} chi={chi} dbg={dbg} (1 no gather, 2 no mfma, 4 no scatter): {(time.perf_counter()-t0)/5*1e3:.2f} ms per corner", flush=True)
eng.set_option("layer2_dbg", 0)
