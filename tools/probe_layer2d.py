"""Fused two-layer kernel per corner type at D=8 chi=256: time with the gather / scatter / MFMA parts disabled (layer2_dbg)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "peps-torch_amd"))
import torch, _native
eng = _native.engine()
D, chi = 8, 256
g = lambda *s: torch.rand(*s, dtype=torch.float64, device="cuda")
a = g(2, D, D, D, D)
C = g(chi, chi)
# T layouts per direction: UP (chi,D2,chi) LEFT (chi,chi,D2) DOWN (D2,chi,chi) RIGHT (chi,D2,chi)
Tup, Tle, Tdo, Tri = g(chi, D * D, chi), g(chi, chi, D * D), g(D * D, chi, chi), g(chi, D * D, chi)
T12 = {0: (Tup, Tle), 1: (Tri, Tup), 2: (Tdo, Tri), 3: (Tle, Tdo)}
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for corner in range(4):
    T1, T2 = T12[corner]
    eng.set_option("use_layer2", 0)
    t_no = timed(lambda: eng.c2x2(corner, C, T1, T2, a), 2)
    eng.set_option("use_layer2", 1)
    res = []
    for dbg in (0, 1, 4, 5, 2):
        eng.set_option("layer2_dbg", dbg)
        res.append(timed(lambda: eng.c2x2(corner, C, T1, T2, a)))
    eng.set_option("layer2_dbg", 0)
    print(f"corner {corner}: full {res[0]:.2f} ms | no gather {res[1]:.2f} | no scatter {res[2]:.2f} | neither {res[3]:.2f} | no mfma {res[4]:.2f} | unfused path {t_no:.2f}", flush=True)
