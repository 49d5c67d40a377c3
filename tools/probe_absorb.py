"""Absorb step per direction at D=8 chi=256: time with the gather / scatter parts of the fused two-layer kernel disabled."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "peps-torch_amd"))
import torch, _native
eng = _native.engine()
D, chi = 8, 256
g = lambda *s: torch.rand(*s, dtype=torch.float64, device="cuda")
a = g(2, D, D, D, D)
Tl = {0: g(chi, D * D, chi), 1: g(chi, chi, D * D), 2: g(D * D, chi, chi), 3: g(chi, D * D, chi)}   # UP LEFT DOWN RIGHT
# per direction: (T1, T, T2) directions (ctmrg.py _ABS): UP: T1=(1,0) RIGHT, T=UP, T2=LEFT ; LEFT: T1=UP,T=LEFT,T2=DOWN ; DOWN: T1=LEFT,T=DOWN,T2=RIGHT ; RIGHT: T1=DOWN,T=RIGHT,T2=UP
trip = {0: (3, 0, 1), 1: (0, 1, 2), 2: (1, 2, 3), 3: (2, 3, 0)}
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
P = [g(chi * D * D, chi) for _ in range(4)]
for d in range(4):
    t1, t, t2 = trip[d]
    ten = (g(chi, chi), Tl[t1], Tl[t], Tl[t2], g(chi, chi), a, P[0], P[1], P[2], P[3])
    res = []
    for dbg in (0, 1, 4, 5, 2):
        eng.set_option("layer2_dbg", dbg)
        res.append(timed(lambda: eng.absorb(d, ten, normalize=False)))
    eng.set_option("layer2_dbg", 0)
    print(f"absorb dir {d}: full {res[0]:.2f} ms | no gather {res[1]:.2f} | no scatter {res[2]:.2f} | neither {res[3]:.2f} | no mfma {res[4]:.2f}", flush=True)
