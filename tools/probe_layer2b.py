import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "peps-torch_amd"))
import torch, _native
eng = _native.engine()
for D, chi in ((8, 256), (7, 196), (6, 128), (5, 100), (4, 64), (3, 36)):
    g = lambda *s: torch.rand(*s, dtype=torch.float64, device="cuda")
    C, T1, T2, a = g(chi, chi), g(chi, D * D, chi), g(chi, chi, D * D), g(2, D, D, D, D)
    ref = None
    for reg in (-1, 1):
        eng.set_option("layer2_reg", reg)
        out = eng.c2x2(0, C, T1, T2, a); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): eng.c2x2(0, C, T1, T2, a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        if ref is None: ref = out
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f"D={D} chi={chi} layer2_reg={reg}: {dt:.3f} ms per corner  maxrel diff vs LDS variant {err:.2e}", flush=True)
