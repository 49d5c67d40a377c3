"""Strip GEMMs of the block iteration (b x n times n x n, and the transposed forms): achieved HBM bandwidth."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "peps-torch_amd"))
import torch, _native
eng = _native.engine()
for kv in os.environ.get('OPTS','').split(','):
    if kv: eng.set_option(kv.split('=')[0], float(kv.split('=')[1]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
B = torch.randn(n, n, dtype=torch.float64, device="cuda")
for b in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "8,16,32,48,64,96,128,257".split(","))]:
    A = torch.randn(b, n, dtype=torch.float64, device="cuda")
    At = torch.randn(n, b, dtype=torch.float64, device="cuda")
    for name, fn in (("A(bxn) B", lambda: eng.gemm(A, B)), ("A B^T", lambda: eng.gemm(A, B, transB=True)),
                     ("B At(nxb)", lambda: eng.gemm(B, At)), ("B^T At", lambda: eng.gemm(B, At, transA=True)),
                     ("B A^T", lambda: eng.gemm(B, A, transB=True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"n={n} b={b:4d} {name:10s} {dt*1e3:7.3f} ms  {n*n*8/dt/1e9:7.0f} GB/s  {2*b*n*n/dt/1e12:6.2f} TF", flush=True)
