"""Strip GEMMs of the block iteration (b x n times n x n, and the transposed form): achieved HBM bandwidth / MFMA rate of the
LDS-free streaming strip kernel and of the LDS-tiled row-block kernel."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "peps-torch_amd"))
import torch, _native
eng = _native.engine()
for kv in os.environ.get('OPTS','').split(','):
    if kv: eng.set_option(kv.split('=')[0], float(kv.split('=')[1]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
B = torch.randn(n, n, dtype=torch.float64, device="cuda")
variants = [("strip", {"rows_kernel_min_m": 1000, "rows_kernel_min_m_kc": 1000}),
            ("rows768", {"rows_kernel_min_m": 1, "rows_kernel_min_m_kc": 1, "rows_target_wgs": 768}),
            ("rows512", {"rows_kernel_min_m": 1, "rows_kernel_min_m_kc": 1, "rows_target_wgs": 512}),
            ("rows1024", {"rows_kernel_min_m": 1, "rows_kernel_min_m_kc": 1, "rows_target_wgs": 1024})]
for b in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "16,32,48,64".split(","))]:
    A = torch.randn(b, n, dtype=torch.float64, device="cuda")
    for vname, opts in variants:
        for k_, v_ in opts.items(): eng.set_option(k_, v_)
        for name, fn in (("A B  (n-contig)", lambda: eng.gemm(A, B)), ("A B^T (k-contig)", lambda: eng.gemm(A, B, transB=True))):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print(f"n={n} b={b:4d} {vname:9s} {name:17s} {dt*1e3:7.3f} ms  {n*n*8/dt/1e9:7.0f} GB/s  {2*b*n*n/dt/1e12:6.2f} TF", flush=True)
