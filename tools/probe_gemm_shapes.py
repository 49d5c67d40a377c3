"""Histogram of the GEMM shapes of one sweep of a bench configuration (debug option gemm_log)."""
import sys, os, collections, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    sys.argv = ["bench.py", "--no-cpu-baseline", "--serial-units", "--steps", "1", "--warmup", "2", "--opt", "gemm_log=1"] + sys.argv[2:]
    import runpy
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
else:
    r = subprocess.run([sys.executable, __file__, "--child"] + sys.argv[1:], capture_output=True, text=True)
    c = collections.Counter(l.strip() for l in r.stderr.splitlines() if l.startswith("GEMM"))
    for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:60]:
        print(v, k)
    print(r.stdout[-300:])
