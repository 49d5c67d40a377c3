"""GPU busy time (union of kernel intervals) and per-kernel totals from a rocprofv3 kernel_trace.csv."""
import csv, sys, glob, os
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
iv, tot = [], defaultdict(float)
for r in csv.DictReader(open(f)):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    iv.append((a, b)); tot[r["Kernel_Name"][:70]] += (b - a)
iv.sort()
t0, t1 = iv[0][0], max(b for _, b in iv)
busy, ca, cb = 0, None, None
for a, b in iv:
    if cb is None or a > cb:
        if cb is not None: busy += cb - ca
        ca, cb = a, b
    else: cb = max(cb, b)
busy += cb - ca
print(f"kernels {len(iv)}  span {1e-9*(t1-t0):.3f} s  busy(union) {1e-9*busy:.3f} s  sum {1e-9*sum(b-a for a,b in iv):.3f} s")
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:8]: print(f"  {1e-9*v:8.3f} s  {k}")
