"""Where a serially issued run leaves the GPU idle: gaps between consecutive kernels of a rocprofv3 kernel_trace.csv, by the pair of
kernels around the gap and by size class.  usage: trace_gaps.py DIR [skip_fraction]"""
import csv, sys, glob, os, re
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:40]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f)))
t0, t1 = ev[0][0], ev[-1][1]
ev = [e for e in ev if e[0] >= t0 + skip * (t1 - t0)]
span = ev[-1][1] - ev[0][0]
busy = 0; gaps = defaultdict(lambda: [0, 0.0]); classes = defaultdict(lambda: [0, 0.0]); after = defaultdict(lambda: [0, 0.0])
end = ev[0][1]; prev = ev[0][2]; busy = ev[0][1] - ev[0][0]
for a, b, n in ev[1:]:
    if a > end:
        g = a - end
        gaps[(prev, n)][0] += 1; gaps[(prev, n)][1] += g
        after[prev][0] += 1; after[prev][1] += g
        c = "<5us" if g < 5e3 else "<10us" if g < 1e4 else "<20us" if g < 2e4 else "<50us" if g < 5e4 else "<200us" if g < 2e5 else "<1ms" if g < 1e6 else ">=1ms"
        classes[c][0] += 1; classes[c][1] += g
        busy += b - a; end = b; prev = n
    else:
        if b > end: busy += b - end; end = b; prev = n
print(f"kernels {len(ev)}  span {1e-9*span:.3f} s  busy {1e-9*busy:.3f} s  idle {1e-9*(span-busy):.3f} s ({100*(span-busy)/span:.1f} %)")
print("gap size classes:")
for c in ["<5us", "<10us", "<20us", "<50us", "<200us", "<1ms", ">=1ms"]:
    print(f"  {c:8s} {classes[c][0]:8d} gaps  {1e-6*classes[c][1]:10.1f} ms")
print("idle after kernel (top 12):")
for k, v in sorted(after.items(), key=lambda x: -x[1][1])[:12]: print(f"  {1e-6*v[1]:10.1f} ms  {v[0]:8d} gaps  avg {1e-3*v[1]/v[0]:7.1f} us  after {k}")
print("idle by kernel pair (top 25):")
for k, v in sorted(gaps.items(), key=lambda x: -x[1][1])[:25]: print(f"  {1e-6*v[1]:10.1f} ms  {v[0]:8d} gaps  avg {1e-3*v[1]/v[0]:7.1f} us  {k[0]} -> {k[1]}")
