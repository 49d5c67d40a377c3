"""Kernel timeline of one CTM sweep from a rocprofv3 --kernel-trace CSV (diagnostic for the latency-bound paths).
usage: trace_window.py <dir with *_kernel_trace.csv> <delimiter kernel substring> <which delimiter launch (negative: from the end)> [must-contain substring]
The window runs from the chosen launch of the delimiter kernel (one per sweep, e.g. layer2_reg_kernel) to the next one; with a
fourth argument the last window that contains a kernel of that name is taken instead (fifth: that many windows later)."""
import bisect, csv, glob, sys
d, marker, which = sys.argv[1], sys.argv[2], int(sys.argv[3])
must = sys.argv[4] if len(sys.argv) > 4 else None
shift = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # with `must`: that many delimiter windows later
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if must:
    hits = [i for i, r in enumerate(rows) if must in r[2]]
    w = bisect.bisect_left(marks, hits[-1]) - 1 + shift
    a, b = marks[w], marks[w + 1]
else:
    a, b = marks[which], marks[which + 1]
t0 = rows[a][0]; prev = t0; busy = 0; agg = {}
for s, e, name in rows[a:b]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  {short}")
    prev = max(prev, e); busy += e - s
    k = agg.setdefault(short, [0, 0]); k[0] += 1; k[1] += e - s
print(f"window {(rows[b][0] - t0) / 1e3:.1f} us, kernels {b - a}, busy {busy / 1e3:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"  {v[1] / 1e3:9.1f} us  {v[0]:4d} x  {k}")
