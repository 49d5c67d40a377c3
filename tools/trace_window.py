"""Print the kernel timeline of one CTM sweep from a rocprofv3 --kernel-trace CSV (diagnostic for latency-bound paths).
usage: trace_window.py <dir with *_kernel_trace.csv> <marker kernel substring> <index of the marker launch to start at> [n markers]"""
import csv, glob, sys
d, marker, start = sys.argv[1], sys.argv[2], int(sys.argv[3])
nmark = int(sys.argv[4]) if len(sys.argv) > 4 else 1
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
print("markers:", len(marks))
a, b = marks[start], marks[start + nmark]
t0 = rows[a][0]
prev_end = t0
busy = 0
agg = {}
for s, e, name in rows[a:b]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  {short}")
    prev_end = max(prev_end, e)
    busy += e - s
    k = agg.setdefault(short, [0, 0]); k[0] += 1; k[1] += e - s
print(f"window {(rows[b][0] - t0) / 1e3:.1f} us, kernels {b - a}, busy {busy / 1e3:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"  {v[1] / 1e3:9.1f} us  {v[0]:4d} x  {k}")
