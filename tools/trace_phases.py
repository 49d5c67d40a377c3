"""Wall time of a (concurrent) run split into: some chip-filling kernel active / only small kernels active / idle, from a rocprofv3
kernel_trace.csv.  usage: trace_phases.py DIR [skip_fraction]"""
import csv, sys, glob, os, re
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
HEAVY = ("gemm_rows_kernel", "layer2_", "gemm_f64_fast_kernel", "gemm_f64_kernel<4, 4>", "gemm_strip_kernel", "permute", "transpose")
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]; g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    heavy = any(h in n for h in HEAVY) and g >= 128
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), heavy, n))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= t0 + skip * (t1 - t0)]
pts = []
for a, b, h, _ in ev:
    pts.append((a, 1, h)); pts.append((b, -1, h))
pts.sort()
nh = nl = 0; last = pts[0][0]; T = {"heavy": 0, "light_only": 0, "idle": 0}; sumh = suml = 0
for t, d, h in pts:
    dt = t - last
    if nh > 0: T["heavy"] += dt
    elif nl > 0: T["light_only"] += dt
    else: T["idle"] += dt
    sumh += nh * dt; suml += nl * dt
    if h: nh += d
    else: nl += d
    last = t
span = pts[-1][0] - pts[0][0]
print(f"span {1e-9*span:.3f} s: heavy kernel active {1e-9*T['heavy']:.3f} s ({100*T['heavy']/span:.1f} %), only small kernels {1e-9*T['light_only']:.3f} s "
      f"({100*T['light_only']/span:.1f} %), idle {1e-9*T['idle']:.3f} s ({100*T['idle']/span:.1f} %)")
print(f"sum of kernel durations: heavy {1e-9*sumh:.3f} s, small {1e-9*suml:.3f} s")
from collections import defaultdict
tot = defaultdict(lambda: [0, 0.0])
for a, b, h, n in ev:
    k = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "").split("(")[0][:44]
    tot[k][0] += 1; tot[k][1] += b - a
for k, v in sorted(tot.items(), key=lambda x: -x[1][1])[:12]: print(f"  {1e-6*v[1]:9.1f} ms {v[0]:7d} calls avg {1e-3*v[1]/v[0]:8.1f} us  {k}")
