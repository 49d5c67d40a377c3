"""Text timeline of a window of a rocprofv3 kernel_trace.csv: start (us, relative), duration, queue, kernel.
usage: trace_timeline.py DIR frac_start window_ms"""
import csv, sys, glob, os, re
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
frac, win = float(sys.argv[2]), float(sys.argv[3]) * 1e6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Stream_Id"], r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"]) for r in csv.DictReader(open(f)))
t0, t1 = ev[0][0], ev[-1][1]
w0 = t0 + frac * (t1 - t0)
for a, b, q, s, n, gx, gy, wx in ev:
    if a < w0 or a > w0 + win: continue
    k = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "").split("(")[0][:30]
    print(f"{1e-3*(a-w0):10.1f} {1e-3*(b-a):8.1f} q{q} s{s} {k} wgs={int(gx)*int(gy)//int(wx)}")
