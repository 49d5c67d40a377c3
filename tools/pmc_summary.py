#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel.

    python tools/pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> > profiles/rNN_..._pmc_hbm_traffic.csv

FETCH_SIZE / WRITE_SIZE are reported in KiB units by rocprofv3 (x1024 -> bytes) and FETCH_SIZE under-counts by 2x on gfx950
(MI355X_MICROARCH.md, HBM section): hbm_bytes = 2 * FETCH * 1024 + WRITE * 1024."""
import csv, glob, os, sys
from collections import defaultdict


def load(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"]
                tot[k] += float(row["Counter_Value"]); cnt[k] += 1
    return tot, cnt


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    ft, fc = load(fd, "FETCH_SIZE")
    wt, wc = load(wd, "WRITE_SIZE")
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KB_total", "WRITE_SIZE_KB_total", "hbm_bytes_per_launch(2x_fetch_corrected)"])
    rows = []
    for k in set(ft) | set(wt):
        n = max(fc.get(k, 0), wc.get(k, 0), 1)
        b = (2.0 * ft.get(k, 0.0) + wt.get(k, 0.0)) * 1024.0
        rows.append((b, k, n, ft.get(k, 0.0), wt.get(k, 0.0), b / n))
    for b, k, n, f_, w_, per in sorted(rows, reverse=True):
        w.writerow([k, n, round(f_, 1), round(w_, 1), int(per)])


if __name__ == "__main__":
    main()
