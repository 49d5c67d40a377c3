#!/usr/bin/env python
"""MFMA utilisation per kernel from ONE rocprofv3 counter pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (both fit one pass: SQ and
GRBM slots are independent, MI355X_MICROARCH.md).

    python tools/pmc_mfma_summary.py <dir of the pass> > profiles/rNN_..._pmc_mfma.csv

rocprofv3 sums both counters over the 8 XCDs: SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles x the number of FP64 16x16x4 MFMAs of ALL 1024 SIMDs
(checked: gemm_rows_kernel<4,true>, M = 64, K = N = 16384: 3.436e10 flop / 2048 flop per MFMA x 64 = 1.0737e9 = the counter per launch,
exactly), GRBM_GUI_ACTIVE = 8 x the active cycles of the launch.
  util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)   -- fraction of the chip's matrix-pipe cycles that were busy
  clock = GRBM_GUI_ACTIVE / 8 / launch duration (durations: the kernel_stats.csv of the same command)
util x 78.6 TFLOP/s x (clock / 2.4 GHz) is the rate the pipes delivered."""
import csv, glob, os, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    cnt[k] += 1
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", "SQ_VALU_MFMA_BUSY_CYCLES_total", "GRBM_GUI_ACTIVE_total", "active_cycles_per_launch = gui_active / 8 / launches",
                "mfma_util = busy / (gui_active / 8 * 1024)"])
    rows = []
    for k, c in tot.items():
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        rows.append((busy, k, cnt[k], act, busy / (act / 8.0 * 1024.0) if act > 0 else 0.0))
    for busy, k, n, act, u in sorted(rows, reverse=True)[:40]:
        w.writerow([k, n, int(busy), int(act), int(act / 8.0 / max(n, 1)), round(u, 4)])


if __name__ == "__main__":
    main()
