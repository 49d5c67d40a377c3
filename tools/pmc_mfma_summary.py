#!/usr/bin/env python
"""MFMA utilisation per kernel from ONE rocprofv3 counter pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (both fit one pass: SQ and
GRBM slots are independent, MI355X_MICROARCH.md).

    python tools/pmc_mfma_summary.py <dir of the pass> > profiles/rNN_..._pmc_mfma.csv

util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs): the fraction of the chip's matrix-pipe cycles that were busy
while the kernel ran (the gfx94x `MfmaUtil` formula; rocprofv3 sums the SQ counter over all shader engines / XCDs).  An FP64
16x16x4 MFMA holds its pipe 64 cycles, so util x 78.6 TFLOP/s x (clock / 2.4 GHz) is the rate the pipes delivered."""
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
    w.writerow(["kernel", "launches", "SQ_VALU_MFMA_BUSY_CYCLES_total", "GRBM_GUI_ACTIVE_total", "mfma_util = busy / (gui_active * 256 * 4)"])
    rows = []
    for k, c in tot.items():
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        rows.append((busy, k, cnt[k], act, busy / (act * 1024.0) if act > 0 else 0.0))
    for busy, k, n, act, u in sorted(rows, reverse=True)[:40]:
        w.writerow([k, n, int(busy), int(act), round(u, 4)])


if __name__ == "__main__":
    main()
