#!/usr/bin/env python3
"""Aggregate rocprofv3 `*_counter_collection.csv` files (one or more --pmc passes) per kernel.

    python tools/pmc_summary.py OUT.csv DIR_OR_CSV [DIR_OR_CSV ...]

Writes Kernel_Name, Counter_Name, launches, sum, mean, mean_dur_us — the format of
profiles/*_pmc_per_kernel.csv that bench.py reads for `roofline.traffic`."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    files = []
    for s in srcs:
        files += [s] if s.endswith(".csv") else sorted(glob.glob(os.path.join(s, "**", "*counter_collection.csv"),
                                                                 recursive=True))
    acc = defaultdict(lambda: [0, 0.0, 0.0])          # (kernel, counter) -> launches, sum, dur_ns
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                key = (row["Kernel_Name"], row["Counter_Name"])
                a = acc[key]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                try:
                    a[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Counter_Name", "launches", "sum", "mean", "mean_dur_us"])
        for (k, c), (n, s, d) in sorted(acc.items()):
            w.writerow([k, c, n, s, s / n, d / n / 1e3])
    print(f"{out}: {len(acc)} rows from {len(files)} file(s)")


if __name__ == "__main__":
    main()
