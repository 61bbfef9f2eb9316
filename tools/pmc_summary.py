#!/usr/bin/env python3
"""Average rocprofv3 counter_collection.csv rows per (kernel, counter).

usage: pmc_summary.py <counter_collection.csv> [...]  > summary.csv
rocprofv3 writes one row per (dispatch, counter); kernels are keyed by the name up to the first '('.
"""
import csv
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: [0, 0.0])
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                k = (name, row["Counter_Name"])
                acc[k][0] += 1
                acc[k][1] += float(row["Counter_Value"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "dispatches", "mean_value"])
    for (name, ctr), (n, tot) in sorted(acc.items()):
        w.writerow([name, ctr, n, f"{tot / n:.1f}"])


if __name__ == "__main__":
    main(sys.argv[1:])
