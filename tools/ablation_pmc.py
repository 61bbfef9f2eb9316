#!/usr/bin/env python3
"""Per-phase instruction counts of k_mask_annotate_q20 from `rocprofv3 --pmc ... -- scratch/ma` (the ablation harness
launches the kernel 11 times per stop point): prints per-read VALU / scalar / LDS instructions and wave cycles."""
import collections
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0])))
d = collections.OrderedDict()
for r in rows:
    if "q20" not in r["Kernel_Name"]:
        continue
    d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(d)
per = int(sys.argv[2]) if len(sys.argv) > 2 else 11      # q20 launches per stop point
pick = int(sys.argv[3]) if len(sys.argv) > 3 else 5     # which of them to read
prev = None
for m in range(len(ids) // per):
    x = d[ids[m * per + pick]]
    w = x["SQ_WAVES"]
    cur = (x["SQ_INSTS_VALU"] / w, x["SQ_INSTS_SALU"] / w, x["SQ_INSTS_LDS"] / w, 4 * x["SQ_WAVE_CYCLES"] / w)
    delta = "" if prev is None or m >= 5 else "   (+%.0f VALU +%.0f scalar +%.0f LDS)" % tuple(c - p for c, p in zip(cur[:3], prev[:3]))
    print("stop %d: VALU %.0f scalar %.0f LDS %.0f cycles/wave %.0f%s" % ((m + 1,) + cur + (delta,)))
    prev = cur
