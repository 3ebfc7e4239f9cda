#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass (csv) per kernel.

usage: tools/pmc_traffic.py <dir with *_counter_collection.csv> [...]
On gfx950 FETCH_SIZE reads exactly HALF the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md §HBM): the x2 correction is applied to FETCH_SIZE and stated in the output;
WRITE_SIZE is taken as is. Both counters are in KiB.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    rows = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name", "?")
                short = name.split("(")[0].replace("void ppals::", "").replace("ppals::", "")
                rows[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"{'kernel':50s} {'counter':12s} {'launches':>8s} {'avg value':>14s} {'avg bytes (corrected)':>22s}")
    for k in sorted(rows):
        for c, vals in rows[k].items():
            avg = sum(vals) / len(vals)
            b = avg * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
            print(f"{k[:50]:50s} {c:12s} {len(vals):8d} {avg:14.1f} {b:22.4e}")


if __name__ == "__main__":
    main()
