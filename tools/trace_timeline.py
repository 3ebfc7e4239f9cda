#!/usr/bin/env python3
"""timeline of a window of a rocprofv3 kernel trace (csv): start offset, duration and the idle gap
in front of every launch — what a chain of dependent launches really costs.
usage: tools/trace_timeline.py <kernel_trace.csv> <anchor-substring> <occurrence> [count=120]
       prints `count` launches starting at the `occurrence`-th launch whose name contains the anchor,
       then totals per kernel name inside the window (busy time, gaps in front)."""
import csv
import sys
from collections import defaultdict


def main():
    path, anchor, occ = sys.argv[1], sys.argv[2], int(sys.argv[3])
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 120
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    hits = [i for i, r in enumerate(rows) if anchor in r[2]]
    if occ >= len(hits):
        sys.exit(f"only {len(hits)} launches match {anchor!r}")
    i0 = hits[occ]
    win = rows[i0:i0 + count]
    t0 = win[0][0]
    busy = defaultdict(lambda: [0, 0.0, 0.0])
    prev_end = None
    print("#  start_us   dur_us   gap_us  queue  kernel")
    for s, e, n, q in win:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        short = n.split("(")[0].replace("void ", "").replace("ppals::", "")
        print(f"{(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {gap:8.2f}  {q:>4}  {short[:60]}")
        b = busy[short]
        b[0] += 1
        b[1] += (e - s) / 1e3
        b[2] += max(gap, 0.0)
        prev_end = max(prev_end, e) if prev_end is not None else e
    span = (max(r[1] for r in win) - t0) / 1e3
    print(f"# window: {len(win)} launches, {span:.1f} us")
    print("# kernel, calls, busy_us, gaps_in_front_us")
    for n, b in sorted(busy.items(), key=lambda kv: -kv[1][1]):
        print(f"# {n[:60]:60s} {b[0]:4d} {b[1]:9.1f} {b[2]:9.1f}")


if __name__ == "__main__":
    main()
