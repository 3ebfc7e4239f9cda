#!/usr/bin/env python3
"""Pick the launches of bench.py's TIMED region out of a rocprofv3 kernel trace of the same command.

usage: tools/trace_timed_launches.py <bench line .json> <..._kernel_trace.csv> [kernel substring]

bench.py records the host clocks (monotonic / boottime / realtime, ns) at both ends of the timed
region of every record (`timed_region_ns`); the trace's Start/End timestamps are in one of those
clock domains — the one whose window contains launches is used. Prints the launches of the scan
kernel inside the headline's window and their average, next to `roofline.avg_launch_ms`.
"""
import csv
import json
import sys


def main():
    line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]
    d = json.loads(line)
    pat = sys.argv[3] if len(sys.argv) > 3 else "k_scan_suffix"
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if pat in r["Kernel_Name"]]
    reg = d.get("timed_region_ns") or {}
    for clock, (a, b) in reg.items():
        inside = [r for r in rows if a <= int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= b]
        if not inside:
            continue
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in inside]
        names = sorted({r["Kernel_Name"].split("(")[0].replace("void ppals::", "") for r in inside})
        print(f"clock {clock}: {len(inside)} launches of {names} inside the timed region "
              f"({(b - a) / 1e6:.3f} ms): avg {sum(dur) / len(dur):.2f} us, min {min(dur):.2f}, max {max(dur):.2f}")
        r = d.get("roofline", {})
        print(f"bench.py by HIP events: {r.get('launches')} launches, avg {1e3 * r.get('avg_launch_ms', 0):.2f} us")
        return
    print("no clock domain of timed_region_ns contains launches of", pat)


if __name__ == "__main__":
    main()
