#!/usr/bin/env python3
"""per-kernel totals from a rocprofv3 rocpd database (the default output of ROCm 7.x):
tools/rocpd_stats.py <dir-or-db> [top=20] -> name, calls, total us, avg us"""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    for db in dbs:
        c = sqlite3.connect(db)
        rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3 "
                         "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        print(f"# {db}: {sum(r[1] for r in rows)} launches, {tot / 1e3:.3f} ms of kernel time")
        print("name,calls,total_us,avg_us,min_us,percent")
        for r in rows[:top]:
            print(f"\"{r[0][:110]}\",{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]:.2f},{100 * r[2] / tot:.1f}")


if __name__ == "__main__":
    main()
