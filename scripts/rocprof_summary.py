#!/usr/bin/env python3
"""Summarise rocprofv3 output (rocpd sqlite or csv) into a small text table for profiles/."""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path, out):
    db = sqlite3.connect(path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    out.write("# kernel-trace stats (durations in us)\n# calls total_us avg_us pct name\n")
    for name, calls, total, avg, pct in rows:
        out.write(f"{calls:6d} {total:12.1f} {avg:10.3f} {pct:6.2f}  {name}\n")
    try:
        pmc = db.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
        if pmc:
            out.write("# counters: kernel counter dispatches sum avg_per_dispatch\n")
            for name, cname, n, s in pmc:
                out.write(f"{name.split('(')[0]:40s} {cname:28s} {n:5d} {s:18.1f} {s / n:16.1f}\n")
    except sqlite3.Error as e:
        out.write(f"# (no counter table: {e})\n")


def main():
    src = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    dbs = [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
    for d in dbs:
        out.write(f"## {os.path.relpath(d, os.path.dirname(src) or '.')}\n")
        from_db(d, out)


if __name__ == "__main__":
    main()
