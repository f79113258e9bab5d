#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --output-format csv of scripts/drive_regions.py -> the launch sequence of a region, averaged by position over the
regions of the trace (a region = the launches between two gaps of more than 12 us: the host's synchronisation): kernel, duration, idle gap before.
    python scripts/trace_regions.py <dir with *kernel_trace.csv> [ticks]"""
import csv, glob, os, re, statistics, sys
paths = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))
if not paths:
    sys.exit("no *kernel_trace.csv under " + sys.argv[1])
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void ow::|\(.*", "", r["Kernel_Name"])))
rows.sort()
regions, cur = [], []
for i, (s, e, k) in enumerate(rows):
    gap = (s - rows[i - 1][1]) / 1e3 if i else 0.0
    if i and gap > 12.0:
        regions.append(cur); cur = []
    cur.append((k, (e - s) / 1e3, gap))
regions.append(cur)
lens = [len(r) for r in regions]
mode = statistics.mode(lens)
typical = [r for r in regions[2:] if len(r) == mode]
print(f"{len(rows)} launches, {len(regions)} regions, {mode} launches in a typical region ({len(typical)} such regions)")
tot = 0.0
for pos in range(mode):
    d = statistics.median(r[pos][1] for r in typical); g = statistics.median(r[pos][2] for r in typical)
    tot += d + (g if pos else 0.0)
    if pos < 4 or pos >= mode - 3:
        print(f"  launch {pos:3d}: {typical[0][pos][0][:48]:48s} {d:8.2f} us   gap before {g:7.2f} us" + ("  (the host's synchronisation + launch latency)" if pos == 0 else ""))
    elif pos == 4:
        mid = [statistics.median(r[q][1] for r in typical) for q in range(4, mode - 3)]
        gaps = [statistics.median(r[q][2] for r in typical) for q in range(4, mode - 3)]
        print(f"  launches 4 .. {mode - 4}: median duration {statistics.median(mid):.2f} us, median gap {statistics.median(gaps):.2f} us")
print(f"  first start -> last end of a region: {tot:.1f} us; gap between regions (median): {statistics.median(r[0][2] for r in typical):.1f} us")
