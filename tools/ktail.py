#!/usr/bin/env python3
"""Development aid: the last kernels of a rocprofv3 kernel trace with their durations and the gaps between them.
usage: ktail.py <dir> [count]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
prev = None
for s, e, k, g in rows[-n:]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%-42s grid %8s dur %8.1f us gap %7.1f us" % (k, g, (e - s) / 1e3, gap))
    prev = e
