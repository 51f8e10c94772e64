#!/usr/bin/env python3
"""rocprofv3 --kernel-trace: the last N dispatches with durations and the gap to the previous one.  tools/trace_gaps.py <dir> [N]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
prev = None
for s, e, n in rows[-N:]:
    print("%8.1f us  gap %7.1f us  %s" % ((e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, n[:60]))
    prev = e
