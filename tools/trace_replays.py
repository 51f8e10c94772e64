#!/usr/bin/env python3
"""Per-replay view of a rocprofv3 --kernel-trace of bench.py: the witness kernel's dispatches grouped into runs of back-to-back launches
(a gap of more than --gap-us starts a new run); per run: launches, mean kernel duration, mean gap to the next launch, span / launches.
  tools/trace_replays.py <dir with *kernel_trace.csv> [--kernel witness_pm] [--gap-us 500]"""
import csv, glob, os, sys
d = sys.argv[1]
kern = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "witness"
gap_us = float(sys.argv[sys.argv.index("--gap-us") + 1]) if "--gap-us" in sys.argv else 500.0
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
runs, cur = [], []
for s, e, n in rows:
    if kern not in n:
        continue
    if cur and (s - cur[-1][1]) / 1e3 > gap_us:
        runs.append(cur); cur = []
    cur.append((s, e))
if cur:
    runs.append(cur)
print("runs of back-to-back %s launches: %d" % (kern, len(runs)))
for i, run in enumerate(runs):
    n = len(run)
    dur = [(e - s) / 1e3 for s, e in run]
    gaps = [(run[k + 1][0] - run[k][1]) / 1e3 for k in range(n - 1)]
    span = (run[-1][1] - run[0][0]) / 1e3
    first = " ".join("%.1f" % x for x in dur[:6])
    print("run %2d: %4d launches  dur mean %.2f us (min %.2f max %.2f)  gap mean %.2f us  span/launch %.2f us   first durations: %s" %
          (i, n, sum(dur) / n, min(dur), max(dur), sum(gaps) / max(len(gaps), 1), span / n, first))
