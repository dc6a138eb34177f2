#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/kstats_top.py <dir> [n] [steps]"""
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
for path in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: -float(r["TotalDurationNs"]))
    print(f"total kernel time {sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps:.3f} ms per step")
    for r in rows[:n]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:96]
        print(f"{name:96s} calls {int(r['Calls']) / steps:7.1f} total {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms  avg {float(r['AverageNs']) / 1e3:8.1f} us")
