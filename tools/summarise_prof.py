"""Summarise rocprofv3 outputs (kernel stats + PMC csv) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-60s calls %6s  avg %10.2f us  total %10.2f ms  %5s%%" % (
            row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3,
            float(row["TotalDurationNs"]) / 1e6, row["Percentage"]))

for f in find("*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:48]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
    print("== counters:", os.path.relpath(f, out))
    for k, d in acc.items():
        for c, v in sorted(d.items()):
            n = cnt[(k, c)]
            print("  %-48s %-24s per-dispatch avg %16.1f  (n=%d)" % (k, c, v / n, n))
