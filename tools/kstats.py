#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 --kernel-trace --stats kernel_stats.csv found under the given directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(f)))[:n_rows]:
    n = r["Name"].replace("(anonymous namespace)::", "")[:84]
    print("%-84s calls=%6s tot_ms=%9.2f avg_us=%9.1f pct=%s" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                               float(r["AverageNs"]) / 1e3, r["Percentage"]))
