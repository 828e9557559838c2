#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per (kernel, grid) averages of each counter."""
import collections
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.split("(")[0][-70:]
        if "conv_" not in name:
            continue
        cfg = name[name.find("Cfg<"):name.find(">", name.find("Cfg<")) + 1] + name[name.find(">", name.find("Cfg<")) + 1:][:12]
        key = (("halo " if "halo" in name else "gemm ") + cfg, r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
cnames = sorted({c for v in agg.values() for c in v})
print("kernel | grid | n | " + " | ".join(cnames))
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = max(len(x) for x in v.values())
    print(f"{key[0]:45s} {key[1]:>9s} {n:4d} " + " ".join(f"{sum(v[c]) / max(len(v[c]), 1):12.4g}" for c in cnames))
