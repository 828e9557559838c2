#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (the *_kernel_trace.csv under the given
directory): total busy time, total gap time and the gap histogram -- what a hipGraph or fewer launches could recover."""
import csv
import glob
import sys

import numpy as np

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
o = np.argsort(st)
st, en = st[o], en[o]
gap = st[1:] - np.maximum.accumulate(en)[:-1]
busy = (en - st).sum()
small = gap[(gap > 0) & (gap < 50_000)]
print(f"{len(rows)} kernels; busy {busy / 1e6:.2f} ms; span {(en.max() - st.min()) / 1e6:.2f} ms")
print(f"gaps < 50 us: n={len(small)} total {small.sum() / 1e6:.2f} ms mean {small.mean() / 1e3:.2f} us "
      f"median {np.median(small) / 1e3:.2f} us; overlapping launches: {(gap <= 0).sum()}")
print(f"gaps >= 50 us (host-side pauses): n={(gap >= 50_000).sum()} total {gap[gap >= 50_000].sum() / 1e6:.2f} ms")
