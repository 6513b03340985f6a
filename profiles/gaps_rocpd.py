#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd SQLite): for every ordered pair of
kernel names (A then B on the device timeline) the median / mean of start(B) - end(A), next to the kernel durations.

    cd /tmp && rocprofv3 --kernel-trace --stats -d DIR -o g -- python profiles/adam_steps.py
    python profiles/gaps_rocpd.py DIR/g_results.db"""
import sqlite3
import sys
from collections import defaultdict

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: n.split("(")[0].replace("void ", "")
gaps, durs = defaultdict(list), defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
    gaps[(short(n0), short(n1))].append((s1 - e0) * 1e-3)
for n, s, e in rows:
    durs[short(n)].append((e - s) * 1e-3)
print("# %s: %d dispatches" % (sys.argv[1], len(rows)))
print("%-40s %8s %10s %10s" % ("kernel", "calls", "median_us", "mean_us"))
for n, v in sorted(durs.items(), key=lambda kv: -len(kv[1])):
    print("%-40s %8d %10.2f %10.2f" % (n[:40], len(v), np.median(v), np.mean(v)))
print()
print("%-40s -> %-40s %8s %10s %10s" % ("kernel", "next kernel", "pairs", "median_us", "mean_us"))
for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 5:
        print("%-40s -> %-40s %8d %10.2f %10.2f" % (a[:40], b[:40], len(v), np.median(v), np.mean(v)))
