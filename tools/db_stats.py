#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results .db (the default output format when --output-format is not given).
    python3 tools/db_stats.py <results.db> [top_n]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
agg = collections.defaultdict(lambda: [0, 0])
for name, s, e in db.execute("select name,start,end from kernels"):
    agg[name[:78]][0] += 1
    agg[name[:78]][1] += e - s
for k, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    print(f"{t / 1e6:9.3f} ms {n:6d} calls {t / n / 1e3:9.1f} us  {k}")
