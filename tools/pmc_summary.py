#!/usr/bin/env python3
"""Per-kernel, per-launch averages of the counters in rocprofv3 --pmc rocpd databases.
usage: python tools/pmc_summary.py <results.db> [<results.db> ...] [> profiles/<name>.txt]"""
import sqlite3
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    for name, counter, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if name.startswith("__amd_rocclr") or "at::native" in name or "at::" in name:
            continue
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        a = acc[short][counter]
        a[0] += value
        a[1] += 1
print("# rocprofv3 --pmc per-launch averages from: " + " ".join(sys.argv[1:]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print(f"    {c:28s} {s / n:16.1f}   ({n} launches)")
