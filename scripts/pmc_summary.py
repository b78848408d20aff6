#!/usr/bin/env python3
"""Sums rocprofv3 --pmc CSV output per kernel and counter: pmc_summary.py <dir> [kernel substring]."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel"
for path in sorted(glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*/*/*counter_collection.csv")):
    acc, calls = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            if want in row["Kernel_Name"]:
                key = (row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])
                acc[key] += float(row["Counter_Value"])
                calls[key].add(row["Dispatch_Id"])
    for (k, c), v in sorted(acc.items()):
        n = len(calls[(k, c)])
        print("%-44s %-28s calls=%-3d per_call=%.6g" % (k, c, n, v / max(n, 1)))
