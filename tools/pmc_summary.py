#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: sum of each counter over all
dispatches of a kernel and the kernel's dispatch count."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void goss::", "").replace("goss::", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add((f, r["Dispatch_Id"]))
for name in sorted(acc, key=lambda n: -acc[n].get("SQ_WAVE_CYCLES", acc[n].get("FETCH_SIZE", 0))):
    print(name[:60].ljust(60), "calls=%d" % len({d for f, d in calls[name]}))
    for c in sorted(acc[name]):
        print("    %-28s %.4g" % (c, acc[name][c]))
