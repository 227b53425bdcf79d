#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_<tag>/pass*/**/*counter_collection.csv): per kernel and counter, mean per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmc_%s/pass*/**/*counter_collection.csv" % tag, recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("av1mi::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-24s mean %16.1f  n=%d" % (c, sum(v) / len(v), len(v)))
