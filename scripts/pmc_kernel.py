#!/usr/bin/env python3
"""Sums the PMC counters of the kernels whose name contains `pattern` in a rocprofv3 --pmc output directory.
usage: pmc_kernel.py <dir> <pattern>"""
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
tot, n = {}, {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = r["Counter_Name"]
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
            n[k] = n.get(k, 0) + 1
for k in sorted(tot):
    print("%-28s %16.0f  over %d launches" % (k, tot[k], n[k]))
