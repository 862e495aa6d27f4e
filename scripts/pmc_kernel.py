#!/usr/bin/env python3
"""Mean of every counter of one kernel over its dispatches, from rocprofv3 --pmc csv output dirs.
usage: pmc_kernel.py <kernel substring> <dir> [<dir> ...]"""
import csv, glob, sys, collections
pat = sys.argv[1]
acc = collections.defaultdict(list)
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (disp, name), v in per.items():
            acc[name].append(v)
for name, vals in sorted(acc.items()):
    print("%-32s mean %.6g over %d dispatches (min %.6g max %.6g)" % (name, sum(vals) / len(vals), len(vals), min(vals), max(vals)))
