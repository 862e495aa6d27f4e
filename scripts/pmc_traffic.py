#!/usr/bin/env python3
"""HBM-side traffic per kernel from two rocprofv3 --pmc passes of the same command (FETCH_SIZE and WRITE_SIZE cannot
share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <n> <dim> <commit> > profiles/rNN_pmc_traffic.json
gfx950 correction (same guide, HBM section): FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads ->
read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact (KB)."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    """Kernel name without return type, namespaces and the argument list.  The anonymous-namespace marker goes FIRST: it
    contains a parenthesis, and cutting at the first "(" before removing it turned every asx::(anonymous namespace)::kernel
    into the empty string (r02: 53 launches and 11.3 GB under "kernel": "")."""
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("asx::", "")
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):   # the argument list opens at the first "(" outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut][:110]


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for cf in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(cf)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
kernels = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
    nf, f = fetch.get(k, [0, 0.0])
    nw, w = write.get(k, [0, 0.0])
    launches = max(nf, nw, 1)
    total = (2.0 * f + w) * 1024.0
    if total < 1e6:
        continue
    kernels.append({"kernel": k, "launches": launches, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                    "hbm_bytes_total": total, "hbm_bytes_per_launch": total / launches})
print(json.dumps({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- "
                             "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs",
                  "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE counts half of wide streaming reads), "
                                "WRITE_SIZE x 1024 exact", "n": int(sys.argv[3]), "dim": int(sys.argv[4]), "commit": sys.argv[5],
                  "kernels": kernels}, indent=1))
