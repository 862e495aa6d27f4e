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
import os
import re
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_sha import source_sha16  # noqa: E402


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


def demangle_one(m):
    """Enough of the Itanium scheme for this library's kernels: _ZN <len><ident>... [I <integral template args> E] E ...
    -> ident<args>.  (binutils' c++filt in the image gives up on _Float16 parameter types, DF16_.)"""
    if not m.startswith("_ZN"):
        return m
    i, last = 3, None
    while i < len(m) and m[i].isdigit():
        j = i
        while m[j].isdigit():
            j += 1
        ln = int(m[i:j])
        last = m[j:j + ln]
        i = j + ln
    if last is None:
        return m
    args = []
    if i < len(m) and m[i] == "I":
        i += 1
        while i < len(m) and m[i] == "L":
            j = m.index("E", i)
            kind, val = m[i + 1], m[i + 2:j]
            val = ("-" + val[1:]) if val.startswith("n") else val
            args.append(("true" if val == "1" else "false") if kind == "b" else val + ("u" if kind == "j" else ""))
            i = j + 1
        if not (i < len(m) and m[i] == "E"):
            args = None   # a type argument: leave the argument list out
    return last + ("<" + ", ".join(args) + ">" if args else "")


def demangle(names):
    return {n: demangle_one(n) for n in names if n.startswith("_Z")}


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for cf in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(cf)) if r["Counter_Name"] == counter]
        plain = demangle({r["Kernel_Name"] for r in rows})
        for r in rows:
            k = short(plain.get(r["Kernel_Name"], r["Kernel_Name"]))
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
                  "recorded_unix": int(time.time()), "source_sha16": source_sha16(),
                  "kernels": kernels}, indent=1))
