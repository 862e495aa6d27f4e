"""Which limit a kernel sits at, from a `rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass: instructions per wave and the share of the SIMDs' cycles in which a vector
instruction issued (near 1: instruction-bound; far below with few waves: latency-bound).  usage: pmc_issue.py <dir> [name filter ...]"""
import csv, glob, sys, collections
d = sys.argv[1]
flt = sys.argv[2:]
cf = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(cf)):
    key = (r["Dispatch_Id"], r["Kernel_Name"])
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if "Start_Timestamp" in r:
        acc[key]["_wall_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
per = collections.defaultdict(lambda: collections.defaultdict(float))
for (did, name), c in acc.items():
    p = per[name]
    p["n"] += 1
    for k, v in c.items():
        p[k] += v
def short(n):
    return n.replace("void ", "").replace("asx::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
print("%-48s %4s %9s %9s %8s %8s %10s %10s" % ("kernel", "n", "wall us", "waves", "valu/wv", "salu/wv", "valu busy", "wave occ"))
for name, p in sorted(per.items(), key=lambda kv: -kv[1]["_wall_ns"]):
    if flt and not any(f in name for f in flt):
        continue
    n = p["n"]
    wall = p["_wall_ns"] / n
    if wall < 3e4:
        continue
    gui = p["GRBM_GUI_ACTIVE"] / n / 8.0  # cycles of the launch (the counter is summed over the 8 XCDs)
    waves = p["SQ_WAVES"] / n
    simd_cycles = gui * 1024.0
    # SQ_ACTIVE_INST_VALU counts, per SIMD... summed over the chip: quad-cycles in which a VALU instruction was in flight
    busy = p["SQ_ACTIVE_INST_VALU"] / n * 4.0 / simd_cycles if simd_cycles else 0.0
    occ = p["SQ_WAVE_CYCLES"] / n * 4.0 / simd_cycles if simd_cycles else 0.0  # mean resident waves per SIMD
    print("%-48s %4d %9.1f %9.0f %8.0f %8.0f %10.3f %10.2f" % (short(name), n, wall / 1e3, waves, p["SQ_INSTS_VALU"] / max(1.0, p["SQ_WAVES"]),
          p["SQ_INSTS_SALU"] / max(1.0, p["SQ_WAVES"]), busy, occ))
