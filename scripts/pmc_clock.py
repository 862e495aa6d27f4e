"""Effective clock and MFMA-pipe occupancy of the long kernels from a `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES` run (MI355X_MICROARCH.md, DVFS give-back): clock = GRBM_GUI_ACTIVE / 8 /
wall; MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock x wall)."""
import csv, glob, sys, collections
d = sys.argv[1]
cf = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(cf)))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    key = (r["Dispatch_Id"], r["Kernel_Name"][:90])
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if "Start_Timestamp" in r:
        acc[key]["_wall_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for (did, name), c in acc.items():
    wall = c.get("_wall_ns", 0.0)
    if wall < 5e6:
        continue
    clk = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / wall  # GHz
    out = "%-90s wall %8.2f ms clock %.3f GHz" % (name, wall / 1e6, clk)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and clk > 0:
        out += "  mfma busy %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * clk * wall))
    if "SQ_BUSY_CU_CYCLES" in c and clk > 0:
        out += "  cu busy raw %.3e" % c["SQ_BUSY_CU_CYCLES"]
    print(out)
