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
# per kernel name: launches, mean wall time, clock and matrix-pipe busy fraction over its launches (kernels of 0.2 ms or more on average;
# until round 5 only launches of 5 ms or more were listed one by one -- at 1 M x 128 no launch is that long any more)
per = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for (did, name), c in acc.items():
    wall = c.get("_wall_ns", 0.0)
    if wall <= 0:
        continue
    p = per[name]
    p[0] += 1
    p[1] += wall
    p[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
    p[3] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
for name, (cnt, wall, gui, busy) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    if wall / cnt < 2e5:
        continue
    clk = gui / 8.0 / wall  # GHz
    out = "%-90s launches %3d  mean wall %7.3f ms  clock %.3f GHz" % (name, cnt, wall / cnt / 1e6, clk)
    if busy and clk > 0:
        out += "  mfma busy %.3f" % (busy / (1024.0 * clk * wall))
    print(out)
