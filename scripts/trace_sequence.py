"""The kernels of the LAST build in a rocprofv3 --kernel-trace csv, in order, with their durations and the gap in front of each:
usage trace_sequence.py <dir>"""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "row_norms" in r[2]]
seg = rows[starts[-1]:]
def short(n):
    n = n.replace("void ", "").replace("asx::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]
t0 = seg[0][0]; prev_end = t0
tot_gap = 0.0
for s, e, name in seg:
    gap = (s - prev_end) / 1e3
    tot_gap += max(0.0, gap)
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, short(name)))
    prev_end = max(prev_end, e)
print("total %.3f ms, gaps %.3f ms, %d kernels" % ((prev_end - t0) / 1e6, tot_gap / 1e3, len(seg)))
