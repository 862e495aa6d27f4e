"""Idle gaps between the kernels of the LAST build in a rocprofv3 --kernel-trace csv: usage trace_gaps.py <dir> [min gap us]"""
import csv, glob, sys
d = sys.argv[1]; min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last build starts at the last row-norms kernel
starts = [i for i, r in enumerate(rows) if "row_norms" in r[2]]
i0 = starts[-1]
seg = rows[i0:]
t0 = seg[0][0]; t1 = max(r[1] for r in seg)
busy = 0; cur_end = seg[0][0]; gaps = []
for s, e, name in seg:
    if s > cur_end:
        gaps.append(((s - cur_end) / 1e3, prev, name))
    else:
        pass
    busy += max(0, e - max(s, cur_end))
    if e > cur_end:
        cur_end = e; prev = name
    if s == seg[0][0]:
        prev = name
print("last build: %.3f ms wall, %.3f ms busy, %.3f ms idle in %d gaps, %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), len(seg)))
def short(n):
    n = n.replace("void ", "").replace("asx::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]
for g, a, b in sorted(gaps, reverse=True):
    if g >= min_gap:
        print("  %8.1f us  after %-60s before %s" % (g, short(a), short(b)))
