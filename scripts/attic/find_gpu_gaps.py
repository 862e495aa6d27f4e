import csv, glob, sys
d = sys.argv[1]
ops = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))))
ops.sort()
prev_end = ops[0][1]
big = []
for i, (s, e, n) in enumerate(ops):
    gap = (s - prev_end) / 1e6
    if gap > 30:
        big.append((gap, i))
    prev_end = max(prev_end, e)
print(len(ops), "operations;", len(big), "gaps over 30 ms")
for gap, i in big:
    print("gap %.1f ms at t=%.1f ms" % (gap, (ops[i][0] - ops[0][0]) / 1e6))
    for j in range(max(0, i - 4), min(len(ops), i + 3)):
        print("   %s %10.3f ms  dur %8.3f ms  %s" % ("->" if j == i else "  ", (ops[j][0] - ops[0][0]) / 1e6, (ops[j][1] - ops[j][0]) / 1e6, ops[j][2]))
