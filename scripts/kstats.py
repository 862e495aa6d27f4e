import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in sys.argv[2:] or ["asx"]):
        print("%-100s calls %4s  avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"].replace("asx::(anonymous namespace)::", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
