"""Dev tool: print the last N kernel launches of a rocprofv3 --kernel-trace CSV with the gaps between them."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prev = None
for s, e, name in rows[-n:]:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print("gap %8.1f us  run %8.1f us  %s" % ((s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, name))
    prev = e
