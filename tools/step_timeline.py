"""Dev tool: per-step kernel timeline (durations and gaps) from a rocprofv3 --kernel-trace CSV.
usage: python tools/step_timeline.py <kernel_trace.csv> [steps_to_show]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:40]
scan = [i for i, r in enumerate(rows) if "vc_scan_kernel" in r[2]]
show = int(sys.argv[2]) if len(sys.argv) > 2 else 2
# a step = kernels after the previous scan's select up to and including this step's select
for si in scan[-show - 1:-1]:
    j = si
    while j > 0 and "vc_scan_kernel" not in rows[j - 1][2]:
        j -= 1
    # print from the first kernel after previous scan to the kernel before next scan
    k = si + 1
    while k < len(rows) and "vc_scan_kernel" not in rows[k][2]:
        k += 1
    print("--- step around scan #%d" % si)
    prev_end = None
    for s, e, n in rows[j:k]:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("  gap %7.1f us   run %8.1f us   %s" % (gap, (e - s) / 1e3, short(n)))
        prev_end = e
periods = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(scan[-show - 6:-1], scan[-show - 5:])]
print("scan-to-scan period (us):", " ".join("%.1f" % p for p in periods))
