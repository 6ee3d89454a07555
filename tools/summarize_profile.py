"""Dev tool: turn one tools/profile_round.sh output directory into the files committed under profiles/.
usage: python tools/summarize_profile.py gpurun_out/<tag> profiles/<prefix>
writes <prefix>_bench.json, <prefix>_bench_kernel_stats.csv, <prefix>_bench_pmc_scan_kernel.json and
<prefix>_scan_traffic.json (HBM bytes per vc_scan_kernel launch: FETCH_SIZE and WRITE_SIZE are in KB, collected in
separate --pmc passes; gfx950 correction for a wide coalesced stream = FETCH_SIZE x 2, MI355X_MICROARCH.md HBM section)."""
import csv
import glob
import json
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
KERNEL = "vc_scan_kernel"


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if len(files) != 1:
        raise SystemExit("expected one file for %s, found %d" % (pattern, len(files)))
    return files[0]


def pmc(pass_dir):
    """average counter value and duration per vc_scan_kernel dispatch of one --pmc pass"""
    acc, dur = {}, {}
    with open(one("%s/%s/**/*_counter_collection.csv" % (src, pass_dir))) as f:
        for row in csv.DictReader(f):
            if KERNEL not in row["Kernel_Name"]:
                continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    out = {k: sum(v) / len(v) for k, v in acc.items()}
    out["pmc_%s_avg_ms" % pass_dir.split("_", 1)[1]] = sum(dur.values()) / len(dur)
    out["pmc_%s_launches" % pass_dir.split("_", 1)[1]] = len(dur)
    return out


bench = json.load(open("%s/bench.json" % src))
shutil.copy("%s/bench.json" % src, prefix + "_bench.json")
shutil.copy(one("%s/stats/**/*_kernel_stats.csv" % src), prefix + "_bench_kernel_stats.csv")

counters = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    counters.update(pmc(d))
json.dump(counters, open(prefix + "_bench_pmc_scan_kernel.json", "w"), indent=1)

cfg = bench["config"]
alg = cfg["n_codes"] * cfg["bits"] // 8
hbm = (counters["FETCH_SIZE"] * 2 + counters["WRITE_SIZE"]) * 1024
json.dump({
    "kernel": KERNEL, "n_codes": cfg["n_codes"], "bits": cfg["bits"], "query_tile": cfg["query_tile"],
    "FETCH_SIZE_KB_per_launch": counters["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": counters["WRITE_SIZE"],
    "correction": "FETCH_SIZE x 2: on gfx950 FETCH_SIZE reports exactly half of a wide coalesced 16 B/lane streaming "
                  "read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is; separate --pmc passes (profiles/README.md)",
    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
}, open(prefix + "_scan_traffic.json", "w"), indent=1)
print(json.dumps({"scan_ms_bench": bench["roofline"]["avg_launch_ms"], "frac": bench["roofline"]["frac"],
                  "traffic_over_algorithmic": hbm / alg, **{k: v for k, v in counters.items() if k.startswith("pmc_")}}))
