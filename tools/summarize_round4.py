"""Dev tool: turn one tools/profile_round4.sh output directory into the files committed under profiles/.
usage: python tools/summarize_round4.py gpurun_out/<tag> profiles/r04
Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB per dispatch; on gfx950 FETCH_SIZE counts a wide coalesced
16 B/lane stream at half its bytes (MI355X_MICROARCH.md, HBM section) -> x 2 for the verify kernel's stream.  For the
MIH kernels' 4..64-byte gathers that correction is uncalibrated: the raw figure (64-byte requests) and x 2 are both recorded."""
import csv
import glob
import json
import os
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if len(files) != 1:
        raise SystemExit("expected one file for %s, found %d" % (pattern, len(files)))
    return files[0]


def pmc(pass_dir, kernel):
    """average counter value and duration per dispatch of `kernel` in one --pmc pass (first dispatch dropped: cold)"""
    acc, dur = {}, {}
    with open(one("%s/%s/**/*_counter_collection.csv" % (src, pass_dir))) as f:
        for row in csv.DictReader(f):
            if kernel not in row["Kernel_Name"]:
                continue
            acc.setdefault(row["Counter_Name"], {})[row["Dispatch_Id"]] = float(row["Counter_Value"])
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    if not dur:
        return None
    ids = sorted(dur, key=int)[1:] or sorted(dur, key=int)
    out = {k: sum(v[i] for i in ids) / len(ids) for k, v in acc.items()}
    out["pass_avg_ms"] = sum(dur[i] for i in ids) / len(ids)
    out["pass_launches"] = len(ids)
    return out


def bench(name):
    with open("%s/bench_%s.json" % (src, name)) as f:
        return json.load(f)


def stats(name, dst):
    """kernel-stats summary with the (very long) library kernel names cut to 110 characters"""
    with open(one("%s/stats_%s/**/*_kernel_stats.csv" % (src, name))) as f, open(dst, "w", newline="") as g:
        w = csv.writer(g, quoting=csv.QUOTE_MINIMAL)
        for row in csv.reader(f):
            row[0] = row[0][:110]
            w.writerow(row)


# ---- headline
b = bench("c3")
json.dump(b, open(prefix + "_bench.json", "w"))
stats("c3", prefix + "_bench_kernel_stats.csv")
counters = {}
for d in ("pmc_fetch", "pmc_write"):
    for k, v in pmc(d, "vc_scan_kernel").items():
        counters[k if not k.startswith("pass_") else d + "_" + k] = v
json.dump(counters, open(prefix + "_bench_pmc_scan_kernel.json", "w"), indent=1)
cfg = b["config"]
alg = cfg["n_codes"] * cfg["bits"] // 8
hbm = (counters["FETCH_SIZE"] * 2 + counters["WRITE_SIZE"]) * 1024
json.dump({
    "kernel": "vc_scan_kernel", "n_codes": cfg["n_codes"], "bits": cfg["bits"], "query_tile": cfg["query_tile"],
    "FETCH_SIZE_KB_per_launch": counters["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": counters["WRITE_SIZE"],
    "correction": "FETCH_SIZE x 2 (wide coalesced 16 B/lane stream on gfx950), WRITE_SIZE as is; separate --pmc passes",
    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
    "bench_line_traffic": b["roofline"].get("traffic"),
}, open(prefix + "_scan_traffic.json", "w"), indent=1)
print("c3: %.1f q/s, scan %.4f ms, frac %.3f, traffic/alg %.4f (bench line: %s); extras: %s" % (
    b["value"], b["roofline"]["avg_launch_ms"], b["roofline"]["frac"], hbm / alg, b["roofline"].get("traffic"),
    {k: (round(v.get("value", 0)), round(v.get("frac") or 0, 3)) for k, v in b.get("extras", {}).items()}))

# ---- the other workloads
for w in ("c1", "sharded1dev", "knn_mih_1e9", "knn_mih", "knn_mih_q16k", "knn_mih_1e9_q16k", "knn_approx", "c2", "knn_uniform", "knn_uniform_1e9"):
    bw = bench(w)
    json.dump(bw, open("%s_bench_%s.json" % (prefix, w), "w"))
    if w not in ("knn_uniform", "knn_uniform_1e9", "c1", "knn_mih_q16k", "knn_mih_1e9_q16k"):
        stats(w, "%s_%s_kernel_stats.csv" % (prefix, w))
    print("%s: %.1f q/s, %s %.4f ms/launch, cpu %.1f q/s on %d threads" % (
        w, bw["value"], bw["roofline"]["kernel"], bw["roofline"].get("avg_launch_ms") or float("nan"),
        bw.get("cpu_baseline", {}).get("value", float("nan")), bw.get("cpu_baseline", {}).get("cores", 0)))

# ---- MIH counters: effective traffic of the probing kernels next to the algorithmic bytes
for w, bname, kernel in (("knn_mih_1e9", "knn_mih_1e9", "mih_query_kernel"), ("knn_mih", "knn_mih", "mih_query_kernel"),
                         ("knn_mih_q16k", "knn_mih_q16k", "mih_query_kernel"),
                         ("knn_approx", "knn_approx", "mih_query_kernel"),
                         ("c2m2", "c2", "mih_query_kernel"), ("c2m4", "c2", "mih_bucket_stream_kernel")):
    bw = bench(bname)
    fetch = pmc("pmc_%s_fetch" % w, kernel)
    tcc = pmc("pmc_%s_tcc" % w, kernel)
    if fetch is None:
        print("%s: no %s dispatches in the counter pass" % (w, kernel))
        continue
    rl = bw["roofline"] if w != "c2m4" else bw["config"]["variants"]["m4_s16"]["roofline"]
    raw = fetch["FETCH_SIZE"] * 1024
    hit, miss = (tcc or {}).get("TCC_HIT_sum"), (tcc or {}).get("TCC_MISS_sum")
    out = {
        "kernel": kernel, "workload": bw["config"]["workload"] + (" [m = 4 variant]" if w == "c2m4" else ""),
        "FETCH_SIZE_KB_per_launch": fetch["FETCH_SIZE"], "fetch_pass_avg_ms": fetch["pass_avg_ms"], "fetch_pass_launches": fetch["pass_launches"],
        "TCC_HIT_sum_per_launch": hit, "TCC_MISS_sum_per_launch": miss,
        "l2_hit_rate": hit / (hit + miss) if hit is not None and hit + miss > 0 else None,
        "memory_side_bytes_per_launch_raw": raw, "memory_side_bytes_per_launch_x2": raw * 2,
        "memory_side_GBps_raw": raw / (fetch["pass_avg_ms"] * 1e-3) / 1e9,
        "note": "FETCH_SIZE = 64-byte read requests of the L2 towards memory; for 4..16-byte gathers the gfx950 x2 correction of a wide "
                "stream is uncalibrated (raw and x2 both given); for the streaming kernel's 512-byte-per-instruction reads x2 applies",
        "bench_algorithmic_bytes_per_launch": rl.get("algorithmic_bytes_per_launch"), "bench_per_query": rl.get("per_query"),
        "bench_avg_launch_ms": rl.get("avg_launch_ms"),
    }
    if rl.get("algorithmic_bytes_per_launch"):
        out["traffic_over_algorithmic_raw"] = raw / rl["algorithmic_bytes_per_launch"]
    if kernel == "mih_query_kernel":
        # the yardstick of a gather kernel: 64-byte requests per second against the random-sector ceiling measured with
        # tools/ubench_sectors.hip (profiles/r03_ubench_sectors.txt): 54 G/s over a 2 GB footprint, 49-50 G/s over 16-128 GB
        peak = 50.0 if w == "knn_mih_1e9" else 54.0
        ach = raw / 64.0 / (fetch["pass_avg_ms"] * 1e-3) / 1e9
        out["sectors"] = {"achieved_G_per_s": ach, "peak_G_per_s": peak, "frac": ach / peak,
                          "peak_how": "tools/ubench_sectors.hip: independent random-sector loads, %s footprint" % ("64-128 GB" if w == "knn_mih_1e9" else "2 GB")}
    json.dump(out, open("%s_mih_%s_pmc.json" % (prefix, w), "w"), indent=1)
    print("%s pmc: FETCH %.0f KB/launch (raw %.1f MB = %.2f TB/s), L2 hit rate %s, algorithmic %.1f MB/launch" % (
        w, fetch["FETCH_SIZE"], raw / 1e6, out["memory_side_GBps_raw"] / 1e3, out["l2_hit_rate"],
        (rl.get("algorithmic_bytes_per_launch") or 0) / 1e6))

if os.path.exists("%s/shard/timeline.txt" % src):
    shutil.copy("%s/shard/timeline.txt" % src, prefix + "_shard_timeline.txt")
    shutil.copy("%s/shard/bench.json" % src, prefix + "_shard_bench.json")
