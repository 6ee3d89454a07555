#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c11; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -v --timeout=300 --timeout-method=thread --durations=5 > $O/pytest.txt 2>&1; rc=$?
grep -v PASSED $O/pytest.txt | tail -16
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - $O/bench_default.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("headline %.1f q/s  step %.4f ms  scan %.4f ms frac %.3f (excl resident %.1f GB/s) traffic %s check %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["achieved_excl_resident"], r["traffic"], d["results_check"]))
print("cpu", {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != "sample"}) for k, v in d["cpu_baseline"].items() if k != "sample"})
for k, v in d.get("extras", {}).items(): print("extra", k, {kk: vv for kk, vv in v.items() if kk not in ("workload", "per_query")})
P
