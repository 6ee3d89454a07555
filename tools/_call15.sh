#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c15; mkdir -p $O
cd $R
python tools/bench_build.py 1e9 128 4 2>&1 | tail -2
python tools/bench_build.py 1e8 128 4 2>&1 | tail -2
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout=300 --timeout-method=thread > $O/pytest.txt 2>&1; rc=$?
tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --workload knn_mih --db-size 1e9 --steps 10 --no-traffic --cpu-seconds 0 > $O/knn9.json 2> $O/knn9.err || { tail -8 $O/knn9.err; exit 1; }
python - $O/knn9.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("1e9 %.3f M q/s  step %.3f ms  kernel %.3f ms  check %s" % (d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"], d["results_check"]))
P
