#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c14; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q --timeout=300 --timeout-method=thread > $O/pytest.txt 2>&1; rc=$?
tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
for cfg in "bent X=1" "nobent VC_MIH_BENT=0"; do set -- $cfg; name=$1; shift
env "$@" timeout -k 10 300 python bench.py --workload knn_mih --db-size 1e9 --steps 10 --no-traffic --cpu-seconds 0 > $O/knn9_$name.json 2> $O/knn9_$name.err || { tail -8 $O/knn9_$name.err; exit 1; }
python - $name $O/knn9_$name.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("1e9 %-8s %.3f M q/s  step %.3f ms  kernel %.3f ms  check %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"], d["results_check"]))
P
done
python tools/bench_build.py 1e9 128 4 2>&1 | tail -2
