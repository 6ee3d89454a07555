#!/bin/bash
# round 3, call 5: full suite (scan switch, bucket streaming) + c2 bench both variants
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c5; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.txt 2>&1; rc=$?
tail -30 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
for cfg in "stream X=1" "nostream VC_MIH_STREAM=0"; do set -- $cfg; name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload c2 --steps 10 --no-traffic --cpu-seconds 0 > $O/c2_$name.json 2> $O/c2_$name.err || { echo FAILED $name; tail -5 $O/c2_$name.err; exit 1; }
  python - $name $O/c2_$name.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("c2 %-9s m2 %.3f M q/s (kernel %.3f ms)  check %s" % (sys.argv[1], d["value"]/1e6, r["avg_launch_ms"], d["results_check"]))
for k,v in d["config"]["variants"].items():
    rr=v["roofline"]
    print("   %s %.3f M q/s  step %.3f ms  kernel %s %s ms  achieved %s GB/s  check %s" % (k, v["value"]/1e6, v["ms_per_step"], rr.get("kernel"), rr.get("avg_launch_ms"), rr.get("achieved"), v["results_check"]))
P
done
