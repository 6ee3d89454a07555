#!/bin/bash
# round 3, call 8: suite after removing the wave stage / grouped first pass; A/B of the group depth; phases
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c8; mkdir -p $O
cd $R
B="python bench.py --workload knn_mih --steps 10 --no-traffic --cpu-seconds 0"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > $O/knn_$name.json 2> $O/knn_$name.err || { echo "FAILED $name"; tail -5 $O/knn_$name.err; return 1; }; python - "$name" $O/knn_$name.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("%-12s %.2f M q/s  step %.3f ms  kernels %.3f ms per step in %d launches  check %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"]*r["launches"]/d["steps"], r["launches"], d["results_check"]))
P
grep phases $O/knn_$name.err | tail -1 || true
}
run adaptive X=1 && run g1 VC_MIH_GROUP=1 VC_MIH_PHASES=1 && run g2 VC_MIH_GROUP=2 VC_MIH_PHASES=1 && run g3 VC_MIH_GROUP=3 VC_MIH_PHASES=1 && run adaptive2 X=1
B="python bench.py --workload knn_mih --db-size 1e9 --steps 10 --no-traffic --cpu-seconds 0"
run 1e9 VC_MIH_PHASES=1 && run 1e9g2 VC_MIH_GROUP=2
