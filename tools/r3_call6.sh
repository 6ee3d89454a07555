#!/bin/bash
# round 3, call 6: full suite with per-test time limits (scan switch, bucket streaming, histogram-threshold block kernel) + benches
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c6; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -v --timeout=240 --timeout-method=thread --durations=8 > $O/pytest.txt 2>&1; rc=$?
grep -v PASSED $O/pytest.txt | tail -30
[ $rc -ne 0 ] && exit 1
B="python bench.py --workload knn_mih --steps 10 --no-traffic --cpu-seconds 0"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > $O/knn_$name.json 2> $O/knn_$name.err || { echo "FAILED $name"; tail -5 $O/knn_$name.err; return 1; }; python - "$name" $O/knn_$name.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("%-12s %.2f M q/s  step %.3f ms  kernels %.3f ms per step in %d launches  check %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"]*r["launches"]/d["steps"], r["launches"], d["results_check"]))
P
}
run default X=1 && run nowave VC_MIH_WAVE=0 && run nowave_nopair VC_MIH_WAVE=0 VC_MIH_PAIR01=0 && run shells2 VC_MIH_WAVE_SHELLS=2 && run default2 X=1
B="python bench.py --workload knn_mih --db-size 1e9 --steps 10 --no-traffic --cpu-seconds 0"
run 1e9 X=1 && run 1e9nowave VC_MIH_WAVE=0
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
