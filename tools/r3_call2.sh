#!/bin/bash
# round 3, call 2: GPU suite with the wave-per-query MIH kernel + A/B of its variants at 1e8, then 1e9
set -o pipefail
O=gpurun_out/r3c2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.txt 2>&1; rc=$?
tail -30 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
B="python bench.py --workload knn_mih --steps 10 --no-traffic --cpu-seconds 0"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > $O/knn_$name.json 2> $O/knn_$name.err || { echo "FAILED $name"; tail -5 $O/knn_$name.err; return 1; }; python - "$name" $O/knn_$name.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("%-12s %.2f M q/s  step %.3f ms  kernel avg %.3f ms x %d launches  check %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"], r["launches"], d["results_check"]))
P
}
run default X=1 && run nowave VC_MIH_WAVE=0 && run w4 VERTICUT_GPU_LIB=$PWD/verticut_amd/lib/variants/libvc_w4.so && run w4g2e2 VERTICUT_GPU_LIB=$PWD/verticut_amd/lib/variants/libvc_w4g2e2.so && run nopair VC_MIH_PAIR01=0 && run shells2 VC_MIH_WAVE_SHELLS=2 && run default2 X=1
B="python bench.py --workload knn_mih --db-size 1e9 --steps 10 --no-traffic --cpu-seconds 0"
run 1e9 X=1
