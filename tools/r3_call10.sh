#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c10; mkdir -p $O
cd $R
V=$R/verticut_amd/lib/variants
for lib in b128 b64; do
VERTICUT_GPU_LIB=$V/libvc_$lib.so timeout -k 10 600 python -m pytest tests/test_mih_gpu.py tests/test_fixtures_gpu.py -m gpu -x -q --timeout=300 --timeout-method=thread > $O/pytest_$lib.txt 2>&1; rc=$?
tail -3 $O/pytest_$lib.txt
[ $rc -ne 0 ] && exit 1
done
B="python bench.py --workload knn_mih --steps 10 --no-traffic --cpu-seconds 0"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > $O/knn_$name.json 2> $O/knn_$name.err || { echo "FAILED $name"; tail -5 $O/knn_$name.err; return 1; }; python - "$name" $O/knn_$name.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("%-12s %.2f M q/s  step %.3f ms  kernels %.3f ms per step in %d launches  check %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"]*r["launches"]/d["steps"], r["launches"], d["results_check"]))
P
}
run default X=1 && run b128 VERTICUT_GPU_LIB=$V/libvc_b128.so && run b128h256 VERTICUT_GPU_LIB=$V/libvc_b128h256.so && run b64 VERTICUT_GPU_LIB=$V/libvc_b64.so && run b64g3 VERTICUT_GPU_LIB=$V/libvc_b64.so VC_MIH_GROUP=3 && run b128g2 VERTICUT_GPU_LIB=$V/libvc_b128.so VC_MIH_GROUP=2
B="python bench.py --workload knn_mih --db-size 1e9 --steps 10 --no-traffic --cpu-seconds 0"
run 1e9 X=1 && run 1e9b128 VERTICUT_GPU_LIB=$V/libvc_b128.so && run 1e9b64 VERTICUT_GPU_LIB=$V/libvc_b64.so
B="python bench.py --workload c2 --tables 2 --steps 10 --no-traffic --cpu-seconds 0"
run c2 X=1 && run c2b128 VERTICUT_GPU_LIB=$V/libvc_b128.so
