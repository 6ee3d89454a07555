#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c9; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_mih_gpu.py tests/test_fixtures_gpu.py tests/test_random_gpu.py tests/test_edge_gpu.py -m gpu -x -q --timeout=300 --timeout-method=thread > $O/pytest.txt 2>&1; rc=$?
tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
V=$R/verticut_amd/lib/variants
B="python bench.py --workload knn_mih --steps 10 --no-traffic --cpu-seconds 0"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > $O/knn_$name.json 2> $O/knn_$name.err || { echo "FAILED $name"; tail -5 $O/knn_$name.err; return 1; }; python - "$name" $O/knn_$name.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("%-12s %.2f M q/s  step %.3f ms  kernels %.3f ms per step in %d launches  check %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"]*r["launches"]/d["steps"], r["launches"], d["results_check"]))
P
}
run default X=1 && run g2 VC_MIH_GROUP=2 && run g1 VC_MIH_GROUP=1 && run h512 VERTICUT_GPU_LIB=$V/libvc_h512.so && run occ5 VERTICUT_GPU_LIB=$V/libvc_occ5.so && run occ5e2 VERTICUT_GPU_LIB=$V/libvc_occ5e2.so && run occ6 VERTICUT_GPU_LIB=$V/libvc_occ6.so && run default2 X=1
B="python bench.py --workload knn_mih --db-size 1e9 --steps 10 --no-traffic --cpu-seconds 0"
run 1e9 X=1 && run 1e9occ5 VERTICUT_GPU_LIB=$V/libvc_occ5.so && run 1e9occ6 VERTICUT_GPU_LIB=$V/libvc_occ6.so
