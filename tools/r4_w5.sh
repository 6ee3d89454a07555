#!/bin/bash
# Dev tool (GPU box): mih_query_kernel compiled for 5 waves per SIMD (96 VGPRs, spills) with a 512-entry hit list (29 KB of LDS: 5 blocks per CU)
cd $GRAFT_REPO_ROOT
V=$PWD/verticut_amd/lib/variants
VERTICUT_GPU_LIB=$V/libvc_w5.so python -m pytest tests/test_mih_gpu.py tests/test_fixtures_gpu.py -x -q 2>&1 | tail -2
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1  %.0f q/s  step %.4f ms  kernel %.4f ms  %s' % (j['value'], j['ms_per_step'], r.get('avg_launch_ms') or 0, j.get('results_check')))"; }
for n in 1e8 1e9; do for Q in 4096 16384; do for lib in base h512 w5; do
  L=$PWD/verticut_amd/lib/libverticut_gpu.so; [ $lib != base ] && L=$V/libvc_$lib.so
  VERTICUT_GPU_LIB=$L python3 bench.py --workload knn_mih --db-size $n --queries $Q --steps $((49152/Q)) --warmup 2 --no-traffic --cpu-seconds 0 --no-extras 2>/dev/null | line "n=$n Q=$Q $lib"
done; done; done
