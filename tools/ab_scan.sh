#!/bin/bash
# Dev tool (GPU box): A/B of verify-kernel variants inside one gpurun call (boxes differ by up to 7 % in raw HBM rate).
OUT=${1:-gpurun_out/ab_scan.txt}
L=$GRAFT_REPO_ROOT/verticut_amd/lib
run() {  # label, env...
  label=$1; shift
  for n in 1e9 1.25e8; do
    r=$(env "$@" python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-traffic --db-size $n 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.1f qps  %.4f ms/step  scan %.4f ms  frac %.3f  %s' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['results_check']))")
    echo "$label n=$n  $r" | tee -a $OUT
  done
}
for rep in 1 2 3; do
run "small nb2 w4  rep$rep" VC_SCAN_SMALL=1
run "small nb2 w3  rep$rep" VERTICUT_GPU_LIB=$L/libverticut_gpu_w3.so
done
