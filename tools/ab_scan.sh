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
run "small nb2 w4          " VC_SCAN_SMALL=1
run "small nb1 w8          " VC_SCAN_SHAPE=4,256,0
run "small nb1 w6          " VC_SCAN_SHAPE=4,256,0 VERTICUT_GPU_LIB=$L/libverticut_gpu_w4.so
run "general               " VC_SCAN_SMALL=0
run "small nb2 w4 again    " VC_SCAN_SMALL=1
run "small nb1 w8 again    " VC_SCAN_SHAPE=4,256,0
