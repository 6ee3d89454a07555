#!/bin/bash
# Dev tool (GPU box): A/B of verify-kernel variants inside one gpurun call (boxes differ by up to 7 % in raw HBM rate).
# usage: tools/ab_scan.sh <outfile> ; prints "label qps ms_per_step frac" per variant, 1e9 and 1.25e8 codes
OUT=${1:-gpurun_out/ab_scan.txt}
L=$GRAFT_REPO_ROOT/verticut_amd/lib
run() {  # label, env...
  label=$1; shift
  for n in 1e9 1.25e8; do
    r=$(env "$@" python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-traffic --db-size $n 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.1f qps  %.4f ms/step  scan %.4f ms  frac %.3f  %s' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['results_check']))")
    echo "$label n=$n  $r" | tee -a $OUT
  done
}
run "general-loop(r1)      " VC_SCAN_SMALL=0
run "small w5 nb2          " VC_SCAN_SMALL=1
run "small w4 nb2          " VERTICUT_GPU_LIB=$L/libverticut_gpu_w4.so
run "general-loop(r1) again" VC_SCAN_SMALL=0
run "small w5 nb2 again    " VC_SCAN_SMALL=1
