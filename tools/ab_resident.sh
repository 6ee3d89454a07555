#!/bin/bash
# Dev tool (GPU box): database prefix kept in the Infinity Cache by the verify pass (VC_SCAN_RESIDENT_MB), one gpurun call.
OUT=${1:-gpurun_out/resident.txt}
for rep in 1 2; do
for mb in 0 64 128 192 224 256 320; do
  for n in 1.25e8 1e9; do
    r=$(VC_SCAN_RESIDENT_MB=$mb timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-traffic --db-size $n 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.1f qps  %.4f ms/step  scan %.4f ms  frac %.3f  %s' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['results_check']))") || exit 1
    echo "resident=${mb}MB rep$rep n=$n  $r" | tee -a $OUT
  done
done
done
