#!/bin/bash
# Dev tool (GPU box): threshold-bootstrap variants (stage sizes), alternating repetitions
OUT=${1:-gpurun_out/ab_boot.txt}
run() {
  label=$1; shift
  for n in 1e9 1.25e8; do
    r=$(env "$@" python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-traffic --db-size $n 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f ms/step  scan %.4f ms  fixed %.1f us  %s' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], (j['ms_per_step']-j['roofline']['avg_launch_ms'])*1e3, j['results_check']))")
    echo "$label n=$n  $r" | tee -a $OUT
  done
}
for rep in 1 2 3 4; do
run "default (64K + 2M) rep$rep"
run "s1=1M   s2=0      rep$rep" VC_SAMPLE1=1048576 VC_SAMPLE2=0
run "s1=1.5M s2=0      rep$rep" VC_SAMPLE1=1572864 VC_SAMPLE2=0
done
