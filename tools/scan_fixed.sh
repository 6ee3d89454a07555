#!/bin/bash
# Dev tool (GPU box): is the fixed ~30 us of a verify launch the threshold's convergence (rare-path appends while tau
# is still the sample's)?  Compare the default bootstrap (1.5 M-code sample) with an exact threshold (VC_SAMPLE1 = n:
# the bootstrap then costs a full pass, only the verify launch time matters here).
OUT=${1:-gpurun_out/scan_fixed.txt}
for rep in 1 2; do
for n in 62500000 125000000 250000000; do
  for s in default $n; do
    if [ $s = default ]; then envs="VC_X=0"; else envs="VC_SAMPLE1=$s"; fi
    r=$(env $envs timeout -k 10 300 python bench.py --steps 20 --warmup 4 --cpu-seconds 0 --no-traffic --db-size $n 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f ms/step  scan %.4f ms  %s' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['results_check']))") || exit 1
    echo "rep$rep n=$n sample=$s  $r" | tee -a $OUT
  done
done
done
