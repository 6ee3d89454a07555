#!/bin/bash
# Dev tool (GPU box): size of the threshold bootstrap's sample (VC_SAMPLE1) -- threshold convergence costs the verify
# pass nothing (profiles/r02_sweeps.md), so how small can the sample be?  A/B inside one gpurun call.
OUT=${1:-gpurun_out/boot.txt}
for rep in 1 2 3; do
for s in default 131072 262144 524288 1048576; do
  if [ $s = default ]; then ev="VC_X=0"; else ev="VC_SAMPLE1=$s"; fi
  for n in 1.25e8 1e9; do
    r=$(env $ev timeout -k 10 300 python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-traffic --db-size $n 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f ms/step  scan %.4f ms  fixed %.1f us  %s' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], (j['ms_per_step']-j['roofline']['avg_launch_ms'])*1e3, j['results_check']))") || exit 1
    echo "sample=$s rep$rep n=$n  $r" | tee -a $OUT
  done
done
done
