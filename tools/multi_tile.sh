#!/bin/bash
# Dev tool (GPU box): is the ~30 us a verify launch costs beyond its bytes paid again by a second launch right behind
# the first?  8 / 16 / 32 / 64 queries per step at a tile of 8 = 1 / 2 / 4 / 8 verify launches per step.
for rep in 1 2; do
for q in 8 16 32 64; do
  r=$(VC_BENCH_QUERY_TILE=8 timeout -k 10 300 python bench.py --steps 20 --warmup 4 --cpu-seconds 0 --no-traffic --db-size 1.25e8 --queries $q 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f ms/step  scan %.4f ms per launch (%d launches)  %s' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['launches'], j['results_check']))") || exit 1
  echo "queries/step=$q rep$rep  $r"
done
done
