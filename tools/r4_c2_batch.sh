#!/bin/bash
# Dev tool (GPU box): configs[1] (radius search) against the queries per call -- does a launch carry a fixed tail here too?
cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1  %.0f q/s  step %.4f ms  kernel %.4f ms  %s' % (j['value'], j['ms_per_step'], r.get('avg_launch_ms') or 0, j.get('results_check')))"; }
for m in 2 4; do for Q in 1024 2048 4096; do
  python3 bench.py --workload c2 --tables $m --queries $Q --steps 10 --warmup 3 --no-traffic --cpu-seconds 0 --no-extras 2>/dev/null | line "m=$m Q=$Q"
done; done
