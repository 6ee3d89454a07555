#!/bin/bash
# Dev tool (GPU box): exact top-100 through MIH against the queries per launch (VC_MIH_QTILE) and per call -- a launch ends with
# its longest query, so it carries a fixed tail (profiles/r04_sweeps.md, 9; made there with a -DMIH_QTILE build, now a knob)
cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1  %.0f q/s  step %.4f ms  kernel %.4f ms  %s' % (j['value'], j['ms_per_step'], r.get('avg_launch_ms') or 0, j.get('results_check')))"; }
for n in 1e8 1e9; do for rep in 1 2; do
  for spec in "4096 4096 24" "16384 8192 12" "16384 16384 6" "4096 16384 6" "32768 32768 4"; do set -- $spec
    VC_MIH_QTILE=$1 python3 bench.py --workload knn_mih --db-size $n --queries $2 --steps $3 --warmup 2 --no-traffic --cpu-seconds 0 --no-extras 2>/dev/null | line "n=$n tile $1 Q=$2 rep$rep"
  done
done; done
