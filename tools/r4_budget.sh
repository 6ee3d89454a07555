#!/bin/bash
# Dev tool (GPU box): what do the shell-3 queries cost a launch?  VC_MIH_BUDGET (probes a query may spend inside mih_query_kernel)
# at 0 = default (shells 0..4 in the block), 30000 (0..3), 3000 (0..2: the 6 % that need shell 3 continue in the multi-block kernels)
cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1  %.0f q/s  step %.4f ms  kernel %.4f ms  %s' % (j['value'], j['ms_per_step'], r.get('avg_launch_ms') or 0, j.get('results_check')))"; }
for Q in 4096 16384; do for b in 0 30000 3000; do
  VC_MIH_BUDGET=$b python3 bench.py --workload knn_mih --queries $Q --steps 12 --warmup 3 --no-traffic --cpu-seconds 0 --no-extras 2>/dev/null | line "Q=$Q budget=$b"
done; done
VC_MIH_TRACE=1 VC_MIH_BUDGET=3000 python3 bench.py --workload knn_mih --queries 4096 --steps 2 --warmup 1 --no-traffic --cpu-seconds 0 --no-extras 2>&1 >/dev/null | grep "vc_mih" | tail -8
