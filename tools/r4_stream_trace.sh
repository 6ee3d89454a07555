#!/bin/bash
# Dev tool (GPU box): mih_bucket_stream_kernel at configs[1] (m = 4): the launch's time (two repeats) and the per-block start /
# look-up / end clocks of VC_STREAM_TRACE=1 -> gpurun_out/<tag>/summary.txt.  (profiles/r04_stream_trace.txt was made with
# experiment builds of this script that also carried knobs for the work counters, the blocks per CU and the parts per query.)
TAG=${1:-r4st}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
B="--workload c2 --tables 4 --steps 10 --warmup 3 --no-traffic --cpu-seconds 0 --no-extras"
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1  %.0f q/s  step %.4f ms  kernel %.4f ms  frac %.3f  %s' % (j['value'], j['ms_per_step'], r.get('avg_launch_ms') or 0, r.get('frac') or 0, j.get('results_check')))"; }
for rep in 1 2; do
  python3 bench.py $B 2> $O/base.$rep.err | line "rep$rep" | tee -a $O/summary.txt
done
VC_STREAM_TRACE=1 python3 bench.py --workload c2 --tables 4 --steps 2 --warmup 1 --no-traffic --cpu-seconds 0 --no-extras 2> $O/trace.err | line "trace" | tee -a $O/summary.txt
grep "stream trace" $O/trace.err | tail -7 | tee -a $O/summary.txt
exit 0
