#!/bin/bash
# Dev tool (GPU box): where do the microseconds between two verify launches of a 125 M-code shard step go?  Kernel timelines
# under rocprofv3 (null stream / a stream of the bench's own) and the same step un-profiled -> gpurun_out/<tag>/summary.txt
# (profiles/r04_fixed_cost.txt: the ~5.6 us hole rocprofv3 shows at every CALL boundary is the profiler's, not the step's)
TAG=${1:-r4fixed}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
run() {   # name ENV=val...
  local name=$1; shift
  echo "=== $name $*" | tee -a $O/summary.txt
  env "$@" bash $GRAFT_REPO_ROOT/tools/timeline_shard.sh $TAG/$name 4 2>&1 | tail -14 | tee -a $O/summary.txt
  rm -rf $O/$name/trace
}
run base VC_NOP=1
run ownstream VC_BENCH_STREAM=own
# the same two un-profiled (step time from the bench's own clock)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in null own; do
    VC_BENCH_STREAM=$v VC_BENCH_TIMING_SAMPLE=4 python3 bench.py --db-size 1.25e8 --steps 200 --warmup 20 --cpu-seconds 0 --no-traffic --no-extras 2> $O/plain_$v.$rep.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v rep$rep  step %.4f ms  median %.4f  scan %.4f ms' % (j['ms_per_step'], j.get('median_ms_per_step') or 0, j['roofline']['avg_launch_ms']))" | tee -a $O/summary.txt
  done
done
exit 0
