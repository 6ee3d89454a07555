#!/bin/bash
# Dev tool (GPU box): kernel timeline of a 125 M-code shard step (what one rank of 8 does) -> gpurun_out/<tag>/timeline.txt
TAG=${1:-tl}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp VC_BENCH_TIMING_SAMPLE=${2:-4}
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --db-size 1.25e8 --steps 30 --warmup 5 --cpu-seconds 0 --no-traffic --no-extras > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('shard step %.4f ms  scan %.4f ms  %.1f qps %s' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['value'], j['results_check']))"
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $(ls $OUT/trace/*/*_kernel_trace.csv | head -1) 2 > $OUT/timeline.txt
cat $OUT/timeline.txt
