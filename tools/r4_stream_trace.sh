#!/bin/bash
# Dev tool (GPU box): mih_bucket_stream_kernel at configs[1] (m = 4): per-block start / look-up / end times (VC_STREAM_TRACE) and
# A/B of runtime knobs -> gpurun_out/<tag>/summary.txt
TAG=${1:-r4st}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
B="--workload c2 --tables 4 --steps 10 --warmup 3 --no-traffic --cpu-seconds 0 --no-extras"
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1  %.0f q/s  step %.4f ms  kernel %.4f ms  frac %.3f  %s' % (j['value'], j['ms_per_step'], r.get('avg_launch_ms') or 0, r.get('frac') or 0, j.get('results_check')))"; }
for rep in 1 2; do
  python3 bench.py $B 2> $O/base.$rep.err | line "base (5/CU resident, 2048 blocks) rep$rep" | tee -a $O/summary.txt
  VC_STREAM_LDS_PAD=12288 python3 bench.py $B 2> $O/pad.$rep.err | line "4/CU resident, 2048 blocks rep$rep" | tee -a $O/summary.txt
  VC_STREAM_LDS_PAD=12288 VC_STREAM_BLOCKS_PER_CU=4 python3 bench.py $B 2> $O/pad4.$rep.err | line "4/CU resident, 1024 blocks rep$rep" | tee -a $O/summary.txt
  VC_STREAM_BLOCKS_PER_CU=4 python3 bench.py $B 2> $O/b4.$rep.err | line "5/CU resident, 1024 blocks rep$rep" | tee -a $O/summary.txt
done
VC_STREAM_LDS_PAD=12288 VC_STREAM_TRACE=1 python3 bench.py --workload c2 --tables 4 --steps 2 --warmup 1 --no-traffic --cpu-seconds 0 --no-extras 2> $O/tracepad.err | line "trace pad" | tee -a $O/summary.txt
grep "stream trace" $O/tracepad.err | tail -7 | tee -a $O/summary.txt
VC_STREAM_BLOCKS_PER_CU=4 VC_STREAM_TRACE=1 python3 bench.py --workload c2 --tables 4 --steps 2 --warmup 1 --no-traffic --cpu-seconds 0 --no-extras 2> $O/traceb4.err | line "trace 1024 blocks" | tee -a $O/summary.txt
grep "stream trace" $O/traceb4.err | tail -7 | tee -a $O/summary.txt
exit 0
