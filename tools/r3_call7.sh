#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c7; mkdir -p $O
cd $R
VC_MIH_WAVE=0 VC_MIH_PHASES=1 timeout -k 10 200 python bench.py --workload knn_mih --steps 3 --warmup 1 --no-traffic --cpu-seconds 0 > $O/ph.json 2> $O/ph.err
grep "phases" $O/ph.err | tail -4
VC_MIH_WAVE=0 VC_MIH_PHASES=1 timeout -k 10 200 python bench.py --workload knn_mih --db-size 1e9 --steps 3 --warmup 1 --no-traffic --cpu-seconds 0 > $O/ph9.json 2> $O/ph9.err
grep "phases" $O/ph9.err | tail -2
