#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r3c15
mkdir -p $O
for g in 2 3; do
  VC_MIH_TILE=1 VC_MIH_GROUP=$g VC_MIH_PHASES=1 timeout -k 10 200 python3 bench.py --workload knn_mih --steps 3 --warmup 1 --no-extras --no-traffic --cpu-seconds 0 --no-check > $O/g$g.json 2> $O/g$g.err || { tail -5 $O/g$g.err; exit 1; }
  echo "== group $g"; grep "phases\]" $O/g$g.err | tail -2
done
