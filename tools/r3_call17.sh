#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r3c17
mkdir -p $O
for g in 0 3; do
  VC_MIH_GROUP=$g VC_MIH_PHASES=1 timeout -k 10 200 python3 bench.py --workload knn_mih --steps 3 --warmup 1 --no-extras --no-traffic --cpu-seconds 0 --no-check > $O/g$g.json 2> $O/g$g.err || { tail -5 $O/g$g.err; exit 1; }
  echo "== group $g"; grep "lifetimes\]\|phases\]" $O/g$g.err | tail -4
done
VC_MIH_PHASES=1 timeout -k 10 300 python3 bench.py --workload knn_mih --db-size 1e9 --steps 3 --warmup 1 --no-extras --no-traffic --cpu-seconds 0 --no-check > $O/e9.json 2> $O/e9.err || { tail -5 $O/e9.err; exit 1; }
echo "== 1e9"; grep "lifetimes\]\|phases\]" $O/e9.err | tail -2
