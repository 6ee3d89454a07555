#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r3c19
mkdir -p $O
for rep in 1 2; do
for g in 0 3; do
  VC_MIH_GROUP=$g timeout -k 10 300 python3 bench.py --workload knn_mih --steps 20 --warmup 3 --no-extras --no-traffic --cpu-seconds 0 > $O/g$g.$rep.json 2> $O/g$g.$rep.err || { tail -5 $O/g$g.$rep.err; exit 1; }
  python3 - <<P
import json
d=json.loads(open("$O/g$g.$rep.json").read().strip().splitlines()[-1])
print("group $g", round(d["value"]), round(d["ms_per_step"],4), round(d["roofline"]["avg_launch_ms"],4), d.get("results_check"))
P
done
done
