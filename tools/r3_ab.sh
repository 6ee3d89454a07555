#!/bin/bash
# same-box A/B of libverticut_gpu.so variants on the exact-MIH bench: tools/r3_ab.sh <out> <name>... (name "base" = the shipped library)
set -o pipefail
O=$PWD/gpurun_out/$1; shift
mkdir -p $O
for rep in 1 2; do
for name in "$@"; do
  lib=$PWD/verticut_amd/lib/variants/libvc_$name.so
  [ "$name" = base ] && lib=$PWD/verticut_amd/lib/libverticut_gpu.so
  VERTICUT_GPU_LIB=$lib timeout -k 10 300 python3 bench.py --workload knn_mih --steps 20 --warmup 3 --no-extras --no-traffic --cpu-seconds 0 ${AB_ARGS} > $O/$name.$rep.json 2> $O/$name.$rep.err || { tail -5 $O/$name.$rep.err; exit 1; }
  python3 - <<P
import json
d=json.loads(open("$O/$name.$rep.json").read().strip().splitlines()[-1])
print("$name", round(d["value"]), round(d["ms_per_step"],4), round(d["roofline"]["avg_launch_ms"],4), d.get("results_check"))
P
done
done
