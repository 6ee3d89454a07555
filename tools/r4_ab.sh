#!/bin/bash
# Dev tool (GPU box): same-box A/B of library variants / runtime knobs on a bench workload.
#   tools/r4_ab.sh <out-dir> <bench args in quotes> <variant>...     variant = name[:lib][:ENV=val[,ENV=val...]]
#     lib = "base" (the shipped library) or a name under verticut_amd/lib/variants/libvc_<lib>.so (tools/build_variant.sh)
# e.g. tools/r4_ab.sh r4ab "--workload knn_mih" old:ff0:VC_MIH_LINES=0 new:base:VC_MIH_LINES=1
set -o pipefail
O=$PWD/gpurun_out/$1; shift
BARGS=$1; shift
mkdir -p $O
for rep in 1 2; do
for spec in "$@"; do
  IFS=: read -r name libn envs <<< "$spec"
  lib=$PWD/verticut_amd/lib/libverticut_gpu.so
  [ -n "$libn" ] && [ "$libn" != base ] && lib=$PWD/verticut_amd/lib/variants/libvc_$libn.so
  E=()
  IFS=, read -ra kv <<< "$envs"
  for x in "${kv[@]}"; do [ -n "$x" ] && E+=("$x"); done
  env VERTICUT_GPU_LIB=$lib "${E[@]}" timeout -k 10 400 python3 bench.py $BARGS --no-extras --no-traffic --cpu-seconds 0 > $O/$name.$rep.json 2> $O/$name.$rep.err || { echo "$name FAILED"; tail -5 $O/$name.$rep.err; continue; }
  python3 - <<P | tee -a $O/summary.txt
import json
d=json.loads(open("$O/$name.$rep.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("%-14s rep$rep  %12.0f q/s  step %.4f ms  kernel %.4f ms  %s" % ("$name", d["value"], d["ms_per_step"], r.get("avg_launch_ms") or 0, d.get("results_check")))
P
  { grep -h "vc_mih phases\|vc_mih lifetimes" $O/$name.$rep.err || true; } | tail -2 | tee -a $O/summary.txt
done
done
exit 0
