#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3c14
mkdir -p $O
run() {  # name, env...
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- python3 bench.py --workload knn_mih --steps 10 --warmup 3 --no-extras --no-traffic --cpu-seconds 0 > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  echo "== $name $(python3 -c "import json;d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])")"
  grep "vc_mih\]" $O/$name.err < /dev/null | tail -2
  python3 - <<P
import csv
for r in csv.DictReader(open("$O/prof_$name/run_kernel_stats.csv")):
    if any(x in r["Name"] for x in ("mih_tile","mih_query","work_reduce")):
        print("   ", r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
P
}
run base VC_MIH_TILE=0 VC_MIH_TRACE=1 || exit 1
run tile VC_MIH_TILE=1 VC_MIH_TRACE=1 || exit 1
run tile_g3 VC_MIH_TILE=1 VC_MIH_GROUP=3 || exit 1
run base_g3 VC_MIH_TILE=0 VC_MIH_GROUP=3 || exit 1
run base_g2 VC_MIH_TILE=0 VC_MIH_GROUP=2 || exit 1
for v in w8g4 w5g4; do
  run ${v}_g3 VC_MIH_TILE=1 VC_MIH_GROUP=3 VERTICUT_GPU_LIB=$PWD/verticut_amd/lib/variants/libvc_$v.so || exit 1
done
exit 0
