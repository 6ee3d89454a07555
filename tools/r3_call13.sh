#!/bin/bash
# tile pipeline: per-kernel times
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3c13
mkdir -p $O
run() {  # name, env...
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- python3 bench.py --workload knn_mih --steps 10 --warmup 3 --no-extras --no-traffic --cpu-seconds 0 > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  f=$(find $O/prof_$name -name '*kernel_stats.csv' | head -1)
  echo "== $name $(python3 -c "import json;d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])")"
  [ -n "$f" ] || { echo "no stats file"; return 1; }
  grep -E "mih_tile|mih_query_kernel|mih_work_reduce" "$f" < /dev/null | cut -c1-60,120-260 | awk -F, '{print $0}' | head -8
}
run base VC_MIH_TILE=0 || exit 1
run tile VC_MIH_TILE=1 || exit 1
for v in w8g4 w5g4 w8g2 v8; do
  [ -f verticut_amd/lib/variants/libvc_$v.so ] && { run $v VC_MIH_TILE=1 VERTICUT_GPU_LIB=$PWD/verticut_amd/lib/variants/libvc_$v.so || exit 1; }
done
exit 0
