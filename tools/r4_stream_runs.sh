#!/bin/bash
# Dev tool (GPU box): mih_bucket_stream_kernel's fraction of the HBM peak against the LENGTH of the contiguous runs it streams
# (a bucket of the 16-bit tables holds n / 65 536 entries of 8 bytes): configs[1] m = 4 at several database sizes.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-runs}; mkdir -p $O
for n in 2.5e7 5e7 1e8 2e8 4e8 8e8; do
  timeout -k 10 400 python3 bench.py --workload c2 --tables 4 --db-size $n --steps 10 --warmup 3 --no-traffic --cpu-seconds 0 > $O/n$n.json 2> $O/n$n.err || { echo "n=$n failed"; tail -3 $O/n$n.err; continue; }
  python3 - <<P | tee -a $O/summary.txt
import json
d=json.loads(open("$O/n$n.json").read().strip().splitlines()[-1]); r=d["roofline"]
pq=r["per_query"]
print("n=%-8s bucket run %7.0f B  entries/query %8.0f  %9.0f q/s  kernel %.4f ms  %6.0f GB/s  frac %.3f  %s" % ("$n", float("$n")/65536*8, pq["entries_verified"], d["value"], r["avg_launch_ms"], r["achieved"], r["frac"], d["results_check"]))
P
done
