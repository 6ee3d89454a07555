#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c13; mkdir -p $O
cd $R
line() { python - "$1" "$2" <<'P'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("%-14s n_gpus %d  %.1f q/s  step %.4f ms  scan %.4f ms  frac %.3f  exchange %s  check %s" % (sys.argv[1], d["n_gpus"], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["config"].get("exchange"), d["results_check"]))
P
}
for bpc in 8 4 2 8 4; do
  VC_SAMPLE_BLOCKS_PER_CU=$bpc timeout -k 10 200 python bench.py --db-size 1.25e8 --steps 60 --no-traffic --no-extras --cpu-seconds 0 > $O/shard_bpc$bpc.json 2> $O/shard_bpc$bpc.err || { tail -5 $O/shard_bpc$bpc.err; exit 1; }
  line "sample_bpc=$bpc" $O/shard_bpc$bpc.json
done
VC_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python bench.py --db-size 1.25e8 --steps 40 --no-traffic --no-extras --cpu-seconds 0 > $O/force_exchange.json 2> $O/force_exchange.err || { tail -8 $O/force_exchange.err; exit 1; }
line "rccl 1 rank" $O/force_exchange.json
VC_BENCH_FORCE_EXCHANGE=1 VC_BENCH_BUCKET=1 timeout -k 10 200 python bench.py --db-size 1.25e8 --steps 40 --no-traffic --no-extras --cpu-seconds 0 > $O/force_exchange_b1.json 2> $O/force_exchange_b1.err || { tail -8 $O/force_exchange_b1.err; exit 1; }
line "rccl bucket=1" $O/force_exchange_b1.json
VC_BENCH_ONE_GPU=1 VC_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --db-size 2.5e8 --steps 20 --warmup 3 --cpu-seconds 0 > $O/two_ranks.json 2> $O/two_ranks.err || { tail -8 $O/two_ranks.err; exit 1; }
line "2 ranks/1 GPU" $O/two_ranks.json
