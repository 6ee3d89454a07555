#!/bin/bash
# Dev tool (GPU box): per-block start / end times of the persistent verify grid (VC_SCAN_TRACE, diagnostic build) on a 125 M-code
# shard (2 GB pass) and on 1e9 codes: where a shard-size pass loses against the 1e9 pass -- ramp-up, tail, steady state.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-trace}; mkdir -p $O
L=$GRAFT_REPO_ROOT/verticut_amd/lib/variants/libvc_diag.so
for n in 1.25e8 1e9; do
  VERTICUT_GPU_LIB=$L VC_SCAN_TRACE=1 timeout -k 10 300 python3 bench.py --db-size $n --steps 6 --warmup 2 --cpu-seconds 0 --no-traffic --no-extras > $O/trace_$n.json 2> $O/trace_$n.err
  echo "== n = $n" | tee -a $O/summary.txt
  grep "scan trace" $O/trace_$n.err | grep -v "appended items by pass" | tail -9 | tee -a $O/summary.txt
done
