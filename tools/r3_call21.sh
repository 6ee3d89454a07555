#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r3c21
mkdir -p $O
python -m pytest tests/test_mih_gpu.py tests/test_host_driver_gpu.py tests/test_edge_gpu.py tests/test_random_gpu.py tests/test_sharded_native_gpu.py -x -q --timeout=900 --timeout-method=thread > $O/pytest.txt 2>&1
rc=$?; tail -3 $O/pytest.txt; [ $rc -ne 0 ] && exit $rc
python -m pytest tests/test_fullsize_gpu.py -x -q -k "config2" --timeout=900 --timeout-method=thread > $O/pytest_full.txt 2>&1
rc=$?; tail -3 $O/pytest_full.txt; [ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
for pl in 0 1; do
  VC_MIH_POLL=$pl timeout -k 10 300 python3 bench.py --workload c2 --steps 30 --warmup 3 --no-extras --no-traffic --cpu-seconds 0 > $O/p$pl.$rep.json 2> $O/p$pl.$rep.err || { tail -5 $O/p$pl.$rep.err; exit 1; }
  python3 - <<P
import json
d=json.loads(open("$O/p$pl.$rep.json").read().strip().splitlines()[-1])
v=d["config"]["variants"]["m4_s16"]
print("poll $pl", round(d["value"]), round(d["ms_per_step"],4), round(d["roofline"]["avg_launch_ms"],4), d.get("results_check"), "| m4", round(v["value"]), round(v["ms_per_step"],4))
P
done
done
