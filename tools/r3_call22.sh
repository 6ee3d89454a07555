#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r3c22
mkdir -p $O
python -m pytest tests/test_mih_gpu.py tests/test_host_driver_gpu.py tests/test_fixtures_gpu.py -x -q --timeout=900 --timeout-method=thread > $O/pytest.txt 2>&1
rc=$?; tail -3 $O/pytest.txt; [ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
for w in knn_mih c2; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 30 --warmup 3 --no-extras --no-traffic --cpu-seconds 0 > $O/$w.$rep.json 2> $O/$w.$rep.err || { tail -5 $O/$w.$rep.err; exit 1; }
  python3 - <<P
import json
d=json.loads(open("$O/$w.$rep.json").read().strip().splitlines()[-1])
print("$w", round(d["value"]), round(d["ms_per_step"],4), round(d["roofline"]["avg_launch_ms"],4), d.get("results_check"))
P
done
done
