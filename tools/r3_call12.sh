#!/bin/bash
# tile pipeline: correctness first, then an A/B at 1e8
set -o pipefail
mkdir -p gpurun_out/r3c12
python -m pytest tests/test_mih_gpu.py -x -q -k "tile_pipeline" --timeout=600 --timeout-method=thread > gpurun_out/r3c12/pytest_tile.txt 2>&1
rc=$?
tail -15 gpurun_out/r3c12/pytest_tile.txt
[ $rc -ne 0 ] && exit $rc
for t in 0 1; do
  VC_MIH_TILE=$t timeout -k 10 300 python bench.py --workload knn_mih --steps 10 --warmup 3 --no-extras > gpurun_out/r3c12/knn_tile$t.json 2> gpurun_out/r3c12/knn_tile$t.err || { tail -5 gpurun_out/r3c12/knn_tile$t.err; exit 1; }
  python - <<P
import json
d=json.loads(open("gpurun_out/r3c12/knn_tile$t.json").read().strip().splitlines()[-1])
print("tile=$t", d["value"], d["ms_per_step"], d.get("roofline"))
P
done
