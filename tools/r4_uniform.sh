#!/bin/bash
# Dev tool (GPU box): uniform-random queries in MIH_EXACT mode (answered by the verify kernel through the cost-model switch) after
# the switch's passes went from 8 to 32 queries; switch / MIH / sharded suites first
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_mih_switch_gpu.py tests/test_mih_gpu.py tests/test_fixtures_gpu.py tests/test_sharded_native_gpu.py -x -q 2>&1 | tail -2
for spec in "1e8 64" "1e9 64" "1e9 256"; do set -- $spec
  python3 bench.py --workload knn_mih --uniform-queries --queries $2 --db-size $1 --steps 3 --warmup 1 --no-traffic --cpu-seconds 0 --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('uniform n=$1 Q=$2', round(j['value'],1), j['ms_per_step'], j['results_check'])"
done
