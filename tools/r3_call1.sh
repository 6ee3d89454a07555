#!/bin/bash
# round 3, call 1: today's baseline -- GPU suite incl. the new full-size oracle legs, then MIH k-NN at 1e9 / 1e8
set -o pipefail
mkdir -p gpurun_out/r3c1
python -c "import os; print('nproc', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))" > gpurun_out/r3c1/host.txt 2>&1
free -g >> gpurun_out/r3c1/host.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r3c1/pytest.txt 2>&1 || { tail -30 gpurun_out/r3c1/pytest.txt; exit 1; }
tail -25 gpurun_out/r3c1/pytest.txt
timeout -k 10 300 python bench.py --workload knn_mih --db-size 1e9 --steps 10 --no-traffic > gpurun_out/r3c1/knn_mih_1e9.json 2> gpurun_out/r3c1/knn_mih_1e9.err || { tail -20 gpurun_out/r3c1/knn_mih_1e9.err; exit 1; }
cat gpurun_out/r3c1/knn_mih_1e9.json
timeout -k 10 300 python bench.py --workload knn_mih --steps 10 --no-traffic --cpu-seconds 0 > gpurun_out/r3c1/knn_mih_1e8.json 2> gpurun_out/r3c1/knn_mih_1e8.err || { tail -20 gpurun_out/r3c1/knn_mih_1e8.err; exit 1; }
cat gpurun_out/r3c1/knn_mih_1e8.json
