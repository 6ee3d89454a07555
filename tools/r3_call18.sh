#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r3c18
mkdir -p $O
python -m pytest tests/test_mih_gpu.py tests/test_fixtures_gpu.py tests/test_host_driver_gpu.py -x -q --timeout=900 --timeout-method=thread > $O/pytest.txt 2>&1
rc=$?; tail -3 $O/pytest.txt; [ $rc -ne 0 ] && exit $rc
bash tools/r3_ab.sh r3c18 g4 base || exit 1
AB_ARGS="--db-size 1e9" bash tools/r3_ab.sh r3c18_1e9 g4 base || exit 1
python tests/campaign/parity_campaign_mih.py 600 30000 > $O/campaign.txt 2>&1 || { tail -5 $O/campaign.txt; exit 1; }
tail -1 $O/campaign.txt
