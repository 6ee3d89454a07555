#!/bin/bash
# final validation of the round: whole GPU suite, then the two seeded parity campaigns on the final kernels
# usage (through gpurun): bash tools/r3_final.sh [cases [seed0]]
set -o pipefail
CASES=${1:-3000}; SEED=${2:-20000}
O=$PWD/gpurun_out/r3final
mkdir -p $O
python -m pytest tests -m gpu -x -q --timeout=900 --timeout-method=thread > $O/pytest_gpu.txt 2>&1
rc=$?
tail -4 $O/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
python tests/campaign/parity_campaign_mih.py $CASES $SEED > $O/campaign_mih.txt 2>&1 || { tail -5 $O/campaign_mih.txt; exit 1; }
tail -1 $O/campaign_mih.txt
python tests/campaign/parity_campaign.py $CASES $SEED > $O/campaign_lin.txt 2>&1 || { tail -5 $O/campaign_lin.txt; exit 1; }
tail -1 $O/campaign_lin.txt
