#!/bin/bash
# final validation of the round (GPU box): whole GPU suite, the round's profile set, the two seeded parity campaigns on the final code
# usage (through gpurun): bash tools/r4_final.sh [cases [seed0]]
set -o pipefail
CASES=${1:-1500}; SEED=${2:-400000}
O=$PWD/gpurun_out/r4final
mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.txt 2>&1
rc=$?
tail -4 $O/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
tools/profile_round4.sh r04e > $O/profile.log 2>&1 || { tail -5 $O/profile.log; exit 1; }
tail -3 $O/profile.log
python tests/campaign/parity_campaign_mih.py $CASES $SEED > $O/campaign_mih.txt 2>&1 || { tail -5 $O/campaign_mih.txt; exit 1; }
tail -1 $O/campaign_mih.txt
python tests/campaign/parity_campaign.py $CASES $SEED > $O/campaign_lin.txt 2>&1 || { tail -5 $O/campaign_lin.txt; exit 1; }
tail -1 $O/campaign_lin.txt
