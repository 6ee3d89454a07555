#!/bin/bash
# dev tool: the two seeded parity campaigns only -- bash tools/r3_campaign_only.sh <cases> <seed0>
set -o pipefail
CASES=${1:-3000}; SEED=${2:-100000}
O=$PWD/gpurun_out/r3camp
mkdir -p $O
python tests/campaign/parity_campaign_mih.py $CASES $SEED > $O/campaign_mih.txt 2>&1 || { tail -5 $O/campaign_mih.txt; exit 1; }
tail -1 $O/campaign_mih.txt
python tests/campaign/parity_campaign.py $CASES $SEED > $O/campaign_lin.txt 2>&1 || { tail -5 $O/campaign_lin.txt; exit 1; }
tail -1 $O/campaign_lin.txt
