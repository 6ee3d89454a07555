#!/bin/bash
# single-word probes: MIH suite, then same-box A/B at 1e8
set -o pipefail
O=$PWD/gpurun_out/r3c16
mkdir -p $O
python -m pytest tests/test_mih_gpu.py -x -q --timeout=900 --timeout-method=thread > $O/pytest_mih.txt 2>&1
rc=$?
tail -5 $O/pytest_mih.txt
[ $rc -ne 0 ] && exit $rc
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload knn_mih --steps 20 --warmup 3 --no-extras --no-traffic --cpu-seconds 0 > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python3 - <<P
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name", round(d["value"]), d["ms_per_step"], d["roofline"]["avg_launch_ms"])
P
}
run pre VERTICUT_GPU_LIB=$PWD/verticut_amd/lib/variants/libvc_pre.so || exit 1
run cold A=1 || exit 1
run pre2 VERTICUT_GPU_LIB=$PWD/verticut_amd/lib/variants/libvc_pre.so || exit 1
run cold2 A=1 || exit 1
VERTICUT_GPU_LIB=$PWD/verticut_amd/lib/variants/libvc_pre.so timeout -k 10 300 python3 bench.py --workload c2 --steps 20 --warmup 3 --no-extras --no-traffic --cpu-seconds 0 > $O/c2pre.json 2> $O/c2pre.err || exit 1
timeout -k 10 300 python3 bench.py --workload c2 --steps 20 --warmup 3 --no-extras --no-traffic --cpu-seconds 0 > $O/c2cold.json 2> $O/c2cold.err || exit 1
python3 - <<P
import json
for n in ("c2pre","c2cold"):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(d["value"]), d["ms_per_step"], {k:(round(v["value"]) if isinstance(v,dict) and "value" in v else None) for k,v in d.get("variants",{}).items()})
P
python tests/campaign/parity_campaign_mih.py 300 8000 > $O/campaign.txt 2>&1 || { tail -5 $O/campaign.txt; exit 1; }
tail -2 $O/campaign.txt
