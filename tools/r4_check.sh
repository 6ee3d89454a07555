#!/bin/bash
# Dev tool (GPU box): the round-end checks in one call -- whole GPU suite, smoke(), the driver-like bench run and its extras
mkdir -p gpurun_out/r4c50 && python -m pytest tests -m gpu -x -q > gpurun_out/r4c50/pytest.txt 2>&1; echo pytest rc $?; tail -2 gpurun_out/r4c50/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4c50/bench.json 2> gpurun_out/r4c50/bench.err ) 2>&1 | grep real
python3 - <<'P'
import json
j=json.loads(open("gpurun_out/r4c50/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], j["results_check"])
for k,v in j["extras"].items(): print(k, round(v.get("value",0)), v.get("results_check"), v.get("error"))
P
