#!/bin/bash
# Dev tool (GPU box): kernel timeline of the exact MIH k-NN shells (tools/bench_mih.py knn) from a rocprofv3 kernel trace.
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mihknn -- python $GRAFT_REPO_ROOT/tools/bench_mih.py knn 1e8 > $GRAFT_REPO_ROOT/gpurun_out/mihknn.log 2>&1
cd $GRAFT_REPO_ROOT
grep "^knn" gpurun_out/mihknn.log | head -3
python - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/mihknn/*/*_kernel_trace.csv"))[-1]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx=[i for i,r in enumerate(rows) if "mih_probe" in r[2]]
start=idx[len(idx)//4]
prev=None
for s,e,n in rows[start-2:start+40]:
    n=n.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40]
    print("gap %7.1f run %8.1f  %s" % ((s-prev)/1e3 if prev else 0,(e-s)/1e3,n)); prev=e
PY
