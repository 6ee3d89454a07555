#!/bin/bash
# Dev tool: effective shader clock (GRBM_GUI_ACTIVE / 8 / kernel time) of the verify kernel in three regimes.
cd $GRAFT_REPO_ROOT
VC_BUILD_DIAG=1 python -m verticut_amd.build --force > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for cfg in "hbm 8 0" "hbm 1 0" "cached 8 512" "cached 16 512" "hbm 16 0"; do
  set -- $cfg
  export VC_SCAN_WRAP=$3
  [ "$3" = "0" ] && unset VC_SCAN_WRAP
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/clk_$1_$2 -- python $GRAFT_REPO_ROOT/tools/prof_scan.py 1e9 128 $2 4 > $GRAFT_REPO_ROOT/gpurun_out/clk_$1_$2.log 2>&1 || exit 1
  python - "$1" "$2" $GRAFT_REPO_ROOT/gpurun_out/clk_$1_$2 <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[3] + "/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "vc_scan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
for r in rows[-2:]:
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"{sys.argv[1]:7s} qt={sys.argv[2]:3s} kernel {ms:7.3f} ms  GRBM_GUI_ACTIVE {float(r['Counter_Value']):.0f}  -> {float(r['Counter_Value'])/8/ms/1e6:.3f} GHz")
PY
done
cd $GRAFT_REPO_ROOT && python -m verticut_amd.build --force > /dev/null 2>&1
