#!/bin/bash
# round 3, call 4: where does the two-stage MIH k-NN spend its time (kernel stats + SQ counters); tau fold A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c4; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_linear_gpu.py tests/test_edge_gpu.py tests/test_random_gpu.py tests/test_sharded_gpu.py tests/test_fixtures_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
tail -8 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
V=$R/verticut_amd/lib/variants
for cfg in "fold X=1" "nofold VC_TAU_FOLD=0" "fold2 X=1" "nofold2 VC_TAU_FOLD=0"; do set -- $cfg; name=$1; shift
  for size in 1e9 1.25e8; do
    env "$@" timeout -k 10 200 python bench.py --db-size $size --steps 40 --no-traffic --no-extras --cpu-seconds 0 > $O/c3_${name}_$size.json 2> $O/c3_${name}_$size.err || { echo FAILED $name $size; tail -5 $O/c3_${name}_$size.err; exit 1; }
    python - $name $size $O/c3_${name}_$size.json <<'P'
import json,sys
d=json.load(open(sys.argv[3])); r=d["roofline"]
print("c3 %-8s %-7s %.1f q/s  step %.4f ms  scan %.4f ms  frac %.3f  check %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["results_check"]))
P
  done
done
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload knn_mih --steps 10 --cpu-seconds 0 --no-check --no-traffic"
for cfg in "w3 X=1" "w4 VERTICUT_GPU_LIB=$V/libvc_w4.so" "nowave VC_MIH_WAVE=0"; do set -- $cfg; name=$1; shift
  export "$@"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -- $B > $O/stats_$name.log 2>&1 || { tail -5 $O/stats_$name.log; exit 1; }
  unset X VERTICUT_GPU_LIB VC_MIH_WAVE
  f=$(find $O/stats_$name -name '*kernel_stats.csv' | head -1); echo "== $name"; head -8 $f | cut -c1-200
done
export VERTICUT_GPU_LIB=$V/libvc_w4.so
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --workload knn_mih --steps 3 --warmup 1 --cpu-seconds 0 --no-check --no-traffic > $O/pmc_sq.log 2>&1 || { tail -5 $O/pmc_sq.log; exit 1; }
python3 - $O/pmc_sq <<'P'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "mih_" in k: acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: "%.3g"%(sum(x[-3:])/len(x[-3:])) for c,x in v.items()})
P
