#!/bin/bash
# Dev tool (GPU box): SQ / SQC counters of mih_query_kernel on the exact-MIH bench -> gpurun_out/<tag>/pmc_*.txt
TAG=${1:-pmc}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACCUM_PREV_HIRES SQ_LEVEL_WAVES"; do
  d=$O/run_$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py --workload knn_mih --steps 4 --warmup 2 --no-extras --no-traffic --cpu-seconds 0 --no-check "$@" > $d.log 2>&1 || { echo "set [$set] failed"; tail -3 $d.log; continue; }
  python3 - "$d" <<'P' | tee -a $O/summary.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mih_query_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[1:] if len(v) > 1 else v
    print("%-28s %16.0f per launch (%d launches)" % (k, sum(v) / len(v), len(v)))
P
done
