#!/bin/bash
# Dev tool (run on the GPU box through gpurun): every measured artefact of a round in one call.
#   bench lines (all workloads), rocprofv3 --kernel-trace --stats of the same commands, separate --pmc passes
#   (FETCH_SIZE | WRITE_SIZE GRBM_GUI_ACTIVE | SQ_* for the verify kernel; FETCH_SIZE | TCC_HIT_sum TCC_MISS_sum for the
#   MIH query kernel: one TCC-heavy counter set per pass, the program directly after `--`), a 125 M-code shard timeline.
# usage: tools/profile_round2.sh <tag>   -> files under gpurun_out/<tag>/ ; tools/summarize_round2.py turns them into profiles/
set -o pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
step() { echo "== $*" | tee -a $OUT/log.txt; }
step bench c3; python bench.py --steps 30 --warmup 5 > $OUT/bench_c3.json 2> $OUT/bench_c3.err || { tail -20 $OUT/bench_c3.err; exit 1; }
cat $OUT/bench_c3.json
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-traffic"
step stats c3; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- $B > $OUT/stats_c3.log 2>&1 || { tail -5 $OUT/stats_c3.log; exit 1; }
B5="python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-traffic"
step pmc c3 fetch; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B5 > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
step pmc c3 write; rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B5 > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
step pmc c3 sq; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B5 > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
for w in c2 knn_mih c5shard; do
  step bench $w; (cd $R && python bench.py --workload $w --cpu-seconds 8 > $OUT/bench_$w.json 2> $OUT/bench_$w.err) || { tail -20 $OUT/bench_$w.err; exit 1; }
  cat $OUT/bench_$w.json
  step stats $w; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --cpu-seconds 0 --no-check --no-traffic > $OUT/stats_$w.log 2>&1 || { tail -5 $OUT/stats_$w.log; exit 1; }
done
for w in c2 knn_mih; do   # (c2: the m = 2 engine only, so that the per-launch averages are of one kernel shape)
  step pmc $w fetch; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_${w}_fetch -- python3 $R/bench.py --workload $w --tables 2 --steps 4 --warmup 2 --cpu-seconds 0 --no-check --no-traffic > $OUT/pmc_${w}_fetch.log 2>&1 || { tail -5 $OUT/pmc_${w}_fetch.log; exit 1; }
  step pmc $w tcc; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_${w}_tcc -- python3 $R/bench.py --workload $w --tables 2 --steps 4 --warmup 2 --cpu-seconds 0 --no-check --no-traffic > $OUT/pmc_${w}_tcc.log 2>&1 || { tail -5 $OUT/pmc_${w}_tcc.log; exit 1; }
done
step shard timeline; $R/tools/timeline_shard.sh $TAG/shard 8 > $OUT/shard_timeline.log 2>&1 || { tail -5 $OUT/shard_timeline.log; exit 1; }
tail -30 $OUT/shard_timeline.log
# keep what is merged back small: the raw traces are large
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
du -sh $OUT
