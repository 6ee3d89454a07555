#!/bin/bash
# Dev tool (run on the GPU box through gpurun): the measured artefacts of round 4 in one call.
#   default bench line (with the round's extras: e2e_host, q1, qt32, c2_m2, knn_mih / knn_approx at 1e8, knn_mih / knn_uniform
#   at 1e9) + rocprofv3 --kernel-trace --stats of the same command + separate --pmc passes (FETCH_SIZE | WRITE_SIZE) for the
#   verify kernel; the other workloads' lines (c1, sharded1dev, knn_mih at 1e9 / 1e8, approximate, c2, uniform), kernel stats and
#   FETCH_SIZE | TCC passes of the MIH kernels; a 125 M-code shard timeline.
# usage: tools/profile_round4.sh <tag>   -> gpurun_out/<tag>/ ; tools/summarize_round4.py runs at the end (on the box) and leaves what
# goes to profiles/ in gpurun_out/<tag>/summary/r04_*; the raw passes are deleted
set -o pipefail
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
step() { echo "== $*" | tee -a $OUT/log.txt; }
step bench c3; python bench.py --steps 30 --warmup 5 > $OUT/bench_c3.json 2> $OUT/bench_c3.err || { tail -20 $OUT/bench_c3.err; exit 1; }
cut -c1-400 $OUT/bench_c3.json
for w in "c1 --workload c1" "sharded1dev --workload sharded1dev" "knn_mih_1e9 --workload knn_mih --db-size 1e9" "knn_mih --workload knn_mih" "knn_mih_q16k --workload knn_mih --queries 16384 --steps 8" "knn_mih_1e9_q16k --workload knn_mih --db-size 1e9 --queries 16384 --steps 8" "knn_approx --workload knn_mih --approximate" "c2 --workload c2" "knn_uniform --workload knn_mih --uniform-queries --queries 64" "knn_uniform_1e9 --workload knn_mih --uniform-queries --queries 64 --db-size 1e9"; do set -- $w; name=$1; shift
  step bench $name; python bench.py "$@" --cpu-seconds 8 --no-traffic > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -20 $OUT/bench_$name.err; exit 1; }
  cut -c1-300 $OUT/bench_$name.json
done
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-traffic --no-extras"
step stats c3; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- $B > $OUT/stats_c3.log 2>&1 || { tail -5 $OUT/stats_c3.log; exit 1; }
B5="python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-traffic --no-extras"
step pmc c3 fetch; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B5 > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
step pmc c3 write; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B5 > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
for w in "knn_mih_1e9 --workload knn_mih --db-size 1e9" "knn_mih --workload knn_mih" "knn_approx --workload knn_mih --approximate" "c2 --workload c2" "sharded1dev --workload sharded1dev"; do set -- $w; name=$1; shift
  step stats $name; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 $R/bench.py "$@" --cpu-seconds 0 --no-check --no-traffic > $OUT/stats_$name.log 2>&1 || { tail -5 $OUT/stats_$name.log; exit 1; }
done
for w in "knn_mih_1e9 --workload knn_mih --db-size 1e9" "knn_mih --workload knn_mih" "knn_mih_q16k --workload knn_mih --queries 16384" "knn_approx --workload knn_mih --approximate" "c2m4 --workload c2 --tables 4" "c2m2 --workload c2 --tables 2"; do set -- $w; name=$1; shift
  step pmc $name fetch; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_${name}_fetch -- python3 $R/bench.py "$@" --steps 4 --warmup 2 --cpu-seconds 0 --no-check --no-traffic > $OUT/pmc_${name}_fetch.log 2>&1 || { tail -5 $OUT/pmc_${name}_fetch.log; exit 1; }
  step pmc $name tcc; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_${name}_tcc -- python3 $R/bench.py "$@" --steps 4 --warmup 2 --cpu-seconds 0 --no-check --no-traffic > $OUT/pmc_${name}_tcc.log 2>&1 || { tail -5 $OUT/pmc_${name}_tcc.log; exit 1; }
done
step shard timeline; $R/tools/timeline_shard.sh $TAG/shard 8 > $OUT/shard_timeline.log 2>&1 || { tail -5 $OUT/shard_timeline.log; exit 1; }
tail -30 $OUT/shard_timeline.log
find $OUT -name "*.db" -delete
# summarised on the box: gpurun copies at most 64 MiB back, the raw passes of this script are more
mkdir -p $OUT/summary
python3 $R/tools/summarize_round4.py $OUT $OUT/summary/r04 > $OUT/summary/summary.log 2>&1 || { tail -5 $OUT/summary/summary.log; exit 1; }
cat $OUT/summary/summary.log
rm -rf $OUT/stats_* $OUT/pmc_* $OUT/shard/trace
du -sh $OUT
