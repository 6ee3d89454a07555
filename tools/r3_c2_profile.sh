#!/bin/bash
# dev tool: the configs[1] part of tools/profile_round3.sh again (into the same output directory), after a change to the radius kernels
set -o pipefail
TAG=${1:-r03c}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -m pytest tests/test_mih_gpu.py -x -q --timeout=900 --timeout-method=thread > $OUT/pytest_mih_c2.txt 2>&1 || { tail -5 $OUT/pytest_mih_c2.txt; exit 1; }
tail -1 $OUT/pytest_mih_c2.txt
python bench.py --workload c2 --cpu-seconds 8 --no-traffic > $OUT/bench_c2.json 2> $OUT/bench_c2.err || { tail -20 $OUT/bench_c2.err; exit 1; }
cut -c1-300 $OUT/bench_c2.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/stats_c2 $OUT/pmc_c2m4_fetch $OUT/pmc_c2m4_tcc $OUT/pmc_c2m2_fetch $OUT/pmc_c2m2_tcc
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $R/bench.py --workload c2 --cpu-seconds 0 --no-check --no-traffic > $OUT/stats_c2.log 2>&1 || { tail -5 $OUT/stats_c2.log; exit 1; }
for w in "c2m4 --workload c2 --tables 4" "c2m2 --workload c2 --tables 2"; do set -- $w; name=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_${name}_fetch -- python3 $R/bench.py "$@" --steps 4 --warmup 2 --cpu-seconds 0 --no-check --no-traffic > $OUT/pmc_${name}_fetch.log 2>&1 || { tail -5 $OUT/pmc_${name}_fetch.log; exit 1; }
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_${name}_tcc -- python3 $R/bench.py "$@" --steps 4 --warmup 2 --cpu-seconds 0 --no-check --no-traffic > $OUT/pmc_${name}_tcc.log 2>&1 || { tail -5 $OUT/pmc_${name}_tcc.log; exit 1; }
done
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
echo done
