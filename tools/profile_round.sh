#!/bin/bash
# Dev tool (run on the GPU box through gpurun): bench line + rocprofv3 kernel stats + PMC passes for the verify kernel.
# usage: tools/profile_round.sh <tag>   -> files under gpurun_out/<tag>/
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
# per-kernel time of the same command (must agree with roofline.avg_launch_ms)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --cpu-seconds 0 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
# PMC passes (counters only with --kernel-trace): HBM traffic, then SQ occupancy/wait breakdown
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
ls -R $OUT | head -40
