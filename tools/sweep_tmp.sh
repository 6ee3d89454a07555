timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { python bench.py --db-size $1 --steps 40 --warmup 10 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps %.0f  step %.4f ms  scan %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
for b in 2 4 8; do export VC_SAMPLE_BLOCKS_PER_CU=$b; for r in 1 2; do echo "== 125M blocks/cu=$b"; run 125000000; done; done
unset VC_SAMPLE_BLOCKS_PER_CU
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl125e -- python $GRAFT_REPO_ROOT/bench.py --db-size 125000000 --steps 30 --warmup 5 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/tl125e.log 2>&1; cd $GRAFT_REPO_ROOT && python tools/step_timeline.py $(ls gpurun_out/tl125e/*/*_kernel_trace.csv) 1
